// Camera with the reference's public surface (src/camera.h:29-60) producing the 128-byte
// CameraData / pt_camera record; parameter derivation follows src/camera.cpp:18-58.
#pragma once
#include "../../include/ptamd.h"
#include "transform.h"
#include <cstring>

namespace raytracer {

using CameraData = pt_camera;

class Camera {
public:
    Camera(const Transform& transform, float horizontalFovDeg, float aspectRatio, float focalDistance)
        : m_focalDistance(focalDistance), m_transform(transform), m_aspectRatio(aspectRatio), m_horizontalFov(horizontalFovDeg) {}

    CameraData get_camera_data() const
    {
        CameraData c;
        std::memset(&c, 0, sizeof(c));
        const mat4 world = m_transform.matrix();
        const mat3 rot = mat3_cast(m_transform.orientation);
        // thin-lens geometry: image plane distance from the lens equation 1/f = 1/d_o + 1/d_i
        const float focalLength = m_focalLengthMm / 1000.0f;
        const float apertureDiameter = focalLength / m_aperture;
        const float imageDistance = 1.0f / (1.0f / focalLength - 1.0f / m_focalDistance);
        const float halfWidth = std::tan(radians(m_horizontalFov / 2.0f)) * imageDistance;
        const float halfHeight = halfWidth / m_aspectRatio;
        put(c.eyePoint, m_transform.location);
        put(c.u, rot * vec3(2.0f * halfWidth, 0, 0));
        put(c.v, rot * vec3(0, -2.0f * halfHeight, 0));
        put(c.uNormalized, rot * vec3(1, 0, 0));
        put(c.vNormalized, rot * vec3(0, -1, 0));
        put(c.screenPoint, (world * vec4(-halfWidth, halfHeight, imageDistance, 1.0f)).xyz());
        c.focalDistance = m_focalDistance;
        c.apertureRadius = apertureDiameter / 2.0f;
        c.relativeAperture = m_aperture;
        c.shutterTime = m_shutterTime;
        c.ISO = m_iso;
        c.thinLensEnabled = m_thinLens ? 1 : 0;
        return c;
    }

    float getHorizontalFov() const { return m_horizontalFov; }
    const Transform& getTransform() const { return m_transform; }
    void setTransform(const Transform& t) { m_transform = t; }

    // public like the reference (ImGui sliders bind to them, src/camera.h:45-52); defaults camera.cpp:5-15
    float m_focalDistance;
    float m_focalLengthMm = 50.0f;
    float m_aperture = 8.0f;
    float m_shutterTime = 1.0f / 32.0f;
    float m_iso = 1200.0f;
    bool m_thinLens = true;

private:
    static void put(float* d, vec3 v) { d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = 0.0f; }
    Transform m_transform;
    float m_aspectRatio; // width / height
    float m_horizontalFov; // degrees
};

} // namespace raytracer
