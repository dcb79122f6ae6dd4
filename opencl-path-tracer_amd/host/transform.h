// TRS transform of a scene-graph node / camera.  API of the reference's src/transform.h:8-22;
// matrix() composes translate * rotate * scale (src/transform.cpp:13-20).
#pragma once
#include "math.h"

namespace raytracer {

struct Transform {
    vec3 location {};
    quat orientation {};
    vec3 scale { 1.0f };

    Transform() = default;
    Transform(vec3 loc, quat rot = {}, vec3 scl = vec3(1.0f))
        : location(loc), orientation(rot), scale(scl) {}

    mat4 matrix() const { return raytracer::translate(location) * mat4_cast(orientation) * raytracer::scale(scale); }
    vec3 transform(vec3 p) const { return (matrix() * vec4(p, 1.0f)).xyz(); }
    vec3 transformDirection(vec3 d) const { return mat3_cast(orientation) * d; }
};

} // namespace raytracer
