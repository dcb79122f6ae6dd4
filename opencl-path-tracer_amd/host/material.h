// Material factories with the reference's names and parameter meaning (src/model/material.h:63-192)
// producing the 48-byte device record pt_material (assets/cl/material.cl:3-51).
#pragma once
#include "../../include/ptamd.h"
#include "math.h"
#include <cmath>
#include <cstring>

namespace raytracer {

struct Material : pt_material {
    Material() { std::memset(static_cast<pt_material*>(this), 0, sizeof(pt_material)); }

    static Material Diffuse(vec3 colour)
    {
        Material m;
        m.type = PT_MAT_DIFFUSE;
        set3(m.u.diffuse.diffuseColour, colour);
        m.u.diffuse.textureId = -1;
        return m;
    }
    static Material Diffuse(int textureIndex, vec3 colour = vec3(0.0f))
    {
        Material m = Diffuse(colour);
        m.u.diffuse.textureId = textureIndex;
        return m;
    }
    static Material PBRMetal(vec3 reflectance, float smoothness)
    {
        Material m;
        m.type = PT_MAT_PBR;
        set3(m.u.pbr.baseColour, reflectance);
        m.u.pbr.smoothness = smoothness;
        m.u.pbr.metallic = 1;
        return m;
    }
    static Material PBRDielectric(vec3 baseColour, float smoothness, float f0 = 0.04f)
    {
        Material m;
        m.type = PT_MAT_PBR;
        set3(m.u.pbr.baseColour, baseColour);
        m.u.pbr.smoothness = smoothness;
        m.u.pbr.f0NonMetal = f0;
        m.u.pbr.metallic = 0;
        return m;
    }
    static Material Refractive(float smoothness, float refractiveIndex, vec3 colour = vec3(1.0f), float absorptionFactor = 0.0f)
    {
        Material m;
        m.type = PT_MAT_REFRACTIVE;
        m.u.refractive.smoothness = smoothness;
        m.u.refractive.refractiveIndex = refractiveIndex;
        set3(m.u.refractive.absorption, (vec3(1.0f) - colour) * absorptionFactor);
        return m;
    }
    static Material BasicRefractive(float refractiveIndex, vec3 colour = vec3(1.0f), float absorptionFactor = 0.0f)
    {
        Material m;
        m.type = PT_MAT_BASIC_REFRACTIVE;
        m.u.basicRefractive.refractiveIndex = refractiveIndex;
        set3(m.u.basicRefractive.absorption, (vec3(1.0f) - colour) * absorptionFactor);
        return m;
    }
    static Material Emissive(vec3 colour, float intensityLumen = 500.0f)
    {
        Material m;
        m.type = PT_MAT_EMISSIVE;
        set3(m.u.emissive.emissiveColour, colour * intensityLumen);
        return m;
    }
    // Black-body colour temperature (Kelvin) -> linear RGB, Tanner Helland's fit as used by the
    // reference (material.h:142-190), then scaled like Emissive(colour, lumen).
    static Material Emissive(float kelvin, float intensityLumen)
    {
        float t = kelvin / 100.0f;
        auto clamp01 = [](float v) { return std::fmin(1.0f, std::fmax(0.0f, v)); };
        float r = t <= 66.0f ? 1.0f : clamp01(329.698727446f * std::pow(t - 60.0f, -0.1332047592f) / 255.0f);
        float g = t <= 66.0f ? (99.4708025861f * std::log(t) - 161.1195681661f) : (288.1221695283f * std::pow(t - 60.0f, -0.0755148492f));
        g = clamp01(g / 255.0f);
        float b = t >= 66.0f ? 1.0f : (t <= 19.0f ? 0.0f : clamp01((138.5177312231f * std::log(t - 10.0f) - 305.0447927307f) / 255.0f));
        vec3 lin(std::pow(r, 2.2f), std::pow(g, 2.2f), std::pow(b, 2.2f));
        return Emissive(lin, intensityLumen);
    }

private:
    static void set3(float* dst, vec3 v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = 0.0f; }
};
static_assert(sizeof(Material) == 48, "Material must stay the 48-byte device record");

} // namespace raytracer
