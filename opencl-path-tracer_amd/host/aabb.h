// Axis-aligned box used by the host-side BVH builders (counterpart of the reference's
// src/bvh/aabb.h; same empty-box convention min=+FLT_MAX, max=-FLT_MAX so `fit` needs no flag).
#pragma once
#include "math.h"
#include <limits>

namespace raytracer {

struct AABB {
    vec3 min { std::numeric_limits<float>::max() };
    vec3 max { std::numeric_limits<float>::lowest() };

    AABB() = default;
    AABB(vec3 lo, vec3 hi)
        : min(lo), max(hi) {}

    void fit(vec3 p)
    {
        min = vmin(min, p);
        max = vmax(max, p);
    }
    void fit(const AABB& o)
    {
        min = vmin(min, o.min);
        max = vmax(max, o.max);
    }
    AABB merged(const AABB& o) const { return { vmin(min, o.min), vmax(max, o.max) }; }
    AABB intersection(const AABB& o) const { return { vmax(min, o.min), vmin(max, o.max) }; }
    vec3 center() const { return (min + max) / 2.0f; }
    // negative extents (empty box) clamp to zero, as the reference's AABB::extent (aabb.cpp:61-64)
    vec3 extent() const { return vmax(vec3(0.0f), max - min); }
    float surfaceArea() const
    {
        vec3 e = extent();
        return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x);
    }
    bool contains(vec3 p) const
    {
        return p.x >= min.x && p.y >= min.y && p.z >= min.z && p.x <= max.x && p.y <= max.y && p.z <= max.z;
    }
    bool contains(const AABB& o) const { return contains(o.min) && contains(o.max); }
};

} // namespace raytracer
