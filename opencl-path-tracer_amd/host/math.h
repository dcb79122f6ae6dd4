// Minimal float vector / matrix / quaternion algebra for the host side.
// The reference uses glm (un-vendored, vcpkg baseline 6bc4362f); only the handful of
// operations the hot path's host half needs are provided here: TRS composition
// (reference src/transform.cpp:13-20), general 4x4 inverse (src/bvh/top_bvh_build.cpp:103) and
// quaternion -> matrix (src/camera.cpp:21).  Matrices are column-major, m[col][row], like glm.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace raytracer {

struct vec2 {
    float x = 0, y = 0;
};

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s)
        : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_)
        : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator-(vec3 a) { return { -a.x, -a.y, -a.z }; }
inline vec3 operator*(vec3 a, vec3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, vec3 a) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator/(vec3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline vec3& operator+=(vec3& a, vec3 b) { return a = a + b; }
inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a / length(a); }
inline vec3 vmin(vec3 a, vec3 b) { return { std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z) }; }
inline vec3 vmax(vec3 a, vec3 b) { return { std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z) }; }

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float x_, float y_, float z_, float w_)
        : x(x_), y(y_), z(z_), w(w_) {}
    vec4(vec3 v, float w_)
        : x(v.x), y(v.y), z(v.z), w(w_) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
    vec3 xyz() const { return { x, y, z }; }
};

// Unit quaternion (w, x, y, z), constructor order as glm::quat(w, x, y, z).
struct quat {
    float w = 1, x = 0, y = 0, z = 0;
    quat() = default;
    quat(float w_, float x_, float y_, float z_)
        : w(w_), x(x_), y(y_), z(z_) {}
    // from Euler angles (pitch=x, yaw=y, roll=z), radians
    static quat fromEuler(vec3 e)
    {
        float cx = std::cos(e.x * 0.5f), sx = std::sin(e.x * 0.5f);
        float cy = std::cos(e.y * 0.5f), sy = std::sin(e.y * 0.5f);
        float cz = std::cos(e.z * 0.5f), sz = std::sin(e.z * 0.5f);
        return { cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz };
    }
    static quat angleAxis(float angle, vec3 axis)
    {
        float s = std::sin(angle * 0.5f);
        return { std::cos(angle * 0.5f), axis.x * s, axis.y * s, axis.z * s };
    }
};

struct mat3 {
    vec3 c[3]; // columns
};
inline vec3 operator*(const mat3& m, vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }

struct mat4 {
    float m[4][4]; // m[col][row]
    mat4()
    {
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++)
                m[c][r] = (c == r) ? 1.0f : 0.0f;
    }
    const float* data() const { return &m[0][0]; }
};

inline mat4 operator*(const mat4& a, const mat4& b)
{
    mat4 r;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float s = 0.0f;
            for (int k = 0; k < 4; k++)
                s += a.m[k][row] * b.m[c][k];
            r.m[c][row] = s;
        }
    return r;
}

inline vec4 operator*(const mat4& a, vec4 v)
{
    vec4 r;
    for (int row = 0; row < 4; row++)
        r[row] = a.m[0][row] * v.x + a.m[1][row] * v.y + a.m[2][row] * v.z + a.m[3][row] * v.w;
    return r;
}

inline mat4 translate(vec3 t)
{
    mat4 r;
    r.m[3][0] = t.x;
    r.m[3][1] = t.y;
    r.m[3][2] = t.z;
    return r;
}

inline mat4 scale(vec3 s)
{
    mat4 r;
    r.m[0][0] = s.x;
    r.m[1][1] = s.y;
    r.m[2][2] = s.z;
    return r;
}

inline mat3 mat3_cast(quat q)
{
    float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    mat3 r;
    r.c[0] = { 1 - 2 * (yy + zz), 2 * (xy + wz), 2 * (xz - wy) };
    r.c[1] = { 2 * (xy - wz), 1 - 2 * (xx + zz), 2 * (yz + wx) };
    r.c[2] = { 2 * (xz + wy), 2 * (yz - wx), 1 - 2 * (xx + yy) };
    return r;
}

inline mat4 mat4_cast(quat q)
{
    mat3 r3 = mat3_cast(q);
    mat4 r;
    for (int c = 0; c < 3; c++) {
        r.m[c][0] = r3.c[c].x;
        r.m[c][1] = r3.c[c].y;
        r.m[c][2] = r3.c[c].z;
    }
    return r;
}

// General 4x4 inverse by Gauss-Jordan elimination with partial pivoting (double accumulators).
inline mat4 inverse(const mat4& a)
{
    double w[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            w[r][c] = a.m[c][r];
            w[r][c + 4] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(w[r][col]) > std::fabs(w[piv][col]))
                piv = r;
        if (piv != col)
            for (int c = 0; c < 8; c++)
                std::swap(w[piv][c], w[col][c]);
        double d = w[col][col];
        for (int c = 0; c < 8; c++)
            w[col][c] /= d;
        for (int r = 0; r < 4; r++) {
            if (r == col)
                continue;
            double f = w[r][col];
            if (f != 0.0)
                for (int c = 0; c < 8; c++)
                    w[r][c] -= f * w[col][c];
        }
    }
    mat4 out;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            out.m[c][r] = (float)w[r][c + 4];
    return out;
}

constexpr float kPi = 3.14159265358979323846f;
inline float radians(float deg) { return deg * (kPi / 180.0f); }

} // namespace raytracer
