// Device vector helpers and the two PRNGs of the path.
#pragma once
#include "pt_device.h"
#include <cfloat>

namespace ptd {

struct V3 {
    float x, y, z;
};
__host__ __device__ inline V3 mk(float x, float y, float z) { return { x, y, z }; }
__host__ __device__ inline V3 mk(float s) { return { s, s, s }; }
__host__ __device__ inline V3 xyz(float4 f) { return { f.x, f.y, f.z }; }
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__host__ __device__ inline V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
__host__ __device__ inline V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
__host__ __device__ inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
__host__ __device__ inline V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
__host__ __device__ inline V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
__host__ __device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
__host__ __device__ inline V3 normalize(V3 a)
{
    float len = sqrtf(dot(a, a));
    return { a.x / len, a.y / len, a.z / len };
}
__device__ inline float saturate(float a) { return fminf(fmaxf(a, 0.0f), 1.0f); }
__device__ inline float asF(uint32_t u) { return __uint_as_float(u); }
__device__ inline uint32_t asU(float f) { return __float_as_uint(f); }

// ---- production PRNG: counter-based, stateless (replaces clRNG; DESIGN.md "PRNG") ------------
// The stream of a path is named by 64 bits -- k0 = mix32(pixel ^ mix32(seed ^ phi)), k1 = mix32(sample ^ c), both
// bijections, so distinct (pixel, sample) pairs never share a stream (a 32-bit key would: a batch holds ~2^32 pairs) --
// and walks a Weyl sequence with its own odd step gamma = mix32(k0 ^ k1 ^ c') | 1, so two streams do not overlap in
// shifted windows either.  Draw number ctr = depth * 16 + dim + 1 (depth 0 = camera ray, 1 + b = shade at bounce b;
// dim = index of the draw inside that stage, draw order of SURVEY Appendix C):
//   x = k0 + gamma * ctr;  x ^= x >> 16;  x *= 0x21f0aaad;  x ^= k1;  x ^= x >> 15;  x *= 0x735a2d97;  x ^= x >> 15
//   u = (x >> 8) * 2^-24 in [0,1)      ('lowbias32' finaliser with the second key word injected between its rounds).
// Zero bytes of state traffic; independent of queue slot, compaction order and GPU count.
__host__ __device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
struct CounterKey {
    uint32_t k0, k1, gamma;
};
__host__ __device__ inline CounterKey counterKey(uint32_t pixel, uint32_t sample, uint32_t seed)
{
    CounterKey k;
    k.k0 = mix32(pixel ^ mix32(seed ^ 0x9E3779B9u));
    k.k1 = mix32(sample ^ 0x85EBCA6Bu);
    k.gamma = mix32(k.k0 ^ k.k1 ^ 0xC2B2AE35u) | 1u;
    return k;
}
__host__ __device__ inline uint32_t counterHash(uint32_t weyl, uint32_t k1)
{
    uint32_t x = weyl;
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= k1;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

// ---- parity PRNG: clRNG LFSR113 (third_party/clRNG/include/clRNG/private/lfsr113.c.h:61-93) ----
struct Rng {
    uint32_t g0, g1, g2, g3; // LFSR113 state (parity mode)
    uint32_t weyl, gamma, k1; // counter mode: weyl = k0 + gamma * (depth*16 + dim + 1) of the NEXT draw
    bool lfsr;

    __device__ inline float u01()
    {
        if (lfsr) {
            uint32_t b;
            b = ((g0 << 6) ^ g0) >> 13;
            g0 = ((g0 & 4294967294u) << 18) ^ b;
            b = ((g1 << 2) ^ g1) >> 27;
            g1 = ((g1 & 4294967288u) << 2) ^ b;
            b = ((g2 << 13) ^ g2) >> 21;
            g2 = ((g2 & 4294967280u) << 7) ^ b;
            b = ((g3 << 3) ^ g3) >> 12;
            g3 = ((g3 & 4294967168u) << 13) ^ b;
            uint32_t z = g0 ^ g1 ^ g2 ^ g3;
            return (float)((double)z * 2.3283063e-10); // double constant, as clRNG (can round to 1.0f)
        }
        const uint32_t h = counterHash(weyl, k1);
        weyl += gamma;
        return (float)(h >> 8) * (1.0f / 16777216.0f);
    }
    // clrngLfsr113RandomInteger(i, j) = i + (int)((j-i+1) * U01)
    __device__ inline int randomInteger(int i, int j)
    {
        int r = i + (int)((float)(j - i + 1) * u01());
        if (!lfsr && r > j)
            r = j;
        return r;
    }
};

__device__ inline Rng rngCounter(uint32_t pixel, uint32_t sample, uint32_t seed, uint32_t depth)
{
    Rng r;
    r.g0 = r.g1 = r.g2 = r.g3 = 0;
    const CounterKey k = counterKey(pixel, sample, seed);
    r.weyl = k.k0 + k.gamma * (depth * 16u + 1u);
    r.gamma = k.gamma;
    r.k1 = k.k1;
    r.lfsr = false;
    return r;
}
__device__ inline Rng rngLfsrLoad(const uint4* streams, uint32_t slot)
{
    uint4 s = streams[slot];
    Rng r;
    r.g0 = s.x, r.g1 = s.y, r.g2 = s.z, r.g3 = s.w;
    r.weyl = r.gamma = r.k1 = 0;
    r.lfsr = true;
    return r;
}
__device__ inline void rngLfsrStore(uint4* streams, uint32_t slot, const Rng& r) { streams[slot] = make_uint4(r.g0, r.g1, r.g2, r.g3); }

} // namespace ptd
