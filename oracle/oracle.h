// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// CPU restatement of the reference's ray-queue render path (mathijs727/OpenCL-Path-Tracer @ v1).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the
// checker / the reported CPU baseline.  Every function cites the reference file:line it follows.
//
// Parity status: PINNED.  The reference ships no tests or golden vectors for this path
// (SURVEY.md section 4); the restatement is instead checked (tests/test_oracle_vs_ref.py) against the
// reference's OWN kernels compiled for the host and run one work-item at a time (oracle/_ref, see
// oracle/Makefile) and against the golden vectors those kernels produced (tests/golden/).
//
// Data layouts are the reference's device layouts so arrays can be
// compared field by field with oracle/_ref outputs.
#pragma once
#include <cstddef>
#include <cstdint>

extern "C" {

struct OrcFloat3 { // OpenCL float3: 16 bytes
    float x, y, z, w;
};

struct OrcRayData { // RayData, assets/cl/shading.cl:16-29 (80 B)
    OrcFloat3 origin, direction, multiplier;
    uint64_t outputPixel;
    int32_t flags;
    union {
        float rayLength;
        int32_t numBounces;
    };
    float pdf, t;
    uint8_t _pad[8];
};

struct OrcShadingData { // ShadingData, assets/cl/kernel_data.cl:26-33 (32 B)
    float uv[2];
    const float* invTransform;
    int32_t triangleIndex;
    float t;
    uint8_t hit;
    uint8_t _pad[7];
};

struct OrcCamera { // Camera, assets/cl/camera.cl:7-26 (128 B)
    OrcFloat3 eyePoint, screenPoint, u, v, uNormalized, vNormalized;
    float focalDistance, apertureRadius, relativeAperture, shutterTime, ISO;
    uint8_t thinLensEnabled;
    uint8_t _pad[11];
};

struct OrcKernelData { // KernelData, assets/cl/kernel_data.cl:4-24 (176 B)
    OrcCamera camera;
    uint32_t numEmissiveTriangles, topLevelBvhRoot, rayOffset, scrWidth, scrHeight;
    uint32_t numInRays, numOutRays, numShadowRays, maxRays, newRays;
    uint8_t _pad[8];
};

struct OrcImage { // what oracle/_ref's read_imagef consumes: float RGBA texels [layer][y][x][4]
    int32_t width, height, layers, _pad;
    const float* rgba;
};

struct OrcScene { // same member order as RefScene in ref_build/ref_driver.cpp
    const void* vertices; // 48 B
    const void* triangles; // 16 B
    const void* subBvh; // 48 B
    const void* topBvh; // 112 B
    const void* emissive; // 96 B
    const void* materials; // 48 B
    const OrcImage* materialTextures;
    const OrcImage* skydomeTextures;
};

enum { ORC_RNG_LFSR113 = 0, ORC_RNG_COUNTER = 1 };
// which integrator `shade` runs (kernel.cl:248-279): neeIsShading, the one the reference compiles in; neeMisShading
// (shading.cl:35-349), reachable there under #define COMPARE_SHADING only; or exactly that build: neeMisShading for the
// pixels of the left half of the image, neeIsShading for the right half, both halves showing the left half's view
// (kernel.cl:48-51,248-265)
enum {
    ORC_INTEGRATOR_IS = 0,
    ORC_INTEGRATOR_MIS = 1,
    ORC_INTEGRATOR_COMPARE = 2,
    // the same two with neeMisShading's uninitialised read (shading.cl:590-592) evaluating to 0, as it does in oracle/_ref
    ORC_INTEGRATOR_MIS_AS_COMPILED = 3,
    ORC_INTEGRATOR_COMPARE_AS_COMPILED = 4
};
inline bool orcIsMis(uint32_t i) { return i == ORC_INTEGRATOR_MIS || i == ORC_INTEGRATOR_MIS_AS_COMPILED; }
inline bool orcIsCompare(uint32_t i) { return i == ORC_INTEGRATOR_COMPARE || i == ORC_INTEGRATOR_COMPARE_AS_COMPILED; }
// how NEE picks its light: randomPointOnLight (uniform over the emissive triangles, shading_helper.cl:261-278) or
// weightedRandomPointOnLight (proportional to the solid angle of each triangle seen from the shading point,
// shading_helper.cl:216-259; not called by the reference's kernels)
enum { ORC_LIGHTS_UNIFORM = 0, ORC_LIGHTS_SOLID_ANGLE = 1 };

struct OrcParams {
    uint32_t rngMode; // ORC_RNG_*
    uint32_t sample; // counter mode: sample index of this frame
    uint32_t seed; // counter mode
    uint32_t maxBounces; // 0 -> 4 (MAX_ITERATIONS, kernel.cl:4)
    uint32_t integrator; // ORC_INTEGRATOR_*
    uint32_t lightSampling; // ORC_LIGHTS_*
};

struct OrcCounters {
    uint64_t raysExtension, raysShadow, raysGenerated, shadeHits, deposits;
    uint64_t topVisits, innerSteps, triangleTests; // COUNT_TRAVERSAL-style (scene.cl:108-110,178-180,202-204)
};

// ---- RNG ---------------------------------------------------------------------------------
void orc_lfsr113_create_streams(uint32_t count, void* streams48); // src/lfsr113.c:183-290
float orc_lfsr113_u01(void* stream48); // private/lfsr113.c.h:61-89
float orc_counter_u01(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim, uint32_t seed);

// ---- kernels (serial, gid order) -----------------------------------------------------------
void orc_generatePrimaryRays(size_t global, OrcRayData* outRays, OrcKernelData* kd, void* streams, const OrcParams* p);
void orc_intersectWalk(size_t global, OrcShadingData* out, const OrcRayData* inRays, const OrcKernelData* kd, const OrcScene* s, OrcCounters* c);
void orc_shade(size_t global, OrcFloat3* outputPixels, OrcRayData* outRays, OrcRayData* outShadowRays, const OrcRayData* inRays,
    const OrcShadingData* inShading, OrcKernelData* kd, const OrcScene* s, void* streams, const OrcParams* p, OrcCounters* c);
void orc_intersectShadows(size_t global, OrcFloat3* outputPixels, OrcRayData* shadowRays, const OrcKernelData* kd, const OrcScene* s, OrcCounters* c);
void orc_updateKernelData(OrcKernelData* kd);
void orc_accumulate(uint32_t width, uint32_t height, float* outRgba, const OrcFloat3* input, const OrcKernelData* kd, uint32_t n);
int orc_trace_rays(OrcKernelData* kd, uint32_t maxRays, OrcRayData* rays0, OrcRayData* rays1, OrcRayData* shadow, OrcShadingData* shading,
    void* streams, OrcFloat3* accum, const OrcScene* s, const OrcParams* p, uint32_t* passTrace, int maxPasses, OrcCounters* c);

// ---- batch helpers for kernel-level parity with the HIP path (SoA in/out) -------------------
// traceRay over n rays (scene.cl:61-271). anyHit: tmax[i] = ray length, prim[i] = 1/0.
void orc_intersect_batch(const OrcScene* s, uint32_t topRoot, uint32_t n, const float* ox, const float* oy, const float* oz,
    const float* dx, const float* dy, const float* dz, const float* tmax, int anyHit, float* t, float* u, float* v,
    int32_t* prim, int32_t* inst, int threads, OrcCounters* c);

// Whole-image production-mode render (counter RNG), path by path, `threads` host threads:
// `spp` samples starting at sample index `firstSample` for pixels [0, W*H) restricted to `pixels`
// (nullptr = all).  Equivalent to spp x orc_trace_rays in ORC_RNG_COUNTER mode.  accum += sums.
void orc_render(const OrcKernelData* kd, const OrcScene* s, uint32_t firstSample, uint32_t spp, uint32_t seed, uint32_t maxBounces,
    const uint32_t* pixels, uint32_t numPixels, OrcFloat3* accum, int threads, OrcCounters* c);
// same with an integrator / light-sampling choice (ORC_INTEGRATOR_*, ORC_LIGHTS_*)
void orc_render_ex(const OrcKernelData* kd, const OrcScene* s, uint32_t firstSample, uint32_t spp, uint32_t seed, uint32_t maxBounces,
    uint32_t integrator, uint32_t lightSampling, const uint32_t* pixels, uint32_t numPixels, OrcFloat3* accum, int threads, OrcCounters* c);
// weightedRandomPointOnLight (shading_helper.cl:216-259) alone, for the test against the reference's compiled function
void orc_weighted_light(const OrcScene* s, uint32_t numEmissive, const float* x3, void* stream48, float* outPoint3, float* outNormal3,
    float* outColour3, float* outArea);
}
