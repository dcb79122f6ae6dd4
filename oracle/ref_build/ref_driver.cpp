// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
//
// Serial NDRange driver for the reference's OpenCL kernels compiled for the host CPU
// (see oracle/Makefile).  Each work-item is executed as a work-group of
// size 1 in increasing global-id order, which is a legal OpenCL schedule and makes the
// `atomic_inc` compaction order deterministic (= gid order): this is the oracle's canonical order.
//
// ref_trace_rays() restates the control flow of the reference's host loop
// RayTracer::traceRays (/root/reference/src/raytracer.cpp:289-430), which cannot be compiled
// here (it needs a live OpenCL device, GL and glm); it is ~40 lines of counter bookkeeping.
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {
extern thread_local size_t ref_global_id[3];
extern thread_local size_t ref_local_id[3];

// The reference kernels (kernel.cl:24,86,138,190,303) as plain C symbols in kernel.o.
void generatePrimaryRays(void* outRays, void* kernelData, void* streams);
void intersectShadows(void* outputPixels, void* shadowRays, void* stack, void* kernelData,
    void* vertices, void* triangles, void* subBvh, void* topBvh);
void intersectWalk(void* outShading, void* inRays, void* stack, void* kernelData,
    void* vertices, void* triangles, void* subBvh, void* topBvh, void* outputPixels);
void shade(void* outputPixels, void* outRays, void* outShadowRays, void* inRays, void* inShading,
    void* kernelData, void* vertices, void* triangles, void* emissive, void* materials,
    void* materialTextures, void* skydomeTextures, void* streams);
void updateKernelData(void* kernelData);

// helpers of the same translation unit, called directly by the test door below
typedef float ref_float3 __attribute__((ext_vector_type(3)));
int clrngLfsr113CopyOverStreamsFromGlobal(size_t count, void* destStreams, const void* srcHostStreams);
int clrngLfsr113CopyOverStreamsToGlobal(size_t count, void* destHostStreams, const void* srcStreams);
void weightedRandomPointOnLight(const void* scene, ref_float3 intersection, void* randomStream, ref_float3* outPoint, ref_float3* outLightNormal,
    ref_float3* outLightColour, float* outLightArea); // shading_helper.cl:216-259 (not called by any kernel)
}

namespace {
// KernelData (kernel_data.cl:4-24): Camera is 128 bytes, then 10 uints. Offsets from SURVEY 2.3.
struct KernelDataView {
    uint8_t camera[128];
    uint32_t numEmissiveTriangles, topLevelBvhRoot, rayOffset, scrWidth, scrHeight;
    uint32_t numInRays, numOutRays, numShadowRays, maxRays, newRays;
    uint8_t pad[8];
};
static_assert(sizeof(KernelDataView) == 176, "KernelData is 176 bytes");

inline size_t roundUp(size_t n, size_t m) { return (n + m - 1) / m * m; }

template <typename F>
void ndrange(size_t global, F&& f)
{
    for (size_t gid = 0; gid < global; gid++) {
        ref_global_id[0] = gid;
        ref_global_id[1] = ref_global_id[2] = 0;
        ref_local_id[0] = ref_local_id[1] = ref_local_id[2] = 0;
        f();
    }
}
}

extern "C" {

struct RefScene {
    void* vertices; // VertexData[ ] 48 B
    void* triangles; // TriangleData[ ] 16 B
    void* subBvh; // SubBvhNode[ ] 48 B
    void* topBvh; // TopBvhNode[ ] 112 B
    void* emissive; // EmissiveTriangle[ ] 96 B
    void* materials; // Material[ ] 48 B
    void* materialTextures; // RefImage*
    void* skydomeTextures; // RefImage*
};

void ref_generatePrimaryRays(size_t global, void* outRays, void* kd, void* streams)
{
    ndrange(global, [&] { generatePrimaryRays(outRays, kd, streams); });
}

void ref_intersectWalk(size_t global, void* outShading, void* inRays, void* stack, void* kd, const RefScene* s)
{
    ndrange(global, [&] {
        intersectWalk(outShading, inRays, stack, kd, s->vertices, s->triangles, s->subBvh, s->topBvh, nullptr);
    });
}

void ref_intersectShadows(size_t global, void* outputPixels, void* shadowRays, void* stack, void* kd, const RefScene* s)
{
    ndrange(global, [&] {
        intersectShadows(outputPixels, shadowRays, stack, kd, s->vertices, s->triangles, s->subBvh, s->topBvh);
    });
}

void ref_shade(size_t global, void* outputPixels, void* outRays, void* outShadowRays, void* inRays,
    void* inShading, void* kd, const RefScene* s, void* streams)
{
    ndrange(global, [&] {
        shade(outputPixels, outRays, outShadowRays, inRays, inShading, kd, s->vertices, s->triangles,
            s->emissive, s->materials, s->materialTextures, s->skydomeTextures, streams);
    });
}

void ref_updateKernelData(void* kd)
{
    ndrange(1, [&] { updateKernelData(kd); });
}

// One sample per pixel: restatement of RayTracer::traceRays (raytracer.cpp:289-430).
// kd: 176-byte KernelData with camera/numEmissive/topRoot/scrWidth/scrHeight already filled.
// rays0/rays1/shadow: maxRays*80 B; shading: maxRays*32 B; stack: maxRays*32 uints;
// streams: W*H*48 B; accum: W*H float4.  passTrace (optional): per pass
// {numInRays, newRays, rayOffset, numOutRays(after shade)}, up to maxPasses entries.
int ref_trace_rays(void* kdv, uint32_t maxRays, void* rays0, void* rays1, void* shadow, void* shading,
    void* stack, void* streams, void* accum, const RefScene* s, uint32_t* passTrace, int maxPasses)
{
    KernelDataView* kd = (KernelDataView*)kdv;
    kd->rayOffset = 0; // raytracer.cpp:303-311
    kd->numInRays = 0;
    kd->numOutRays = 0;
    kd->numShadowRays = 0;
    kd->maxRays = maxRays;
    kd->newRays = 0;

    void* rays[2] = { rays0, rays1 };
    int in = 0, out = 1;
    uint32_t surviving = 0;
    int pass = 0;
    while (true) {
        if (surviving != maxRays) // :324
            ref_generatePrimaryRays(roundUp(maxRays - surviving, 32), rays[in], kd, streams);
        ref_intersectWalk(maxRays, shading, rays[in], stack, kd, s); // :338-354
        uint32_t inRays = kd->numInRays, newRays = kd->newRays, rayOffset = kd->rayOffset;
        ref_shade(maxRays, accum, rays[out], shadow, rays[in], shading, kd, s, streams); // :357-378
        surviving = kd->numOutRays; // :381-389 (blocking read-back)
        if (passTrace && pass < maxPasses) {
            passTrace[pass * 4 + 0] = inRays;
            passTrace[pass * 4 + 1] = newRays;
            passTrace[pass * 4 + 2] = rayOffset;
            passTrace[pass * 4 + 3] = surviving;
        }
        pass++;
        uint32_t total = kd->scrWidth * kd->scrHeight;
        if (surviving == 0 && kd->rayOffset + kd->newRays >= total) // :392-395
            break;
        if (surviving != 0) // :398-414
            ref_intersectShadows(roundUp(surviving, 64), accum, shadow, stack, kd, s);
        ref_updateKernelData(kd); // :417-423
        int t = in; // :426
        in = out;
        out = t;
    }
    return pass;
}

// weightedRandomPointOnLight as the reference compiled it, on a Scene record filled like loadScene does (scene.cl:12-59).
// hostStream48: a clrngLfsr113HostStream, advanced by the call (1 + 2 draws).
void ref_test_weighted_light(const RefScene* s, uint32_t numEmissive, const float* x3, void* hostStream48, float* outPoint3,
    float* outNormal3, float* outColour3, float* outArea)
{
    struct SceneRecord { // Scene, scene.cl:12-28 (96 bytes)
        uint32_t numVertices, numTriangles, numEmissiveTriangles, numLights;
        const void *vertices, *triangles, *meshMaterials, *emissiveTriangles, *subBvh, *topLevelBvh;
        uint32_t topLevelBvhRoot;
        float refractiveIndex;
        int32_t cubemapTextureIndices[6];
    } scene {};
    static_assert(sizeof(SceneRecord) == 96, "Scene is 96 bytes");
    scene.numEmissiveTriangles = numEmissive;
    scene.vertices = s->vertices, scene.triangles = s->triangles, scene.meshMaterials = s->materials;
    scene.emissiveTriangles = s->emissive, scene.subBvh = s->subBvh, scene.topLevelBvh = s->topBvh;
    scene.refractiveIndex = 1.000277f;
    alignas(16) unsigned char stream[64] = {}; // clrngLfsr113Stream: current state + pointer to the initial one
    clrngLfsr113CopyOverStreamsFromGlobal(1, stream, hostStream48);
    ref_float3 X = { x3[0], x3[1], x3[2] }, p, n, c;
    float area = 0;
    weightedRandomPointOnLight(&scene, X, stream, &p, &n, &c, &area);
    clrngLfsr113CopyOverStreamsToGlobal(1, hostStream48, stream);
    for (int k = 0; k < 3; k++)
        outPoint3[k] = p[k], outNormal3[k] = n[k], outColour3[k] = c[k];
    *outArea = area;
}

} // extern "C"
