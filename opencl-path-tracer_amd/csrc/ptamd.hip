// C-ABI implementation (include/ptamd.h) of the gfx950 ray-queue render path: context, scene
// conversion into the HBM layouts of pt_device.h, the launch schedule that replaces the reference's
// RayTracer::traceRays (src/raytracer.cpp:289-430), and the kernel-granular test hooks.
#include "../../include/ptamd.h"
#include "../host/parallel.h" // (header only: the worker threads the host library uses, here for the conversion of an upload)
#include "pt_shade.h"
#include "pt_trace.h"
#include "pt_packet.h"
#include "pt_packet_multi.h"
#include "pt_bake.h"
#include "pt_team.h"
#ifndef PT_PACK_WIDE
#define PT_PACK_WIDE 1
#endif
#include <algorithm>
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h> // only for the two enumerators pt_reduce_accum passes (the library itself is bound at run time, below)
#define PT_HAVE_RCCL_HEADER 1
#endif
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace ptd;

namespace {

thread_local std::string g_createError;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0)
            return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

struct RayQueueBuf {
    DevBuf<float4> o, d, thr;
    RayQueue view() const { return { o.p, d.p, thr.p }; }
};
struct ShadowQueueBuf {
    DevBuf<float4> o, d, c;
    ShadowQueue view() const { return { o.p, d.p, c.p }; }
};

} // namespace

// Everything pt_upload_static derives from the caller's static arrays -- the host's mirrors and the device's master copies.  A context holds TWO
// (like the dynamic sets, like the reference's double-buffered cl::Buffers): renders and refits work on the current one while
// pt_upload_static_async converts a rebuilt scene into the other; pt_frame_tick adopts it together with the dynamic state built on it.
struct StaticScene {
    // The static part: the bottom-level trees as packed 4-wide nodes, object-space triangles and shading records.  Converted on the
    // host once per pt_upload_static / pt_update_geometry (buildStaticGeom); one master copy on the device, from which a dynamic
    // set refreshes its own copy (device to device) when its version is stale.
    struct StaticGeom {
        std::vector<WideNode> wide; // packed, object space
        std::vector<WideBoxes> boxes; // exact child boxes of every packed node
        std::vector<uint32_t> leafOfs; // [node][child]: offset of a leaf child's first triangle reference in its mesh's run
        std::vector<uint32_t> refTri; // triangle references in leaf order, mesh by mesh -> caller's triangle index
        std::vector<uint32_t> stackNeed; // per packed node
        // what a REFIT needs of the conversion and cannot change: which pair-node child the box of every packed child slot is, which
        // slots are unused (the collapse's split choices and the packing order stay as they are)
        std::vector<uint32_t> kidSrc; // [node][child]: (pair node << 1) | side
        std::vector<uint8_t> kidEmpty; // [node][child]
        std::vector<uint32_t> kidBoxNode; // [node][child]: the same as a caller's node index (k_refit_nodes, pt_bake.h), 0x80000000 | i: extra box i, ~0: unused
        std::vector<TriFat> fat;
        struct Root {
            uint32_t ref; // device reference of the mesh root (a packed node, or a leaf)
            uint32_t nodeBase, numNodes, refBase, numRefs;
            bool bakeable; // its nodes are one run of their own
        };
        std::vector<Root> roots;
        std::vector<int32_t> rootOfNode; // caller's node index -> roots[] slot, -1: not a root
        std::vector<uint32_t> extraRoots; // interior nodes a top-level leaf has named
        uint32_t emptyRef = 0;
        uint64_t version = 0;
        bool onDevice = false;
        DevBuf<WideNode> dWide;
        DevBuf<WideBoxes> dBoxes;
        DevBuf<uint32_t> dLeafOfs, dRefTri;
        DevBuf<TriIsect> dTris;
        DevBuf<TriFat> dFat;
        // refit (pt_update_geometry): the caller's vertices on the device (the triangles' intersection and shading records are re-made
        // from them by k_refit_tris), pinned staging for them and for the re-quantised nodes, guarded by an event of its own
        DevBuf<pt_vertex> dVerts;
        DevBuf<pt_sub_bvh_node> dNodes; // the caller's nodes as last handed in
        DevBuf<uint32_t> dKidBoxNode;
        DevBuf<float> dExtra;
        // refit on the device alone (pt_refit_vertices): who a packed node reports to and how many arrivals complete it (k_refit_tree, pt_bake.h)
        DevBuf<uint32_t> dParent, dNeed, dArrived;
        uint64_t refitTablesFor = 0; // topology the tables were made for (0: none)
        bool refitTablesOk = false; // false: a node has two parents (roots that share a subtree): the caller refits on the host (pt_update_geometry)
        uint64_t topology = 0; // bumped by every buildStaticGeom
        bool latestInStage = false; // the caller's latest vertices and nodes live in `stage` (vertices first), not in rawVerts / hostSubNodes
        void* stage = nullptr;
        size_t stageBytes = 0;
        hipEvent_t stageRead = nullptr;
        bool stageBusy = false;
    } sg;
    DevBuf<TriShade> triShade;
    DevBuf<Material> materials;
    std::vector<VertexShade> hostVerts;
    std::vector<pt_vertex> rawVerts; // the caller's vertices as last handed in (pt_upload_static / pt_update_geometry)
    std::vector<uint32_t> denseOfNode; // caller's sub-BVH node -> pair node (0xFFFFFFFF: a leaf or a pad)
    uint32_t numDensePairs = 0; // pair nodes [0, numDensePairs) mirror the caller's inner nodes; the rest split leaves of more than kMaxLeafTris
    std::vector<TriIsect> hostTris; // object-space intersection triangles (world-space copies of tiny instances are appended per pt_upload_dynamic)
    std::vector<PairNode> hostBottomNodes; // bottom-level pair nodes (the top level is appended per pt_upload_dynamic)
    std::vector<uint32_t> nodeRef; // reference sub-BVH node index -> device child reference
    std::vector<uint32_t> subtreeDepth; // per reference node (roots queried)
    std::vector<TriShade> hostTriShade; // vertex indices + material of every triangle (kept for pt_update_geometry)
    std::vector<pt_material> hostMaterials;
    std::vector<pt_sub_bvh_node> hostSubNodes; // the caller's sub-BVH as uploaded (topology; boxes are replaced by pt_update_geometry)
    uint32_t numVerts = 0;
    uint32_t numRefNodes = 0, numTris = 0;
    bool hostNodeBoxesStale = false; // the boxes in hostSubNodes are older than rawVerts (pt_refit_vertices: the device refitted its own tree, nobody handed nodes in)
    bool hostGeomStale = false; // hostTris / hostVerts / hostBottomNodes' boxes / sg.wide / sg.boxes / sg.fat are older than the caller's latest arrays (a refit
                                // re-makes the device's copies on the device only; the host's are refreshed if the whole conversion ever runs again)
    bool materialBins = false; // the surfaces are of more than one material type: k_shade shades its tiles in material order
    bool have = false; // holds a converted scene
};

struct pt_ctx {
    pt_config cfg {};
    std::string error;
    int device = 0;
    int numCUs = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    std::vector<hipEvent_t> profEvents;
    bool profile = false;

    // scene (HBM)
    DevBuf<PairNode> nodes;
    DevBuf<WideNode> wide;
    DevBuf<VertexShade> verts;
    // The dynamic part of the scene -- what pt_upload_dynamic(_async) produces: 4-wide nodes of both levels, intersection
    // triangles (object space + world-space copies of instances), instances, lights -- exists TWICE, like the reference's
    // double-buffered cl::Buffers (m_topBvhBuffers[2], m_emissiveTrianglesBuffers[2], ... src/raytracer.h:93-106): renders
    // enqueued so far keep reading set `active` while the next state is converted on the host and copied into the other set on
    // the copy stream; pt_frame_tick makes the render stream wait for that copy and flips (RayTracer::frameTick,
    // src/raytracer.cpp:183-189; the barrier of :593).
    struct DynamicSet {
        DevBuf<WideNode> wide;
        DevBuf<TriIsect> tris;
        DevBuf<TriFat> fat; // shading records: they hold v0 / edges / normals, which a refitted mesh changes with the trees
        DevBuf<Instance> instances;
        DevBuf<Light> lights;
        DevBuf<BakeJob> jobs; // world-space copies to make (pt_bake.h)
        uint64_t staticVersion = 0; // version of the static arrays this set holds (0: none)
        int staticIndex = 0; // which of the context's two static scenes this state was built on
        uint32_t numTris = 0, firstWorldNode = 0; // of that scene, as the kernels need them (SceneDev)
        // pinned staging the asynchronous copies read from (grow-only, like the device buffers)
        void* stage = nullptr;
        size_t stageBytes = 0;
        hipEvent_t stageRead = nullptr; // recorded on the copy stream after the copies out of `stage`
        bool stageBusy = false;
        uint32_t numLights = 0, rootRef = 0;
        uint32_t foldedInstances = 0, instRootBase = 0, numInstRoots = 0;
        DevBuf<float4> instFold; // the table of folded instance transforms (pt_trace.h)
        DevBuf<uint32_t> instRootSrc;
        uint32_t instFoldCount = 0;
        uint32_t rootRefFolded = 0; // the same top level for the per-ray kernels: entry nodes in place of the instances that are a translation + uniform scale (pt_trace.h)
        bool packetOk = false;
        uint32_t stackNeed = 0; // worst-case traversal stack of this state (pt_stats.stack_need)
        bool hasInstances = false; // the tree holds instance references (instances that were not copied to world space)
        bool generalRoute = false; // ... and the per-ray kernels enter them as leaf-kind steps (pt_trace.h, LEVELS 2): some transform is not a translation + uniform scale, or there are more than the fold table holds
        uint32_t enteredInstances = 0; // instances that are entered at traversal (not copied to world space)
        std::vector<uint32_t> instanceTopNode; // instance index -> top-level leaf node index
        hipEvent_t uploaded = nullptr; // recorded on the copy stream after the set's last upload
        hipEvent_t lastUse = nullptr; // recorded on the render stream when the set stopped being the active one
        bool used = false;
    } dyn[2];
    StaticScene stat[2];
    StaticScene* st = &stat[0]; // the static scene the entry points work on: the current one, except while pt_upload_static_async converts the other
    int statCur = 0; // static scene of the active dynamic set
    int statPending = -1; // converted by pt_upload_static_async, waiting for a dynamic state and pt_frame_tick
    uint64_t staticVersions = 0; // versions of the static arrays are drawn from one counter (a dynamic set compares the one it holds with the scene's)
    int active = 0; // set the render kernels read
    int pending = -1; // set with an upload in flight / finished that pt_frame_tick will switch to
    hipStream_t copyStream = nullptr;
    // small launches (a 1-spp interactive frame): the shadow rays of bounce b are traced on `sideStream` while the main stream traces
    // the extension rays of bounce b + 1 (independent: both only need shade b; shade b + 1 waits for both)
    hipStream_t sideStream = nullptr;
    hipStream_t sideStream2 = nullptr; // one sample in flight: the shadow passes of a frame alternate between two side streams (each bounce has its own shadow queue AND
                                       // accumulator plane: nothing orders them but their own shade launch)
    hipEvent_t evShaded[kMaxPasses] = {}, evShadowed[kMaxPasses] = {};
    // ... and, with ONE sample in flight, deposit into an accumulator of their own (merged into the accumulator proper at the end of pt_render) from a shadow
    // queue per bounce: the shadow passes then depend on nothing but their own shade launch and run back to back on the side stream
    DevBuf<float4> accumShadow;
    ShadowQueueBuf shadowQ[kMaxPasses];
    bool mergePending = false;
    DevBuf<uint8_t> texMaterial, texSky; // float4 or BGRA8 texels (Texture::format)
    SceneDev scene {};
    bool haveStatic = false, haveDynamic = false, haveCamera = false;

    // frame state
    CameraDev camera {};
    DevBuf<uint32_t> pixelList;
    std::vector<uint32_t> hostPixelList; // what pixelList holds (pt_set_tiles with the same list again is a no-op)
    DevBuf<uint32_t> pixelOrdinal; // global pixel -> position in pixelList (only when the context owns part of the frame)
    DevBuf<float4> resolveTmp; // pt_resolve's output staging (allocated at first use)
    uint32_t numOwned = 0;
    uint32_t capacity = 0;
    // Queues smaller than a batch (pt_config.ext_queue_fraction / shadow_queue_fraction, round 6): entries the second extension queue (and, for batches of a
    // pinhole's bundles, the origin / throughput planes of the first) and the shadow queue hold; == capacity without fractions.  A batch is sized so that what its
    // FIRST pass emits fits (every later pass emits at most what it was handed): from the largest ratios seen in this epoch (camera, scene state, tiling).
    uint32_t capExt = 0, capShadow = 0;
    bool q0Small = false; // the first queue's origin / throughput planes hold capExt entries (camera rays queued as directions only)
    bool ratiosKnown = false;
    double ratioExt = 0, ratioShadow = 0; // (rays emitted by pass 0) / (entries of the batch), the largest of this epoch
    uint32_t* overflowPinned = nullptr; // set by k_clamp_counts when a batch emitted more than a queue holds after all: sticky, reported by pt_synchronize and the image reads
    uint32_t batchSamples = 0, probeBatches = 0;
    uint32_t epoch = 0, passCountsEpoch = 0; // camera / scene state / tiling the ratios belong to; ... the report in flight was launched in
    bool identityPixels = true;
    DevBuf<float4> accumOwn, accumPlanes;
    uint32_t packetBlocks[2] = { 0, 0 };
    uint32_t multiBlocks[4] = { 0, 0, 0, 0 }; // persistent grid of k_trace_multi [without | with instance references in the tree], [2], [3]: the same for a thin-lens camera's converging bundles
    // live entries per pass of the most recent batch whose counters have come back (a HINT for the next batch's k_shade launches:
    // copied to pinned memory by the stream at the end of every batch, never waited for)
    uint32_t* passCountsPinned = nullptr; // kMaxPasses + 1 words
    hipEvent_t passCountsCopied = nullptr;
    uint32_t passCountsHint[2 * (kMaxPasses + 1)] = {}; // extension rays per pass, then shadow rays per pass
    uint32_t passCountsEntries = 0; // entries of the batch the hint comes from (0: no hint yet)
    uint32_t passCountsPending = 0; // entries of the batch whose copy is in flight
    uint32_t shadeHeadShift = 0; // diagnostics (PTAMD_SHADE_HEAD_SHIFT): shrinks the head of the split k_shade launches so that tests reach the tile-walking kernel
    uint32_t packetUse = 0; // bit 0: primary rays, bit 1: their shadow rays, bit 2: the pt_intersect test hook
    uint64_t packetLaunches = 0, genLaunches = 0, bundleLaunches = 0;
    float4* accum = nullptr;
    uint32_t planes = 1; // samples in flight (fixed schedule)
    uint32_t spp = 0;

    // queues
    RayQueueBuf rays[2], stagedRays;
    ShadowQueueBuf shadow, stagedShadow;
    DevBuf<float4> hitH;
    DevBuf<int32_t> hitInst;
    DevBuf<uint32_t> activeFlag;
    DevBuf<uint4> streams;
    DevBuf<Control> control;
    DevBuf<Totals> totals;
    DevBuf<uint32_t> spill;
    size_t spillHalf = 0;
    uint32_t traceBlocks[3] = { 0, 0, 0 }; // persistent grids: [0] one world-space tree, [1] trees with instance references (folded / parked), [2] the general route (pt_trace.h, LEVELS 2)
    uint32_t teamBlocks = 0; // grid of k_trace_team (pt_team.h: four lanes per ray, for launches that do not fill the machine)
    float teamRounds = 1.5f; // (1 / 1.5 / 1.7 / 2 / 3 measured on four scenes, tools/r5_frames_env.sh) ... used where the previous batch's pass held at most this many rays per team
    uint32_t teamUse = 7; // bit 0: the camera rays of 1-spp frames, bit 2: their shadow rays, bit 1: later passes by the previous batch's counters (PTAMD_TEAM_USE: diagnostics)
    uint32_t batchEntries = 0; // entries of the batch being enqueued (renderSampleFixed)
    uint64_t teamLaunches = 0;
    uint32_t foldPlanes = 0; // extra accumulator planes written since the last fold (folded at the end of pt_render)
    bool queuesReady = false;

    double msLastRender = 0, msIntersect = 0, msShade = 0, msShadow = 0, msGen = 0, msPacket = 0;
};

namespace {

int fail(pt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->error = buf;
    else
        g_createError = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(ctx, PT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// The bodies below build std::vectors and call std::function; nothing may escape across the C ABI, so every entry
// point that can allocate runs inside this guard and reports a failure like any other (the host library's capi.cpp
// does the same).
template <typename F>
int guarded(pt_ctx* c, const char* what, F&& body)
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(c, PT_ERR_UNSUPPORTED, "%s: out of host memory", what);
    } catch (const std::exception& e) {
        return fail(c, PT_ERR_INVALID, "%s: %s", what, e.what());
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "%s: unknown exception", what);
    }
}

inline AccumView accumView(const pt_ctx* c) { return { c->accum, c->accumPlanes.p, c->pixelOrdinal.p, c->planes - 1u }; }
inline uint32_t maxBounces(const pt_ctx* c) { return c->cfg.max_bounces ? c->cfg.max_bounces : 4u; }
inline bool parityMode(const pt_ctx* c) { return c->cfg.rng_mode == PT_RNG_LFSR113_PARITY; }
// anything but the integrator the reference compiles in (neeIsShading, uniform light choice) runs the general shading kernel
inline bool generalShading(const pt_ctx* c) { return (c->cfg.flags & (PT_FLAG_INTEGRATOR_MIS | PT_FLAG_COMPARE_SHADING | PT_FLAG_SOLID_ANGLE_LIGHTS)) != 0u; }

void refreshSceneView(pt_ctx* c)
{
    SceneDev& s = c->scene;
    s.nodes = c->nodes.p;
    const pt_ctx::DynamicSet& d = c->dyn[c->active];
    s.wide = d.wide.p;
    s.tris = d.tris.p;
    s.triFat = d.fat.p;
    s.materials = c->st->materials.p;
    s.instances = d.instances.p;
    s.lights = d.lights.p;
    s.numLights = d.numLights;
    s.rootRef = d.rootRef;
    s.firstWorldNode = d.firstWorldNode; // (of the static scene the active dynamic set was built on)
    s.instRootBase = d.numInstRoots ? d.instRootBase : 0x7FFFFFFFu; // (nothing folded: no node lies behind the world-space ones)
    s.numInstRoots = d.numInstRoots;
    s.materialTex.texels = c->texMaterial.p;
    s.sky.texels = c->texSky.p;
    s.numTriangles = c->haveDynamic ? d.numTris : c->st->numTris;
}

template <typename T>
int uploadVec(pt_ctx* c, DevBuf<T>& buf, const std::vector<T>& host)
{
    // grow-only: a rebuilt scene of about the old size reuses the old buffers (hipFree waits for the whole device -- a frame loop that rebuilds a
    // tree per frame, pt_upload_static_async, must not)
    if (!buf.p || buf.n < std::max<size_t>(host.size(), 1))
        HIPCHK(c, buf.alloc(std::max<size_t>(host.size() + host.size() / 8, 1)));
    if (!host.empty())
        HIPCHK(c, hipMemcpy(buf.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

// clRNG stream spacing jump (published xor/shift network of lfsr113AdvanceState, clRNG
// src/lfsr113.c:183-240): stream k+1 starts 2^55 steps after stream k.
void lfsrJump(uint32_t g[4])
{
    uint32_t z, b; // (the library computes in int: its left shifts overflow -- undefined in C++, found by UBSan; every right shift is masked down to
                   // the bits a logical shift yields, so unsigned arithmetic gives the same words)
    z = g[0] & (uint32_t)(-2);
    b = (z << 6) ^ z;
    z = (z) ^ (z << 2) ^ (z << 3) ^ (z << 10) ^ (z << 13) ^ (z << 16) ^ (z << 19) ^ (z << 22) ^ (z << 25) ^ (z << 27) ^ (z << 28)
        ^ ((b >> 3) & 0x1FFFFFFF) ^ ((b >> 4) & 0x0FFFFFFF) ^ ((b >> 6) & 0x03FFFFFF) ^ ((b >> 9) & 0x007FFFFF)
        ^ ((b >> 12) & 0x000FFFFF) ^ ((b >> 15) & 0x0001FFFF) ^ ((b >> 18) & 0x00003FFF) ^ ((b >> 21) & 0x000007FF);
    g[0] = z;
    z = g[1] & (uint32_t)(-8);
    b = (z << 2) ^ z;
    z = ((b >> 13) & 0x0007FFFF) ^ (z << 16);
    g[1] = z;
    z = g[2] & (uint32_t)(-16);
    b = (z << 13) ^ z;
    z = (z << 2) ^ (z << 4) ^ (z << 10) ^ (z << 12) ^ (z << 13) ^ (z << 17) ^ (z << 25)
        ^ ((b >> 3) & 0x1FFFFFFF) ^ ((b >> 11) & 0x001FFFFF) ^ ((b >> 15) & 0x0001FFFF) ^ ((b >> 16) & 0x0000FFFF) ^ ((b >> 24) & 0x000000FF);
    g[2] = z;
    z = g[3] & (uint32_t)(-128);
    b = (z << 3) ^ z;
    z = (z << 9) ^ (z << 10) ^ (z << 11) ^ (z << 14) ^ (z << 16) ^ (z << 18) ^ (z << 23) ^ (z << 24)
        ^ ((b >> 1) & 0x7FFFFFFF) ^ ((b >> 2) & 0x3FFFFFFF) ^ ((b >> 7) & 0x01FFFFFF) ^ ((b >> 9) & 0x007FFFFF)
        ^ ((b >> 11) & 0x001FFFFF) ^ ((b >> 14) & 0x0003FFFF) ^ ((b >> 15) & 0x0001FFFF) ^ ((b >> 16) & 0x0000FFFF)
        ^ ((b >> 23) & 0x000001FF) ^ ((b >> 24) & 0x000000FF);
    g[3] = z;
}

// Collapse the pair-node tree into 4-wide nodes (pt_device.h, WideNode): which descendants of pair node i become the (up to four)
// children of its wide node.  kids[i] describes the same subtree as pair[i], so child references keep their indices; the boxes are the
// exact ones (quantiseWideNode, pt_bake.h, makes the 8-bit planes; the world-space copies of instances are re-fitted from the exact boxes).
#ifndef PT_COLLAPSE_OPTIMAL
#define PT_COLLAPSE_OPTIMAL 1
#endif
struct WideKids {
    float lo[4][3], hi[4][3];
    uint32_t ref[4];
    uint32_t src[4]; // where the box of child k comes from: (pair node << 1) | side -- what a refit re-reads (refitStaticGeom)
    bool empty[4];
};
// Leaf formation inside the collapse (round 5).  The reference's builders stop at <= 3 triangles per leaf with Ct 1.5 / Ci 1.0 tuned for a binary tree
// (src/bvh/bvh_build.cpp:15-18); for THIS traversal a visit of a 4-wide node costs ~105 vector instructions and a triangle test ~35, and a leaf step runs
// to the longest leaf among its lanes.  So the collapse may turn a whole subtree into ONE leaf where that is cheaper:
//   asLeaf[n] = area(n) * (leaf0 + tri * (alpha * tris(n) + (1 - alpha) * cap))      (alpha 1: cost per triangle; alpha 0: every leaf visit costs the cap)
//   asRoot[n] = area(n) * inner + cheapest distribution of its (up to four) slots
// possible only where the subtree's triangle references are one contiguous run (the reference's builders emit leaves depth first: always) of <= cap.
// cap 0 = the leaves are given (rounds 1-4).  PTAMD_LEAF_FORMATION="cap[,inner,leaf0,tri,alpha]" overrides at run time (sweeps).
#ifndef PT_LEAF_CAP
#define PT_LEAF_CAP 0
#endif
struct CollapseCosts {
    uint32_t cap = PT_LEAF_CAP;
    double inner = 105.0, leaf0 = 20.0, tri = 35.0, alpha = 1.0;
    double leaf(uint32_t n) const { return leaf0 + tri * (alpha * (double)n + (1.0 - alpha) * (double)std::max(cap, 1u)); }
};
CollapseCosts collapseCostsFromEnv()
{
    CollapseCosts k;
    if (const char* e = getenv("PTAMD_LEAF_FORMATION")) {
        unsigned cap = k.cap;
        const int got = sscanf(e, "%u,%lf,%lf,%lf,%lf", &cap, &k.inner, &k.leaf0, &k.tri, &k.alpha);
        if (got >= 1)
            k.cap = std::min(cap, kMaxLeafTris);
    }
    return k;
}

std::vector<WideKids> collapseKids(const std::vector<PairNode>& pair, const CollapseCosts costs = CollapseCosts { 0u })
{
    std::vector<WideKids> out(pair.size());
    struct Child {
        float lo[3], hi[3];
        uint32_t ref;
        uint32_t src;
    };
    auto childOf = [&pair](const PairNode& n, int side) {
        Child c;
        c.src = ((uint32_t)(&n - pair.data()) << 1) | (uint32_t)side;
        const float* bx = &n.bx.x;
        const float* by = &n.by.x;
        const float* bz = &n.bz.x;
        c.lo[0] = bx[side * 2], c.hi[0] = bx[side * 2 + 1];
        c.lo[1] = by[side * 2], c.hi[1] = by[side * 2 + 1];
        c.lo[2] = bz[side * 2], c.hi[2] = bz[side * 2 + 1];
        c.ref = side ? n.right : n.left;
        return c;
    };
    auto area = [](const Child& c) {
        const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
        return dx >= 0.f && dy >= 0.f && dz >= 0.f ? dx * dy + dy * dz + dz * dx : -1.f;
    };
    // Which descendants become the (up to four) children of the wide node made from pair node i?  The cost of a wide tree is the
    // sum over its inner nodes of the chance a ray visits them ~ their surface area (the leaves are given).  Minimised exactly by
    // dynamic programming over the binary tree (as in Ylitie et al. 2017 for 8-wide trees):
    //   asRoot[n]   = area(n) + min over i of  atMost[left][i] + atMost[right][4 - i]          (n becomes a wide node)
    //   atMost[n][k] = cheapest way to hand subtree n to a parent that has k child slots for it:
    //                  n itself as one child (asRoot[n]), or split between its two children (i and k - i slots)
    // Round 1 opened the child of largest area until four were collected (surface-area greedy): 3 % more inner-node area on the
    // benchmark's meshes (17.67 vs 17.12 / 16.49 vs 16.02 root areas).
#if PT_COLLAPSE_OPTIMAL
    const size_t N = pair.size();
    auto isInner = [&](uint32_t r) { return r != kRefNone && refCount(r) == 0u && refIndex(r) < N; };
    struct Dp {
        double atMost[5]; // [1..4]
        uint8_t split[5]; // 0: the node itself, i: i slots to the left child
        uint8_t rootSplit, done;
        uint8_t asLeaf; // as ONE child the subtree is a leaf of [leafFirst, leafFirst + leafCount)
        uint32_t leafFirst, leafCount; // the subtree's triangle references, when they are one run of <= cap (leafCount 0: not)
    };
    const bool leafCosts = costs.cap > 0u; // the leaves are no longer given: they enter the cost
    auto childArea = [&](const PairNode& n, int side) {
        const Child c = childOf(n, side);
        const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
        return dx >= 0.0 && dy >= 0.0 && dz >= 0.0 ? dx * dy + dy * dz + dz * dx : 0.0;
    };
    std::vector<Dp> dp(N);
    for (Dp& d : dp)
        d.done = 0;
    auto nodeArea = [&](size_t n) { // box of pair node n = union of its two child boxes
        const Child a = childOf(pair[n], 0), b = childOf(pair[n], 1);
        double lo[3], hi[3];
        bool any = false;
        for (const Child* c : { &a, &b }) {
            if (!(c->lo[0] <= c->hi[0]) || c->ref == kRefNone)
                continue;
            for (int ax = 0; ax < 3; ax++) {
                lo[ax] = any ? std::min(lo[ax], (double)c->lo[ax]) : c->lo[ax];
                hi[ax] = any ? std::max(hi[ax], (double)c->hi[ax]) : c->hi[ax];
            }
            any = true;
        }
        if (!any)
            return 0.0;
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    {
        // post-order over the subtree below `root` (a stack of its own per caller: subtrees are disjoint, so several can be solved side by side)
        auto solve = [&](size_t root, std::vector<uint32_t>& stack) {
            if (dp[root].done)
                return;
            stack.push_back((uint32_t)root);
            while (!stack.empty()) {
                const uint32_t n = stack.back();
                if (dp[n].done == 2) {
                    stack.pop_back();
                    continue;
                }
                const uint32_t kids[2] = { pair[n].left, pair[n].right };
                if (dp[n].done == 0) { // first visit: children first (done = 1 marks 'on the stack': a cycle cannot loop forever)
                    dp[n].done = 1;
                    for (uint32_t r : kids)
                        if (isInner(r) && dp[refIndex(r)].done == 0)
                            stack.push_back(refIndex(r));
                    continue;
                }
                // children are final (or n sits on a cycle, which upload validation has already excluded): combine
                auto costSide = [&](int side, int k) {
                    const uint32_t r = kids[side];
                    if (isInner(r) && dp[refIndex(r)].done == 2)
                        return dp[refIndex(r)].atMost[k];
                    if (leafCosts && r != kRefNone && refCount(r) >= 1u && refCount(r) <= kMaxLeafTris) // a given leaf: what a visit of it costs
                        return costs.leaf(refCount(r)) * childArea(pair[n], side);
                    return 0.0;
                };
                Dp& d = dp[n];
                // the subtree's triangle references as one run?
                d.asLeaf = 0, d.leafFirst = 0, d.leafCount = 0;
                if (leafCosts) {
                    uint32_t first[2] = { 0, 0 }, cnt[2] = { 0, 0 };
                    for (int side = 0; side < 2; side++) {
                        const uint32_t r = kids[side];
                        if (isInner(r) && dp[refIndex(r)].done == 2)
                            first[side] = dp[refIndex(r)].leafFirst, cnt[side] = dp[refIndex(r)].leafCount;
                        else if (r != kRefNone && refCount(r) >= 1u && refCount(r) <= kMaxLeafTris)
                            first[side] = refIndex(r), cnt[side] = refCount(r);
                    }
                    if (cnt[0] && cnt[1] && cnt[0] + cnt[1] <= costs.cap && (first[0] + cnt[0] == first[1] || first[1] + cnt[1] == first[0]))
                        d.leafFirst = std::min(first[0], first[1]), d.leafCount = cnt[0] + cnt[1];
                }
                double best = 1e300;
                for (int i = 1; i <= 3; i++) {
                    const double v = costSide(0, i) + costSide(1, 4 - i);
                    if (v < best)
                        best = v, d.rootSplit = (uint8_t)i;
                }
                d.atMost[1] = (leafCosts ? costs.inner : 1.0) * nodeArea(n) + best;
                if (d.leafCount) {
                    const double asLeaf = costs.leaf(d.leafCount) * nodeArea(n);
                    if (asLeaf < d.atMost[1])
                        d.atMost[1] = asLeaf, d.asLeaf = 1;
                }
                d.split[1] = 0;
                for (int k = 2; k <= 4; k++) {
                    d.atMost[k] = d.atMost[1];
                    d.split[k] = 0;
                    for (int i = 1; i < k; i++) {
                        const double v = costSide(0, i) + costSide(1, k - i);
                        if (v < d.atMost[k])
                            d.atMost[k] = v, d.split[k] = (uint8_t)i;
                    }
                }
                d.done = 2;
                stack.pop_back();
            }
        };
        // Round 6 (a rebuilt tree per frame: this pass was 0.67 of the 2.4 ms a 20 k-triangle scene spends in pt_upload_static_async): the subtrees five levels
        // below the roots of large trees are solved on the host library's worker pool, the tops on the calling thread afterwards.  The recurrence has one
        // solution per node whatever the order: the same tree, byte for byte.
        std::vector<uint32_t> tasks;
        const char* seq = getenv("PTAMD_BUILD_THREADS"); // (1: everything on the calling thread, as the host library's builders read it -- tests compare the two)
        if (N >= 4096 && !(seq && atoi(seq) == 1)) {
            std::vector<uint8_t> isChild(N, 0);
            for (size_t n = 0; n < N; n++)
                for (uint32_t r : { pair[n].left, pair[n].right })
                    if (isInner(r))
                        isChild[refIndex(r)] = 1;
            std::vector<uint32_t> level, next;
            for (size_t n = 0; n < N; n++)
                if (!isChild[n])
                    level.push_back((uint32_t)n);
            for (int depth = 0; depth < 5 && !level.empty() && level.size() < 64; depth++) {
                next.clear();
                for (uint32_t n : level)
                    for (uint32_t r : { pair[n].left, pair[n].right })
                        if (isInner(r) && refIndex(r) != n)
                            next.push_back(refIndex(r));
                level.swap(next);
            }
            std::sort(level.begin(), level.end());
            level.erase(std::unique(level.begin(), level.end()), level.end()); // (a shared subtree -- refused by the upload's validation anyway -- is solved once)
            tasks = level;
        }
        if (tasks.size() >= 2)
            raytracer::WorkerPool::get().parallelFor(tasks.size(), 1, [&](size_t t0, size_t t1) {
                std::vector<uint32_t> stack;
                for (size_t t = t0; t < t1; t++)
                    solve(tasks[t], stack);
            });
        std::vector<uint32_t> stack;
        for (size_t root = 0; root < N; root++)
            solve(root, stack);
    }
#endif
    raytracer::WorkerPool::get().parallelFor(pair.size(), 2048, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
        Child kids[4];
        int n = 0;
#if PT_COLLAPSE_OPTIMAL
        {
            struct Item {
                Child c;
                int slots;
            };
            Item todo[8];
            int nt = 0;
            const int ls = dp[i].rootSplit;
            todo[nt++] = { childOf(pair[i], 1), 4 - ls }; // right first: the stack pops the left one first, slot order = tree order
            todo[nt++] = { childOf(pair[i], 0), ls };
            while (nt > 0) {
                const Item it = todo[--nt];
                const uint32_t r = it.c.ref;
                const int sp = isInner(r) && refIndex(r) != i ? dp[refIndex(r)].split[it.slots] : 0;
                if (sp == 0) {
                    kids[n] = it.c;
                    if (isInner(r) && refIndex(r) != i && dp[refIndex(r)].asLeaf) // the whole subtree as ONE leaf: same box, its run of triangles
                        kids[n].ref = makeRef(dp[refIndex(r)].leafFirst, dp[refIndex(r)].leafCount);
                    n++;
                    continue;
                }
                const PairNode& g = pair[refIndex(r)];
                todo[nt++] = { childOf(g, 1), it.slots - sp };
                todo[nt++] = { childOf(g, 0), sp };
            }
        }
#else
        // the two children of the binary node, then (surface-area greedy) the largest inner child is replaced
        // by its own two children until four are collected: the expensive-to-miss boxes are the ones opened up
        kids[n++] = childOf(pair[i], 0);
        kids[n++] = childOf(pair[i], 1);
        while (n < 4) {
            int best = -1;
            float bestArea = -1.f;
            for (int k = 0; k < n; k++) {
                const uint32_t r = kids[k].ref;
                if (r != kRefNone && refCount(r) == 0u && refIndex(r) < pair.size() && refIndex(r) != i && area(kids[k]) > bestArea) {
                    best = k;
                    bestArea = area(kids[k]);
                }
            }
            if (best < 0)
                break;
            const PairNode& g = pair[refIndex(kids[best].ref)];
            kids[best] = childOf(g, 0);
            kids[n++] = childOf(g, 1);
        }
#endif
        WideKids wk {};
        for (int k = 0; k < 4; k++) {
            wk.empty[k] = k >= n || !(kids[k].lo[0] <= kids[k].hi[0]) || kids[k].ref == kRefNone;
            wk.ref[k] = wk.empty[k] ? kRefNone : kids[k].ref;
            wk.src[k] = k < n ? kids[k].src : 0u;
            for (int a = 0; a < 3; a++) {
                wk.lo[k][a] = wk.empty[k] ? 1.f : kids[k].lo[a];
                wk.hi[k][a] = wk.empty[k] ? -1.f : kids[k].hi[a];
            }
        }
        out[i] = wk;
    }
    });
    return out;
}

// world = inverse(invTransform) by Gauss-Jordan in double; m is column-major (TopBvhNode::invTransform).
// On success w[r][4 + c] holds element (r, c) of the world transform.
bool invertTransform(const float* m, double w[4][8])
{
    for (int r = 0; r < 4; r++)
        for (int col = 0; col < 4; col++) {
            w[r][col] = m[col * 4 + r];
            w[r][col + 4] = (r == col) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(w[r][col]) > std::fabs(w[piv][col]))
                piv = r;
        if (std::fabs(w[piv][col]) < 1e-300)
            return false;
        for (int k = 0; k < 8; k++)
            std::swap(w[piv][k], w[col][k]);
        const double dv = w[col][col];
        for (int k = 0; k < 8; k++)
            w[col][k] /= dv;
        for (int r = 0; r < 4; r++)
            if (r != col) {
                const double f = w[r][col];
                for (int k = 0; k < 8; k++)
                    w[r][k] -= f * w[col][k];
            }
    }
    return true;
}

// ---- the static part of a scene: bottom-level trees, converted once per pt_upload_static / pt_update_geometry -----------------------
// Collapse every mesh tree to 4-wide nodes and pack them breadth-first root by root: the four children of a node get neighbouring
// slots (half the footprint in the 4 MB-per-XCD L2, siblings share 128-byte lines) and a mesh's nodes are ONE contiguous run, which is
// what a world-space copy of an instance (pt_bake.h) is made from.  Roots are the caller's nodes no other node refers to, plus any
// node a top-level leaf has ever named (`extraRoots`).
// pair-node boxes of a refit: the caller's refitted boxes for the pairs that mirror its inner nodes (`onlyExtra`: skipped) and, for the pairs that
// split a leaf of more than kMaxLeafTris triangles (appended children first), the bounds of their triangles
void refitPairBoxes(pt_ctx* c, const pt_vertex* verts, const pt_sub_bvh_node* nodes, bool onlyExtra)
{
    std::vector<PairNode>& pair = c->st->hostBottomNodes;
    if (!onlyExtra)
        for (uint32_t i = 0; i < c->st->numRefNodes; i++) {
            const uint32_t d = c->st->denseOfNode[i];
            if (d == 0xFFFFFFFFu)
                continue;
            const uint32_t l = nodes[i].leftChildOrFirstTriangle;
            const pt_sub_bvh_node &L = nodes[l], &R = nodes[l + 1];
            pair[d].bx = make_float4(L.min[0], L.max[0], R.min[0], R.max[0]);
            pair[d].by = make_float4(L.min[1], L.max[1], R.min[1], R.max[1]);
            pair[d].bz = make_float4(L.min[2], L.max[2], R.min[2], R.max[2]);
        }
    if (pair.size() <= c->st->numDensePairs)
        return;
    auto boxOf = [&](uint32_t ref, V3& lo, V3& hi) {
        lo = mk(FLT_MAX), hi = mk(-FLT_MAX);
        if (refCount(ref) != 0u) {
            for (uint32_t t = refIndex(ref); t < refIndex(ref) + refCount(ref); t++) {
                const TriShade& ts = c->st->hostTriShade[t];
                for (uint32_t vi : { ts.i0, ts.i1, ts.i2 }) {
                    const V3 p = mk(verts[vi].vertex[0], verts[vi].vertex[1], verts[vi].vertex[2]);
                    lo = mk(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
                    hi = mk(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
                }
            }
        } else {
            const PairNode& n = pair[refIndex(ref)];
            lo = mk(fminf(n.bx.x, n.bx.z), fminf(n.by.x, n.by.z), fminf(n.bz.x, n.bz.z));
            hi = mk(fmaxf(n.bx.y, n.bx.w), fmaxf(n.by.y, n.by.w), fmaxf(n.bz.y, n.bz.w));
        }
    };
    for (size_t j = c->st->numDensePairs; j < pair.size(); j++) {
        V3 llo, lhi, rlo, rhi;
        boxOf(pair[j].left, llo, lhi);
        boxOf(pair[j].right, rlo, rhi);
        pair[j].bx = make_float4(llo.x, lhi.x, rlo.x, rhi.x);
        pair[j].by = make_float4(llo.y, lhi.y, rlo.y, rhi.y);
        pair[j].bz = make_float4(llo.z, lhi.z, rlo.z, rhi.z);
    }
}

// the packed 4-wide nodes of a refit on the host: same children in the same slots, new boxes (what k_refit_nodes does on the device)
void refitWideOnHost(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    const std::vector<PairNode>& pair = c->st->hostBottomNodes;
    for (size_t q = 0; q < g.wide.size(); q++) {
        float lo[4][3], hi[4][3];
        uint32_t refs[4];
        bool empty[4];
        for (int k = 0; k < 4; k++) {
            empty[k] = g.kidEmpty[q * 4 + k] != 0u;
            refs[k] = g.wide[q].child[k];
            if (empty[k]) {
                for (int a = 0; a < 3; a++)
                    lo[k][a] = 1.f, hi[k][a] = -1.f;
                continue;
            }
            const uint32_t src = g.kidSrc[q * 4 + k];
            const PairNode& n = pair[src >> 1];
            const int side = (int)(src & 1u);
            const float *bx = &n.bx.x, *by = &n.by.x, *bz = &n.bz.x;
            lo[k][0] = bx[side * 2], hi[k][0] = bx[side * 2 + 1];
            lo[k][1] = by[side * 2], hi[k][1] = by[side * 2 + 1];
            lo[k][2] = bz[side * 2], hi[k][2] = bz[side * 2 + 1];
        }
        for (int k = 0; k < 4; k++)
            for (int a = 0; a < 3; a++)
                g.boxes[q].lo[k][a] = lo[k][a], g.boxes[q].hi[k][a] = hi[k][a];
        quantiseWideNode(lo, hi, refs, empty, g.emptyRef, &g.wide[q]);
    }
}

// the caller's latest vertices / nodes: in the pinned staging memory after a device-side refit, in the host vectors otherwise
inline const pt_vertex* latestVerts(const pt_ctx* c) { return c->st->sg.latestInStage ? (const pt_vertex*)c->st->sg.stage : c->st->rawVerts.data(); }
inline const pt_sub_bvh_node* latestNodes(const pt_ctx* c)
{
    return c->st->sg.latestInStage ? (const pt_sub_bvh_node*)((const unsigned char*)c->st->sg.stage + (size_t)c->st->numVerts * sizeof(pt_vertex)) : c->st->hostSubNodes.data();
}

// The host's mirrors from the caller's arrays as last handed in (a refit re-makes the device's records on the device and leaves these behind): pair-node
// boxes, the packed nodes, hostTris / hostVerts.
// the caller's node boxes recomputed from the latest vertices (what refitBVH leaves, reference src/bvh/refit_bvh.cpp:6-34): after a refit on
// the device alone nobody handed refitted nodes in.  Children lie after their parent (validated at upload): one reverse sweep.
void refitHostNodeBoxes(pt_ctx* c)
{
    const pt_vertex* verts = c->st->rawVerts.data();
    std::vector<pt_sub_bvh_node>& nodes = c->st->hostSubNodes;
    for (size_t i = nodes.size(); i-- > 0;) {
        pt_sub_bvh_node& n = nodes[i];
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        if (n.triangleCount != 0) {
            for (uint32_t t = n.leftChildOrFirstTriangle; t < n.leftChildOrFirstTriangle + n.triangleCount; t++) {
                const TriShade& ts = c->st->hostTriShade[t];
                for (uint32_t vi : { ts.i0, ts.i1, ts.i2 })
                    for (int a = 0; a < 3; a++)
                        lo[a] = fminf(lo[a], verts[vi].vertex[a]), hi[a] = fmaxf(hi[a], verts[vi].vertex[a]);
            }
        } else {
            if (c->st->denseOfNode[i] == 0xFFFFFFFFu)
                continue; // an unused pad
            const pt_sub_bvh_node &L = nodes[n.leftChildOrFirstTriangle], &R = nodes[n.leftChildOrFirstTriangle + 1];
            for (int a = 0; a < 3; a++)
                lo[a] = fminf(L.min[a], R.min[a]), hi[a] = fmaxf(L.max[a], R.max[a]);
        }
        for (int a = 0; a < 3; a++)
            n.min[a] = lo[a], n.max[a] = hi[a];
    }
    c->st->hostNodeBoxesStale = false;
}

void refreshHostGeometry(pt_ctx* c)
{
    if (!c->st->hostGeomStale)
        return;
    if (c->st->hostNodeBoxesStale) { // (rawVerts is current then: pt_refit_vertices keeps it so)
        refitHostNodeBoxes(c);
        refitPairBoxes(c, c->st->rawVerts.data(), c->st->hostSubNodes.data(), false);
        if (c->st->sg.wide.size() == c->st->sg.kidEmpty.size() / 4)
            refitWideOnHost(c);
    }
    const pt_vertex* verts = latestVerts(c);
    if (c->st->sg.latestInStage) {
        refitPairBoxes(c, verts, latestNodes(c), false);
        if (c->st->sg.wide.size() == c->st->sg.kidEmpty.size() / 4)
            refitWideOnHost(c);
    }
    auto P = [&](uint32_t vi) { return mk(verts[vi].vertex[0], verts[vi].vertex[1], verts[vi].vertex[2]); };
    for (size_t t = 0; t < c->st->hostTriShade.size(); t++) {
        const TriShade& ts = c->st->hostTriShade[t];
        const V3 v0 = P(ts.i0);
        const V3 e1 = P(ts.i1) - v0, e2 = P(ts.i2) - v0; // shapes.cl:37-38
        c->st->hostTris[t].a = make_float4(v0.x, v0.y, v0.z, e1.x);
        c->st->hostTris[t].b = make_float4(e1.y, e1.z, e2.x, e2.y);
        c->st->hostTris[t].c = make_float4(e2.z, 0.f, 0.f, 0.f);
    }
    for (size_t v = 0; v < c->st->numVerts; v++) {
        c->st->hostVerts[v].n_u = make_float4(verts[v].normal[0], verts[v].normal[1], verts[v].normal[2], verts[v].texCoord[0]);
        c->st->hostVerts[v].v_pad = make_float4(verts[v].texCoord[1], 0.f, 0.f, 0.f);
    }
    if (c->st->sg.latestInStage) { // the host vectors take the latest arrays over (the staging memory is rewritten by the next refit)
        c->st->rawVerts.assign(verts, verts + c->st->numVerts);
        const pt_sub_bvh_node* nodes = latestNodes(c);
        c->st->hostSubNodes.assign(nodes, nodes + c->st->numRefNodes);
        c->st->sg.latestInStage = false;
    }
    c->st->hostGeomStale = false;
}

// shading records of the caller's triangles: one 128-byte line per triangle (TriFat, pt_device.h)
void buildFat(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    g.fat.resize(c->st->hostTriShade.size());
    for (size_t t = 0; t < c->st->hostTriShade.size(); t++) {
        const TriShade& ts = c->st->hostTriShade[t];
        const VertexShade &a0 = c->st->hostVerts[ts.i0], &a1 = c->st->hostVerts[ts.i1], &a2 = c->st->hostVerts[ts.i2];
        const TriIsect& ti = c->st->hostTris[t];
        TriFat f {};
        f.n0u = a0.n_u, f.n1u = a1.n_u, f.n2u = a2.n_u;
        float mbits;
        std::memcpy(&mbits, &ts.material, 4);
        f.vvvm = make_float4(a0.v_pad.x, a1.v_pad.x, a2.v_pad.x, mbits);
        f.e1e = make_float4(ti.a.w, ti.b.x, ti.b.y, ti.b.z); // edge1.xyz, edge2.x
        f.e2v = make_float4(ti.b.w, ti.c.x, ti.a.x, ti.a.y); // edge2.yz, v0.xy
        float m[12]; // the caller's 48-byte material record: colour (16 B), parameters (16 B), type (+ padding)
        std::memcpy(m, &c->st->hostMaterials[ts.material], sizeof m);
        f.v0c = make_float4(ti.a.z, m[0], m[1], m[2]);
        f.mat = make_float4(m[4], m[5], m[6], m[8]);
        g.fat[t] = f;
    }
}

// PTAMD_UPLOAD_TIMING=1: host time of the stages of an upload, one line per call on stderr (diagnostics; tools/r5_upload_timing.sh)
struct StageTimer {
    const char* what;
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    std::string line;
    explicit StageTimer(const char* w)
        : what(w)
        , on(getenv("PTAMD_UPLOAD_TIMING") != nullptr)
    {
        if (on)
            t0 = last = std::chrono::steady_clock::now();
    }
    void lap(const char* name)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof buf, " %s %.3f", name, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    ~StageTimer()
    {
        if (on)
            fprintf(stderr, "[ptamd] %s: total %.3f ms;%s\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), line.c_str());
    }
};

int buildStaticGeom(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    StageTimer tm("buildStaticGeom");
    refreshHostGeometry(c);
    tm.lap("refreshHostGeometry");
    const uint32_t nN = c->st->numRefNodes, nT = c->st->numTris;
    const std::vector<WideKids> kids = collapseKids(c->st->hostBottomNodes, collapseCostsFromEnv());
    tm.lap("collapseKids");
    const uint32_t emptyRef = makeRef(nT, 1u); // the all-zero triangle stored right after the caller's triangles (det == 0: never hit)
    std::vector<uint8_t> isChild(nN, 0);
    for (uint32_t i = 0; i < nN; i++) {
        const pt_sub_bvh_node& n = c->st->hostSubNodes[i];
        const uint32_t l = n.leftChildOrFirstTriangle;
        if (n.triangleCount == 0 && c->st->nodeRef[i] != kRefNone && (uint64_t)l + 1 < nN)
            isChild[l] = isChild[l + 1] = 1;
    }
    std::vector<uint32_t> rootNodes;
    for (uint32_t i = 0; i < nN; i++)
        if (c->st->nodeRef[i] != kRefNone && (!isChild[i] || std::find(g.extraRoots.begin(), g.extraRoots.end(), i) != g.extraRoots.end()))
            rootNodes.push_back(i);
    g.wide.clear(), g.boxes.clear(), g.leafOfs.clear(), g.refTri.clear(), g.roots.clear(), g.kidSrc.clear(), g.kidEmpty.clear(), g.kidBoxNode.clear();
    std::vector<uint32_t> pairLeft(c->st->numDensePairs, 0u); // pair node -> the caller's node that is its left child
    for (uint32_t i = 0; i < nN; i++)
        if (c->st->denseOfNode[i] != 0xFFFFFFFFu)
            pairLeft[c->st->denseOfNode[i]] = c->st->hostSubNodes[i].leftChildOrFirstTriangle;
    g.rootOfNode.assign(nN, -1);
    tm.lap("roots");
    constexpr uint32_t kUnset = 0xFFFFFFFFu;
    std::vector<uint32_t> newIndex(kids.size(), kUnset), order;
    auto isInner = [&](uint32_t r) { return r != kRefNone && refCount(r) == 0u && refIndex(r) < kids.size(); };
    for (uint32_t rn : rootNodes) {
        StaticScene::StaticGeom::Root root {};
        const uint32_t rref = c->st->nodeRef[rn];
        root.nodeBase = (uint32_t)order.size();
        root.refBase = (uint32_t)g.refTri.size();
        root.bakeable = true;
        if (isInner(rref)) {
            if (newIndex[refIndex(rref)] != kUnset) { // reachable from an earlier root too (a top-level leaf names an interior node): shares its run
                root.ref = makeRef(newIndex[refIndex(rref)], 0u);
                root.bakeable = false;
            } else {
                const size_t first = order.size();
                newIndex[refIndex(rref)] = (uint32_t)order.size();
                order.push_back(refIndex(rref));
                for (size_t q = first; q < order.size(); q++) // breadth first
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r = kids[order[q]].ref[k];
                        if (kids[order[q]].empty[k] || !isInner(r))
                            continue;
                        if (newIndex[refIndex(r)] != kUnset) {
                            root.bakeable = false; // shares nodes with another tree: not one run
                            continue;
                        }
                        newIndex[refIndex(r)] = (uint32_t)order.size();
                        order.push_back(refIndex(r));
                    }
                root.ref = makeRef(root.nodeBase, 0u);
            }
            root.numNodes = (uint32_t)order.size() - root.nodeBase;
        } else {
            root.ref = rref; // the mesh is a single leaf
            for (uint32_t k = 0; k < refCount(rref); k++)
                g.refTri.push_back(refIndex(rref) + k);
        }
        // nodes of this run: remapped references, exact boxes, triangle-reference offsets of the leaves
        g.wide.resize(order.size());
        g.boxes.resize(order.size());
        g.leafOfs.resize(order.size() * 4, 0u);
        g.kidSrc.resize(order.size() * 4, 0u);
        g.kidEmpty.resize(order.size() * 4, 1u);
        g.kidBoxNode.resize(order.size() * 4, 0xFFFFFFFFu);
        for (size_t q = root.nodeBase; q < order.size(); q++) { // the leaves' runs in the table of triangle references: in node order, one after the other
            const WideKids& wk = kids[order[q]];
            for (int k = 0; k < 4; k++)
                if (!wk.empty[k] && !isInner(wk.ref[k])) {
                    g.leafOfs[q * 4 + k] = (uint32_t)g.refTri.size() - root.refBase;
                    for (uint32_t t = 0; t < refCount(wk.ref[k]); t++)
                        g.refTri.push_back(refIndex(wk.ref[k]) + t);
                }
        }
        // ... everything else per node on its own (the quantiser is most of a conversion's time): the host library's worker threads take ranges of them
        raytracer::WorkerPool::get().parallelFor(order.size() - root.nodeBase, 512, [&](size_t q0, size_t q1) {
            for (size_t q = root.nodeBase + q0; q < root.nodeBase + q1; q++) {
                const WideKids& wk = kids[order[q]];
                uint32_t refs[4];
                for (int k = 0; k < 4; k++) {
                    g.kidSrc[q * 4 + k] = wk.src[k], g.kidEmpty[q * 4 + k] = wk.empty[k] ? 1u : 0u;
                    if (!wk.empty[k]) { // the caller's node whose box this slot takes: the left / right child of the node its pair mirrors
                        const uint32_t pr = wk.src[k] >> 1, side = wk.src[k] & 1u;
                        g.kidBoxNode[q * 4 + k] = pr < c->st->numDensePairs ? pairLeft[pr] + side : (0x80000000u | ((pr - c->st->numDensePairs) * 2u + side));
                    }
                    refs[k] = wk.empty[k] ? emptyRef : (isInner(wk.ref[k]) ? makeRef(newIndex[refIndex(wk.ref[k])], 0u) : wk.ref[k]);
                    for (int a = 0; a < 3; a++)
                        g.boxes[q].lo[k][a] = wk.lo[k][a], g.boxes[q].hi[k][a] = wk.hi[k][a];
                }
                quantiseWideNode(wk.lo, wk.hi, refs, wk.empty, emptyRef, &g.wide[q]);
            }
        });
        root.numRefs = (uint32_t)g.refTri.size() - root.refBase;
        g.rootOfNode[rn] = (int32_t)g.roots.size();
        g.roots.push_back(root);
    }
    // worst-case number of pending stack entries below every packed node: visiting a node can leave all its other children on the
    // stack (children are visited nearest first, so any order can occur: the bound takes the deepest child first).  Inside a run the
    // children sit after their parent; a child that lies in ANOTHER run (a top-level leaf named an interior node, whose subtree an
    // earlier root had packed already) lies in an earlier one.  So: run by run in ascending order, each run in reverse -- every child
    // is final when its parent is reached.  (One reverse sweep over everything took 0 for the shared children: too small a bound.)
    tm.lap("pack");
    g.stackNeed.assign(g.wide.size(), 0u);
    for (const StaticScene::StaticGeom::Root& root : g.roots)
        for (size_t q = (size_t)root.nodeBase + root.numNodes; q-- > root.nodeBase;) {
            if (refCount(root.ref) != 0u || refIndex(root.ref) != root.nodeBase)
                break; // no run of its own (a single leaf, or the root sits inside an earlier run)
            uint32_t n = 0, deepest = 0;
            for (uint32_t r : g.wide[q].child) {
                if (r == emptyRef)
                    continue;
                n++;
                if (refCount(r) == 0u && refIndex(r) < g.wide.size())
                    deepest = std::max(deepest, g.stackNeed[refIndex(r)]);
            }
            g.stackNeed[q] = (n > 0 ? n - 1 : 0u) + deepest;
        }
    if (getenv("PTAMD_COLLAPSE_REPORT")) { // what the collapse made: packed nodes, leaves by size (tools / sweeps of PTAMD_LEAF_FORMATION)
        uint64_t hist[kMaxLeafTris + 1] = {}, leaves = 0, refs = 0, used = 0;
        for (const WideNode& w : g.wide)
            for (uint32_t r : w.child)
                if (r != emptyRef) {
                    used++;
                    if (refCount(r) >= 1u && refCount(r) <= kMaxLeafTris)
                        hist[refCount(r)]++, leaves++, refs += refCount(r);
                }
        const CollapseCosts k = collapseCostsFromEnv();
        fprintf(stderr, "[ptamd] collapse: cap %u costs %.0f/%.0f/%.0f alpha %.2f -> %zu wide nodes, %.2f used slots per node, %llu leaves, %.2f triangles per leaf; by size:", k.cap, k.inner,
            k.leaf0, k.tri, k.alpha, g.wide.size(), g.wide.empty() ? 0.0 : (double)used / (double)g.wide.size(), (unsigned long long)leaves, leaves ? (double)refs / (double)leaves : 0.0);
        for (uint32_t n = 1; n <= kMaxLeafTris; n++)
            if (hist[n])
                fprintf(stderr, " %u:%llu", n, (unsigned long long)hist[n]);
        fprintf(stderr, "\n");
    }
    tm.lap("stackNeed");
    buildFat(c);
    tm.lap("buildFat");
    g.emptyRef = emptyRef;
    g.version = ++c->staticVersions;
    g.topology = g.version;
    g.onDevice = false;
    return PT_OK;
}

// the static arrays' master copy in device memory (the two dynamic sets take theirs from it, device to device)
int uploadStaticGeom(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    if (g.onDevice)
        return PT_OK;
    StageTimer tm("uploadStaticGeom");
    // the copy stream may still be reading the old master (a set being refreshed from it)
    HIPCHK(c, hipStreamSynchronize(c->copyStream));
    tm.lap("syncCopyStream");
    if (c->st->hostGeomStale) { // refitted before the master copy ever reached the device
        refreshHostGeometry(c);
        buildFat(c);
    }
    std::vector<TriIsect> tris = c->st->hostTris;
    tris.push_back(TriIsect { make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0) }); // what an unused child slot refers to
    int rc;
    if ((rc = uploadVec(c, g.dWide, g.wide)) || (rc = uploadVec(c, g.dBoxes, g.boxes)) || (rc = uploadVec(c, g.dLeafOfs, g.leafOfs))
        || (rc = uploadVec(c, g.dRefTri, g.refTri)) || (rc = uploadVec(c, g.dTris, tris)) || (rc = uploadVec(c, g.dFat, g.fat))
        || (rc = uploadVec(c, g.dVerts, c->st->rawVerts)) || (rc = uploadVec(c, c->st->triShade, c->st->hostTriShade))
        || (rc = uploadVec(c, g.dNodes, c->st->hostSubNodes)) || (rc = uploadVec(c, g.dKidBoxNode, g.kidBoxNode)))
        return rc;
    tm.lap("uploads");
    g.onDevice = true;
    return PT_OK;
}

int resetStreams(pt_ctx* c)
{
    // one stream per pixel of the full image, created in order (raytracer.cpp:739-751); only the
    // `current` state (16 of clRNG's 48 bytes) is ever read by the kernels
    const size_t n = (size_t)c->cfg.width * c->cfg.height;
    std::vector<uint4> host(n);
    uint32_t g[4] = { 987654321u, 987654321u, 987654321u, 987654321u };
    for (size_t i = 0; i < n; i++) {
        host[i] = make_uint4(g[0], g[1], g[2], g[3]);
        lfsrJump(g);
    }
    return uploadVec(c, c->streams, host);
}

#ifndef PT_SPLIT_SHADOW_ACCUM
#define PT_SPLIT_SHADOW_ACCUM 1 // one sample in flight: shadow rays deposit into an accumulator of their own, a shadow queue per bounce (renderSampleFixed)
#endif
// does a context with queues of `cap` entries render with the shadow rays' own accumulator and a shadow queue per bounce (renderSampleFixed)?
inline bool splitShadowAccum(const pt_ctx* c, uint64_t cap)
{
    return PT_SPLIT_SHADOW_ACCUM && c->planes == 1u && !parityMode(c) && c->cfg.max_active_rays == 0 && cap <= (4u << 20) && !(c->packetUse & 2u) && maxBounces(c) <= (uint32_t)kMaxPasses;
}

bool derivedPrimariesCapable(const pt_ctx* c); // (defined with renderSampleFixed's choice of kernels, below)
// a new camera, scene state or tiling: what a batch's first pass emits is no longer known
inline void newEpoch(pt_ctx* c)
{
    c->epoch++;
    c->ratiosKnown = false;
    c->ratioExt = c->ratioShadow = 0;
}
inline bool smallQueues(const pt_ctx* c) { return c->capExt < c->capacity || c->capShadow < c->capacity; }
inline int checkOverflow(pt_ctx* c)
{
    if (c->overflowPinned && *c->overflowPinned)
        return fail(c, PT_ERR_STATE, "a batch emitted more rays than its queues hold (pt_config.ext_queue_fraction / shadow_queue_fraction; the scene changed under a "
                                    "running batch?): the rays beyond were dropped, the image since the last pt_clear is incomplete -- pt_clear and render again");
    return PT_OK;
}
// adopt a pass-counter report that has landed (renderSampleFixed); the ratios of this epoch only ever grow
inline void adoptPassCounts(pt_ctx* c)
{
    std::memcpy(c->passCountsHint, c->passCountsPinned, sizeof(c->passCountsHint));
    c->passCountsEntries = c->passCountsPending;
    c->passCountsPending = 0;
    if (c->passCountsEpoch == c->epoch && c->passCountsEntries) {
        c->ratioExt = std::max(c->ratioExt, (double)c->passCountsHint[1] / (double)c->passCountsEntries);
        c->ratioShadow = std::max(c->ratioShadow, (double)c->passCountsHint[kMaxPasses + 1] / (double)c->passCountsEntries);
        c->ratiosKnown = true;
    }
}
// the largest batch (samples per pixel) whose first pass fits the queues, by the ratios seen so far + 3 % + 64 K entries (a 64th of a small queue)
inline uint32_t safeBatch(const pt_ctx* c)
{
    auto limit = [&](uint32_t cap, double ratio) -> double {
        if (!(ratio > 0.0))
            return (double)c->planes;
        const double room = (double)cap - std::min(65536.0, (double)cap / 64.0);
        return room / (ratio * 1.03 * (double)c->numOwned);
    };
    double b = std::min(limit(c->capExt, c->ratioExt), limit(c->capShadow, c->ratioShadow));
    if (const char* e = getenv("PTAMD_DEBUG_BATCH_SCALE")) // tests: a batch larger than what fits, so that the overflow guard has something to catch
        b *= atof(e);
    return (uint32_t)std::max(1.0, std::min((double)c->planes, b));
}

int ensureQueues(pt_ctx* c)
{
    if (c->queuesReady)
        return PT_OK;
    if (c->numOwned == 0)
        return fail(c, PT_ERR_STATE, "no pixels owned by this context");
    // samples in flight: only when every (pixel, sample) pair gets its own slot (fixed schedule)
    c->planes = 1;
    if (!parityMode(c) && c->cfg.max_active_rays == 0) {
        uint32_t want = c->cfg.samples_in_flight;
        if (want == 0) // auto: keep ~32M path segments per launch (the latency-bound tail of every launch is then a few % of it)
            want = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(1, (32u << 20) / std::max(c->numOwned, 1u)));
        c->planes = std::min(want, 4096u);
        // a multiple of kGenInterleave, or the power of two below: only such batches keep the samples of a pixel together in the queue, and pt_render
        // cuts every batch that way -- planes (and queue entries) beyond it would be budgeted, allocated and never used (auto at 1280 x 720 gave 36:
        // pt_render(72) ran as 32 + 32 + 8)
        if (c->planes >= kGenInterleave) {
            c->planes -= c->planes % kGenInterleave;
        } else {
            uint32_t p2 = 1;
            while (p2 * 2u <= c->planes)
                p2 *= 2u;
            c->planes = p2;
        }
    }
    uint64_t cap64 = c->cfg.max_active_rays ? c->cfg.max_active_rays : (uint64_t)c->numOwned * c->planes;
    if (cap64 > 0x7FFFFFC0ull)
        return fail(c, PT_ERR_UNSUPPORTED, "%llu queue entries (%u owned pixels x %u samples in flight) exceed the 2^31 entries a queue can index: lower samples_in_flight",
            (unsigned long long)cap64, c->numOwned, c->planes);
    uint32_t cap = ((uint32_t)cap64 + 63u) & ~63u;
    {
        // the buffers of the previous tiling are re-made below anyway: give their memory back first, so that the budget check sees it
        // (a context whose queues use more than half of HBM -- 512 samples in flight at 1080p, two ranks sharing a GPU -- could not be re-tiled)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < 2; k++)
            c->rays[k].o.release(), c->rays[k].d.release(), c->rays[k].thr.release();
        c->shadow.o.release(), c->shadow.d.release(), c->shadow.c.release(), c->hitH.release(), c->hitInst.release(), c->accumPlanes.release();
        for (ShadowQueueBuf& q : c->shadowQ)
            q.o.release(), q.d.release(), q.c.release();
        c->stagedRays.o.release(), c->stagedRays.d.release(), c->stagedRays.thr.release();
        c->stagedShadow.o.release(), c->stagedShadow.d.release(), c->stagedShadow.c.release(), c->activeFlag.release();
        c->foldPlanes = 0;
        // Memory budget, checked before anything is allocated so that an oversized configuration fails HERE with a
        // message instead of somewhere in a later hipMalloc: per queue entry two extension queues (3 x 16 B each), the
        // shadow queue (3 x 16 B) and the hit records (20 B); per owned pixel one 16-byte accumulator plane for every
        // extra sample in flight.  (BASELINE config 5 -- 4K, 8 ranks -- at 2 048 samples in flight would be 2.1 G entries.)
        // one sample in flight and a small queue (the 1-spp frames of RayTracer::rayTrace): the shadow rays' own accumulator (16 B per pixel of the
        // image) and a shadow queue per bounce (48 B per entry and bounce), renderSampleFixed -- set aside HERE, not in the first frame
        const bool split = splitShadowAccum(c, cap);
        // queues smaller than the batch (pt_config, round 6): fixed schedule, batches of >= 16 samples (the probe batch must mean something), no per-bounce queues
        c->capExt = c->capShadow = cap;
        c->q0Small = false;
        const float fe = c->cfg.ext_queue_fraction, fs = c->cfg.shadow_queue_fraction;
        if (!parityMode(c) && c->cfg.max_active_rays == 0 && c->planes >= 16u && !split) {
            if (fe > 0.f && fe < 1.f)
                c->capExt = std::min<uint64_t>(cap, (((uint64_t)((double)cap * fe) + 63u) & ~63ull) + 64u);
            if (fs > 0.f && fs < 1.f)
                c->capShadow = std::min<uint64_t>(cap, (((uint64_t)((double)cap * fs) + 63u) & ~63ull) + 64u);
            // camera rays queued as (direction, pixel) only -- a pinhole's bundles, renderSampleFixed `derived` -- leave the first queue's other planes to the later passes
            c->q0Small = c->capExt < cap && derivedPrimariesCapable(c);
        }
        newEpoch(c);
        const uint64_t q0 = 16ull * cap + 32ull * (c->q0Small ? c->capExt : cap);
        const uint64_t need = q0 + 48ull * c->capExt + 48ull * c->capShadow + 20ull * cap + (uint64_t)cap * (parityMode(c) ? 2ull * 48 + 4 : 0)
            + (uint64_t)(c->planes - 1) * c->numOwned * sizeof(float4)
            + (split ? (uint64_t)cap * 48 * maxBounces(c) + (uint64_t)c->cfg.width * c->cfg.height * sizeof(float4) * maxBounces(c) : 0);
        size_t freeB = 0, totalB = 0;
        HIPCHK(c, hipMemGetInfo(&freeB, &totalB));
        if (need > (uint64_t)freeB)
            return fail(c, PT_ERR_UNSUPPORTED, "queues and accumulator planes need %.1f GB (%u owned pixels x %u samples in flight), %.1f GB of device memory are free: lower samples_in_flight or set max_active_rays",
                need / 1e9, c->numOwned, c->planes, freeB / 1e9);
    }
    if (c->planes > 1) {
        const size_t n = (size_t)(c->planes - 1) * c->numOwned; // [owned-pixel ordinal][plane - 1]
        HIPCHK(c, c->accumPlanes.alloc(n));
        // stream-ordered: the context's stream is non-blocking, a null-stream memset could still be running (or not
        // have started) when the first kernels of the render write these buffers
        HIPCHK(c, hipMemsetAsync(c->accumPlanes.p, 0, n * sizeof(float4), c->stream));
    }
    c->capacity = cap;
    HIPCHK(c, c->rays[0].o.alloc(c->q0Small ? c->capExt : cap));
    HIPCHK(c, c->rays[0].d.alloc(cap));
    HIPCHK(c, c->rays[0].thr.alloc(c->q0Small ? c->capExt : cap));
    HIPCHK(c, c->rays[1].o.alloc(c->capExt));
    HIPCHK(c, c->rays[1].d.alloc(c->capExt));
    HIPCHK(c, c->rays[1].thr.alloc(c->capExt));
    HIPCHK(c, c->shadow.o.alloc(c->capShadow));
    HIPCHK(c, c->shadow.d.alloc(c->capShadow));
    HIPCHK(c, c->shadow.c.alloc(c->capShadow));
    HIPCHK(c, c->hitH.alloc(cap));
    HIPCHK(c, c->hitInst.alloc(cap));
    if (parityMode(c)) {
        HIPCHK(c, c->stagedRays.o.alloc(cap));
        HIPCHK(c, c->stagedRays.d.alloc(cap));
        HIPCHK(c, c->stagedRays.thr.alloc(cap));
        HIPCHK(c, c->stagedShadow.o.alloc(cap));
        HIPCHK(c, c->stagedShadow.d.alloc(cap));
        HIPCHK(c, c->stagedShadow.c.alloc(cap));
        HIPCHK(c, c->activeFlag.alloc(cap));
        int rc = resetStreams(c);
        if (rc)
            return rc;
    }
    if (splitShadowAccum(c, cap)) { // (a first-frame stall otherwise: thirteen hipMallocs inside the first pt_render)
        const size_t npx = (size_t)c->cfg.width * c->cfg.height;
        if (!c->accumShadow.p || c->accumShadow.n < npx * maxBounces(c)) { // one plane per bounce: the shadow passes of a frame deposit side by side
            HIPCHK(c, c->accumShadow.alloc(npx * maxBounces(c)));
            HIPCHK(c, hipMemsetAsync(c->accumShadow.p, 0, npx * maxBounces(c) * sizeof(float4), c->stream));
        }
        for (uint32_t b = 0; b < maxBounces(c); b++) {
            HIPCHK(c, c->shadowQ[b].o.alloc(cap));
            HIPCHK(c, c->shadowQ[b].d.alloc(cap));
            HIPCHK(c, c->shadowQ[b].c.alloc(cap));
        }
    }
    HIPCHK(c, c->control.alloc(1));
    HIPCHK(c, hipMemsetAsync(c->control.p, 0, sizeof(Control), c->stream));
    c->queuesReady = true;
    return PT_OK;
}

int ensureSpill(pt_ctx* c)
{
    if (c->spill.p)
        return PT_OK;
    // persistent grids sized to the machine, per instantiation pair ([0]: scenes that are one world-space tree, [1]: scenes with
    // instance references -- pt_trace.h, TWO_LEVEL)
    const void* variants[3][2] = { { (const void*)k_trace<false, 0>, (const void*)k_trace<true, 0> }, { (const void*)k_trace<false, 1>, (const void*)k_trace<true, 1> },
        { (const void*)k_trace<false, 2>, (const void*)k_trace<true, 2> } };
    const void* packetVariants[2][2] = { { (const void*)k_trace_packet<false, false>, (const void*)k_trace_packet<true, false> },
        { (const void*)k_trace_packet<false, true>, (const void*)k_trace_packet<true, true> } };
    for (int tl = 0; tl < 3; tl++) {
        int blocksPerCU = 8;
        for (const void* fn : variants[tl]) {
            int b = 0;
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, fn, kTraceBlock, 0));
            blocksPerCU = std::min(blocksPerCU, b);
        }
        blocksPerCU = std::max(1, blocksPerCU);
        if (const char* e = getenv("PTAMD_TRACE_BLOCKS_PER_CU")) // diagnostics: a smaller persistent grid leaves wave slots to kernels of other streams / processes
            blocksPerCU = std::max(1, std::min(blocksPerCU, atoi(e)));
        c->traceBlocks[tl] = (uint32_t)(blocksPerCU * c->numCUs);
        if (tl >= 2)
            continue; // (the packet kernels know two kinds of scene)
        int pb = 8;
        for (const void* fn : packetVariants[tl]) {
            int b = 0;
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, fn, kPacketBlock, 0));
            pb = std::min(pb, b);
        }
        pb = std::max(1, pb);
        if (const char* e = getenv("PTAMD_PACKET_BLOCKS_PER_CU"))
            pb = std::max(1, std::min(pb, atoi(e)));
        c->packetBlocks[tl] = (uint32_t)(pb * c->numCUs);
    }
    {
        int b = 0;
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)k_trace_multi<PT_MULTI_RAYS, false>, kPacketBlock, 0));
        c->multiBlocks[0] = (uint32_t)(std::max(1, b) * c->numCUs);
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)k_trace_multi<PT_MULTI_RAYS, true>, kPacketBlock, 0));
        c->multiBlocks[1] = (uint32_t)(std::max(1, b) * c->numCUs);
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)k_trace_multi<PT_MULTI_RAYS, false, true>, kPacketBlock, 0));
        c->multiBlocks[2] = (uint32_t)(std::max(1, b) * c->numCUs);
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)k_trace_multi<PT_MULTI_RAYS, true, true>, kPacketBlock, 0));
        c->multiBlocks[3] = (uint32_t)(std::max(1, b) * c->numCUs);
        if (const char* e = getenv("PTAMD_PACKET_BLOCKS_PER_CU")) // the documented knob reaches the bundle kernel too
            for (uint32_t& mb : c->multiBlocks)
                mb = std::max(1u, std::min(mb, (uint32_t)std::max(1, atoi(e)) * (uint32_t)c->numCUs));
    }
    {
        int t0 = 0, t1 = 0;
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&t0, (const void*)k_trace_team<false>, kTeamBlock, 0));
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&t1, (const void*)k_trace_team<true>, kTeamBlock, 0));
        c->teamBlocks = (uint32_t)(std::max(1, std::min(t0, t1)) * c->numCUs);
        if (const char* e = getenv("PTAMD_TEAM_ROUNDS")) // diagnostics: 0 = never use the team kernel
            c->teamRounds = std::max(0.f, (float)atof(e));
        if (const char* e = getenv("PTAMD_TEAM_USE"))
            c->teamUse = (uint32_t)atoi(e);
    }
    const size_t threads = (size_t)std::max(std::max(c->traceBlocks[0], c->traceBlocks[1]), c->traceBlocks[2]) * kTraceBlock;
    HIPCHK(c, c->spill.alloc(3 * threads * kSpillStack)); // second and third part: the traversal kernels that run beside another one (the two side streams)
    c->spillHalf = threads * kSpillStack;
    return PT_OK;
}

inline int sceneKind(const pt_ctx* c)
{
    static const bool forceTwoLevel = getenv("PTAMD_FORCE_TWO_LEVEL_KERNELS") != nullptr; // diagnostics: what do the instantiations that CAN enter instances cost on a scene without any?
    return (c->dyn[c->active].hasInstances || forceTwoLevel) ? 1 : 0;
}

// which instantiation of the per-ray kernel (pt_trace.h, LEVELS): 2 = instances of ANY transform, entered as leaf-kind steps
inline int traceKind(const pt_ctx* c) { return sceneKind(c) == 0 ? 0 : (c->dyn[c->active].generalRoute ? 2 : 1); }

// Is this launch small enough for four lanes per ray (pt_team.h)?  Known only as a hint -- the live count is a device word --: what the same pass of
// the previous batch of the same size held (its counters come back through pinned memory, renderSampleFixed); shadow rays of pass b are at most the
// extension rays of pass b.  Scenes that are one world-space tree whose depth-first stack need fits the team's stack; never in parity mode.
bool teamLaunch(const pt_ctx* c, uint32_t pass, bool anyHit = false)
{
    if (!(c->teamRounds > 0.f) || !c->teamBlocks || parityMode(c) || c->dyn[c->active].hasInstances || c->dyn[c->active].stackNeed > kTeamStackNeedMax)
        return false;
    if (c->cfg.flags & PT_FLAG_TEAM_INTERSECT)
        return true; // the pt_intersect hook (tests)
    if (!c->batchEntries || pass > (uint32_t)kMaxPasses)
        return false;
    // the camera rays of a 1-spp frame (no bundles there: too few samples of a pixel): coherent rays, few leaves per ray -- four lanes per ray walk them
    // faster than one however many there are (1280 x 720: 190 instead of 263 us, profiles/round5/r5l_frame_trace_team.txt)
    if (pass == 0u && c->planes == 1u && c->batchEntries <= (4u << 20) && (c->teamUse & (anyHit ? 4u : 1u)))
        return true;
    if (c->passCountsEntries != c->batchEntries || !(c->teamUse & 2u))
        return false;
    const uint64_t teams = (uint64_t)c->teamBlocks * (kTeamBlock / 4);
    return (double)c->passCountsHint[pass] <= (double)teams * c->teamRounds;
}

void launchTrace(pt_ctx* c, bool anyHit, const TraceArgs& args, hipStream_t stream = nullptr)
{
    if (teamLaunch(c, args.pass, anyHit)) {
        c->teamLaunches++;
        if (anyHit)
            hipLaunchKernelGGL(k_trace_team<true>, dim3(c->teamBlocks), dim3(kTeamBlock), 0, stream ? stream : c->stream, args);
        else
            hipLaunchKernelGGL(k_trace_team<false>, dim3(c->teamBlocks), dim3(kTeamBlock), 0, stream ? stream : c->stream, args);
        return;
    }
    // the instantiation that can enter instances only where the tree holds instance references (pt_trace.h)
    const bool twoLevel = sceneKind(c) != 0;
    const int kind = traceKind(c);
    TraceArgs a = args;
    if (twoLevel) { // the per-ray kernels walk the top level in which folded instances are plain inner references (the packet kernels: the one with instance references)
        a.sc.rootRef = c->dyn[c->active].rootRefFolded;
        a.instFold = c->dyn[c->active].instFold.p, a.instFoldCount = c->dyn[c->active].instFoldCount;
    }
    const dim3 grid(c->traceBlocks[kind]), block(kTraceBlock);
    if (!stream)
        stream = c->stream;
    if (anyHit) {
        if (kind == 2)
            hipLaunchKernelGGL((k_trace<true, 2>), grid, block, 0, stream, a);
        else if (kind == 1)
            hipLaunchKernelGGL((k_trace<true, 1>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((k_trace<true, 0>), grid, block, 0, stream, a);
    } else {
        if (kind == 2)
            hipLaunchKernelGGL((k_trace<false, 2>), grid, block, 0, stream, a);
        else if (kind == 1)
            hipLaunchKernelGGL((k_trace<false, 1>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((k_trace<false, 0>), grid, block, 0, stream, a);
    }
}

TraceArgs traceArgsBase(pt_ctx* c)
{
    TraceArgs a {};
    a.sc = c->scene;
    a.spill = c->spill.p;
    a.totalThreads = c->traceBlocks[traceKind(c)] * kTraceBlock;
    a.parityShadow = parityMode(c) ? 1u : 0u;
    return a;
}

FrameParams frameParams(const pt_ctx* c, uint32_t sample)
{
    FrameParams fp {};
    fp.cam = c->camera;
    fp.width = c->cfg.width;
    fp.height = c->cfg.height;
    fp.sample = sample;
    fp.seed = c->cfg.seed;
    fp.maxBounces = maxBounces(c);
    fp.parity = parityMode(c) ? 1u : 0u;
    fp.numOwned = c->numOwned;
    fp.planes = 1;
    fp.interleave = 1;
    fp.interleaveShift = 0;
    fp.invWidth = 1.0f / (float)c->cfg.width;
    fp.integrator = (c->cfg.flags & PT_FLAG_COMPARE_SHADING) ? INTEGRATOR_COMPARE : ((c->cfg.flags & PT_FLAG_INTEGRATOR_MIS) ? INTEGRATOR_MIS : INTEGRATOR_IS);
    fp.weightedLights = (c->cfg.flags & PT_FLAG_SOLID_ANGLE_LIGHTS) ? 1u : 0u;
    fp.invSpan = 0.f;
    return fp;
}

struct Prof {
    pt_ctx* c;
    size_t next = 0;
    std::vector<std::pair<int, size_t>> marks; // (family, event index of start); stop = +1
    void begin(int family)
    {
        if (!c->profile)
            return;
        if (c->profEvents.size() < next + 2) {
            size_t old = c->profEvents.size();
            c->profEvents.resize(next + 2);
            for (size_t i = old; i < c->profEvents.size(); i++)
                (void)hipEventCreate(&c->profEvents[i]);
        }
        (void)hipEventRecord(c->profEvents[next], c->stream);
        marks.push_back({ family, next });
    }
    void end()
    {
        if (!c->profile)
            return;
        (void)hipEventRecord(c->profEvents[next + 1], c->stream);
        next += 2;
    }
};

void launchGen(pt_ctx* c, const FrameParams& fp, int q, uint32_t first, uint32_t n, uint32_t slotBase, uint32_t pass)
{
    Control* ctl = c->control.p;
    // several samples in flight: one grid row per group of `interleave` samples (k_gen)
    const uint32_t span = fp.planes > 1u ? fp.numOwned * fp.interleave : std::max(n, 1u);
    const uint32_t blocks = (span + 255u) / 256u, rows = fp.planes > 1u ? fp.planes / fp.interleave : 1u;
    c->genLaunches++;
    hipLaunchKernelGGL(k_gen, dim3(blocks, rows), dim3(256), 0, c->stream, fp, c->rays[q].view(), c->identityPixels ? nullptr : c->pixelList.p,
        first, n, slotBase, c->streams.p, &ctl->extCount[pass], &ctl->generated);
}

#ifndef PT_FRAME_BUNDLES
#define PT_FRAME_BUNDLES 0 // 1: the camera rays of a 1-spp frame (pinhole) as bundles of 256 neighbouring pixels through k_trace_multi.  Measured (1280 x 720,
                           // one bundle per wave): 1.39 instead of 1.02 ms per frame -- 3 600 walks of a 32 x 8-pixel beam, each a chain of > 100 dependent leaf visits
#endif
#ifndef PT_FUSED_PRIMARY
#define PT_FUSED_PRIMARY 1 // primary rays regenerated by the packet kernel and the first k_shade instead of queued by k_gen
#endif
#ifndef PT_SHADE_SPLIT
#define PT_SHADE_SPLIT 1
#endif
#ifndef PT_LAST_SHADOW_ON_MAIN
#define PT_LAST_SHADOW_ON_MAIN 1
#endif
#ifndef PT_DERIVED_PRIMARIES
#define PT_DERIVED_PRIMARIES 1
#endif
#ifndef PT_OVERLAP_SMALL
#define PT_OVERLAP_SMALL 1 // small launches: shadow rays of bounce b beside the extension rays of bounce b + 1 (side stream)
#endif
#ifndef PT_PACKET_USE
#define PT_PACKET_USE 1 // primary rays only: shadow rays towards random light points are not coherent enough (2.6x slower)
#endif
constexpr uint32_t kPacketUseDefault = PT_PACKET_USE;

void launchPacket(pt_ctx* c, bool anyHit, const TraceArgs& a)
{
    const bool twoLevel = sceneKind(c) != 0;
    const dim3 grid(c->packetBlocks[twoLevel ? 1 : 0]), block(kPacketBlock);
    c->packetLaunches++;
    if (anyHit) {
        if (twoLevel)
            hipLaunchKernelGGL((k_trace_packet<true, true>), grid, block, 0, c->stream, a);
        else
            hipLaunchKernelGGL((k_trace_packet<true, false>), grid, block, 0, c->stream, a);
    } else {
        if (twoLevel)
            hipLaunchKernelGGL((k_trace_packet<false, true>), grid, block, 0, c->stream, a);
        else
            hipLaunchKernelGGL((k_trace_packet<false, false>), grid, block, 0, c->stream, a);
    }
}

// first pass of a batch: may the camera rays be generated inside k_trace_multi and walk the tree as bundles (pinhole camera)?
// (thin-lens cameras too since round 6: converging bundles, pt_packet_multi.h LENS; PTAMD_LENS_BUNDLES=0: packets of 64 as in rounds 3-5)
inline bool lensBundles()
{
    static const bool on = !(getenv("PTAMD_LENS_BUNDLES") && atoi(getenv("PTAMD_LENS_BUNDLES")) == 0);
    return on;
}
inline bool primaryBundles(const pt_ctx* c)
{
    return PT_MULTI_RAYS > 1 && (!c->camera.thinLens || lensBundles()) && !(c->packetUse & 8u) && c->dyn[c->active].packetOk && (c->packetUse & 1u);
}
// are consecutive entries of the first queue of a batch rays of one pixel or of neighbouring pixels?  >= 16 samples of a pixel next to each other, or the
// can the camera rays of a large batch be queued as (direction, pixel) only?  (renderSampleFixed's `derived`, asked before the batch exists: ensureQueues)
bool derivedPrimariesCapable(const pt_ctx* c)
{
    return c->haveCamera && c->haveDynamic && !c->camera.thinLens && primaryBundles(c) && PT_FUSED_PRIMARY && PT_DERIVED_PRIMARIES && !(c->cfg.flags & PT_FLAG_QUEUE_PRIMARY_RAYS);
}

// pixels of a 1-spp frame in the order of the pixel list (8 x 8 blocks unless the caller chose otherwise) where bundles of 256 serve them
inline bool firstPassCoherent(const pt_ctx* c, const FrameParams& fp, uint32_t batch) { return fp.interleave >= 16u || (PT_FRAME_BUNDLES && batch == 1u && primaryBundles(c)); }

// `coherent`: consecutive queue entries are samples of one pixel (first pass of the fixed schedule)
void launchIntersect(pt_ctx* c, int q, uint32_t pass, bool coherent = false, const FrameParams* fused = nullptr, bool noOrigins = false)
{
    Control* ctl = c->control.p;
    TraceArgs a = traceArgsBase(c);
    if (fused) {
        a.fused = 1u;
        a.fp = *fused;
        a.pixelList = c->identityPixels ? nullptr : c->pixelList.p;
    }
    a.rayO = c->rays[q].o.p;
    a.rayD = c->rays[q].d.p;
    a.hit = c->hitH.p;
    a.inst = c->hitInst.p;
    a.ctl = ctl;
    a.pass = pass;
    if (coherent && c->dyn[c->active].packetOk && (c->packetUse & 1u)) {
        // camera rays of a pinhole generated in the kernel: PT_MULTI_RAYS x 64 consecutive entries -- the samples of
        // one pixel, or of neighbouring pixels -- are ONE bundle and are walked as one (pt_packet_multi.h)
        if (fused && primaryBundles(c)) {
            a.noOrigins = noOrigins ? 1u : 0u;
            c->packetLaunches++;
            c->bundleLaunches++;
            if (c->camera.thinLens) { // converging bundles: every ray its own origin (which stays in the queue: k_shade cannot derive it)
                a.noOrigins = 0u;
                if (sceneKind(c) != 0)
                    hipLaunchKernelGGL((k_trace_multi<PT_MULTI_RAYS, true, true>), dim3(c->multiBlocks[3]), dim3(kPacketBlock), 0, c->stream, a);
                else
                    hipLaunchKernelGGL((k_trace_multi<PT_MULTI_RAYS, false, true>), dim3(c->multiBlocks[2]), dim3(kPacketBlock), 0, c->stream, a);
            } else if (sceneKind(c) != 0)
                hipLaunchKernelGGL((k_trace_multi<PT_MULTI_RAYS, true>), dim3(c->multiBlocks[1]), dim3(kPacketBlock), 0, c->stream, a);
            else
                hipLaunchKernelGGL((k_trace_multi<PT_MULTI_RAYS, false>), dim3(c->multiBlocks[0]), dim3(kPacketBlock), 0, c->stream, a);
        } else {
            launchPacket(c, false, a);
        }
    } else {
        launchTrace(c, false, a);
    }
}

// `own`: the pass's own shadow queue and the shadow rays' own accumulator (one sample in flight, renderSampleFixed)
void launchShadow(pt_ctx* c, uint32_t pass, bool coherent = false, hipStream_t side = nullptr, const ShadowQueueBuf* own = nullptr)
{
    Control* ctl = c->control.p;
    TraceArgs a = traceArgsBase(c);
    if (side) // runs beside the closest-hit traversal of the next bounce (and, with two side streams, beside another shadow pass): a spill region of its own
        a.spill = c->spill.p + c->spillHalf * (side == c->sideStream2 ? 2u : 1u);
    const ShadowQueueBuf& q = own ? *own : c->shadow;
    a.rayO = q.o.p;
    a.rayD = q.d.p;
    a.rayC = q.c.p;
    a.accum = own ? AccumView { c->accumShadow.p + (size_t)pass * c->cfg.width * c->cfg.height, nullptr, nullptr, 0u } : accumView(c); // (own: this bounce's plane)
    a.ctl = ctl;
    a.pass = pass;
    if (coherent && c->dyn[c->active].packetOk && (c->packetUse & 2u))
        launchPacket(c, true, a);
    else
        launchTrace(c, true, a, side);
}

// shade over `launchEntries` slots (upper bound of the live count) of queue `in` -> queue `out` + shadow queue
void launchShade(pt_ctx* c, const FrameParams& fp, int in, int out, uint32_t pass, uint32_t launchEntries, const ShadowQueueBuf* ownShadow = nullptr, bool derivedPrimaries = false)
{
    Control* ctl = c->control.p;
    ShadeArgs a {};
    a.sc = c->scene;
    a.fp = fp;
    a.in = c->rays[in].view();
    a.hits = { c->hitH.p, c->hitInst.p };
    a.accum = accumView(c);
    a.inCount = &ctl->extCount[pass];
    a.outCount = &ctl->extCount[pass + 1];
    a.shadowCount = &ctl->shadowCount[pass];
    a.shadeHits = &ctl->shadeHits[pass];
    a.deposits = parityMode(c) ? &ctl->depositsShade : &ctl->depositSlots[0][0]; // the production kernel spreads its count over the slots (pt_device.h)
    a.streams = c->streams.p;
    a.derivedPrimaries = derivedPrimaries ? 1u : 0u;
    // what the queues written here hold (pt_shade.h): the second queue capExt; the first queue capacity, or capExt where its origin / throughput planes are small
    a.outCap = out == 1 ? c->capExt : (c->q0Small ? c->capExt : c->capacity);
    a.shadowCap = ownShadow ? c->capacity : c->capShadow;
    const uint32_t blocks = (std::max(launchEntries, 1u) + kShadeBlock - 1u) / kShadeBlock; // those beyond the live count leave at once
    if (parityMode(c)) {
        a.out = c->stagedRays.view();
        a.shadow = c->stagedShadow.view();
        a.activeFlag = c->activeFlag.p;
        if (generalShading(c))
            hipLaunchKernelGGL((k_shade<true, true>), dim3(blocks), dim3(kShadeBlock), 0, c->stream, a);
        else
            hipLaunchKernelGGL((k_shade<true, false>), dim3(blocks), dim3(kShadeBlock), 0, c->stream, a);
        CompactArgs ca {};
        ca.staged = c->stagedRays.view();
        ca.out = c->rays[out].view();
        ca.stagedShadow = c->stagedShadow.view();
        ca.outShadow = c->shadow.view();
        ca.activeFlag = c->activeFlag.p;
        ca.inCount = &ctl->extCount[pass];
        ca.outCount = &ctl->extCount[pass + 1];
        ca.shadowCount = &ctl->shadowCount[pass];
        ca.shadeHits = &ctl->shadeHits[pass];
        hipLaunchKernelGGL(k_compact_stable, dim3(1), dim3(1024), 0, c->stream, ca);
    } else {
        a.out = c->rays[out].view();
        a.shadow = ownShadow ? ownShadow->view() : c->shadow.view();
        // The queue of pass b holds what survived b bounces -- 23 / 6 / 1.3 % of the capacity on the benchmark scene, more than half per
        // bounce behind glass -- but how much is a device word, and a million workgroups that leave at once cost 0.6 ms per launch to
        // dispatch.  So pass b >= 1 launches the one-tile kernel over as many tiles as the same pass of the previous batch filled (its
        // counters come back through pinned memory, unwaited-for; the first batch of a context assumes a half per bounce) and, behind
        // it, a 512-workgroup grid of the tile-walking kernel for whatever lies beyond -- a safety net that normally finds nothing.
        uint32_t head = blocks;
        if (PT_SHADE_SPLIT && pass > 0) {
            if (c->passCountsEntries && !c->shadeHeadShift) {
                // what the same pass of the last finished batch held, scaled to this batch's size, + 3 % + 8 tiles
                const double scale = (double)launchEntries / (double)c->passCountsEntries;
                const double guess = (double)c->passCountsHint[pass] * scale * 1.03;
                head = std::min(blocks, (uint32_t)(guess / kShadeBlock) + 8u);
            } else {
                head = std::max(1u, blocks >> std::min(pass + c->shadeHeadShift, 24u)); // no history yet: half per bounce
            }
        }
        if (generalShading(c))
            hipLaunchKernelGGL((k_shade<false, true>), dim3(head), dim3(kShadeBlock), 0, c->stream, a);
        else if (c->st->materialBins)
            hipLaunchKernelGGL((k_shade<false, false, false, true>), dim3(head), dim3(kShadeBlock), 0, c->stream, a);
        else
            hipLaunchKernelGGL((k_shade<false, false>), dim3(head), dim3(kShadeBlock), 0, c->stream, a);
        if (head < blocks) {
            a.firstTile = head;
            // (a 1-spp frame's passes: 64 workgroups -- launching 512 that find nothing took 4-5 us of a ~700 us frame three times over)
            const uint32_t rest = std::min(blocks - head, launchEntries <= (4u << 20) ? 64u : 512u);
            if (generalShading(c))
                hipLaunchKernelGGL((k_shade<false, true, true>), dim3(rest), dim3(kShadeBlock), 0, c->stream, a);
            else
                hipLaunchKernelGGL((k_shade<false, false, true>), dim3(rest), dim3(kShadeBlock), 0, c->stream, a);
        }
    }
}

// frame parameters of a batch of `batch` samples per owned pixel: how many of a pixel's samples sit next to each other in the queue
FrameParams batchFrameParams(pt_ctx* c, uint32_t sample, uint32_t batch)
{
    FrameParams fp = frameParams(c, sample);
    fp.planes = batch;
    fp.interleave = 1;
    while (fp.interleave < kGenInterleave && batch % (fp.interleave * 2u) == 0u)
        fp.interleave *= 2u, fp.interleaveShift++;
    fp.invSpan = 1.0f / (float)((uint64_t)c->numOwned << fp.interleaveShift);
    return fp;
}

// Fixed launch schedule for one sample when every owned pixel has its own queue slot: gen, then
// maxBounces x (intersect, shade, shadow intersect), then the bookkeeping kernel.  No host
// read-back anywhere (the reference blocks on a 176-byte read every pass, raytracer.cpp:381-389).
int renderSampleFixed(pt_ctx* c, uint32_t sample, uint32_t batch, Prof& prof)
{
    FrameParams fp = batchFrameParams(c, sample, batch);
    if (c->passCountsPending) {
        if (hipEventQuery(c->passCountsCopied) == hipSuccess) { // the latest copy has landed: adopt it
            adoptPassCounts(c);
        } else {
            (void)hipGetLastError(); // "not ready" is an answer, not an error: it must not be what the check at the end of the batch finds
        }
    }
    const uint32_t bounces = maxBounces(c);
    const uint32_t entries = c->numOwned * batch;
    c->batchEntries = entries;
    // Where the packet kernel serves the primary rays it generates them itself, from the entry index, and queues them for
    // k_shade: no k_gen launch (3.3 ms of a 121 ms batch, HBM-write-bound) and no read of 32 B per ray in a kernel that has
    // bandwidth to spare for the two stores instead.  (k_shade regenerating the rays as well, so that they are never stored,
    // was measured too: its 70 extra instructions per entry cost 2.3 ms per batch, more than the reads they replace.)
    // The packet kernel serves the first pass when consecutive queue entries are >= 16 samples of one pixel.  (Packets of 8x8 pixel
    // blocks -- what the default pixel order would give a 1-spp frame -- were measured too: the beam test handles them, but a
    // 1280x720 frame is 14 k packets for 8 k persistent waves claiming 16 at a time: 2.2-2.4 ms per frame instead of 1.4-1.6.)
    // (8x8-pixel packets for a 1-spp frame were measured again in round 3 with one packet per claim: 271 us for the 14 400 packets of
    // a 1280 x 720 frame against 251 us through the per-ray kernel -- 1.8 rounds of latency-bound packet walks on 8 192 waves)
    const bool coherentFirst = firstPassCoherent(c, fp, batch);
    const bool packetsFirst = coherentFirst && c->dyn[c->active].packetOk && (c->packetUse & 1u);
    const bool fused = packetsFirst && PT_FUSED_PRIMARY && !(c->cfg.flags & PT_FLAG_QUEUE_PRIMARY_RAYS);
    // ... and where those are the bundles of a pinhole camera, only (direction, pixel) is queued: k_shade takes the eye as the origin and the sample from the
    // entry index (12 instructions; the full regeneration the paragraph above dismissed is 70) -- 16 B per camera ray less written and 16 B less read
    const bool derived = fused && primaryBundles(c) && !c->camera.thinLens && PT_DERIVED_PRIMARIES; // (a thin lens: every ray has an origin of its own, which stays in the queue)
    if (c->q0Small && !derived)
        return fail(c, PT_ERR_STATE, "the first queue was sized for camera rays queued as directions only, and this batch queues their origins");
    prof.begin(0);
    if (fused)
        hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, c->stream, &c->control.p->extCount[0], &c->control.p->generated, entries);
    else
        launchGen(c, fp, 0, 0, entries, 0, 0);
    prof.end();
    // Small launches are latency-bound (every traversal launch of a 1-spp 1280 x 720 frame takes 0.1-0.25 ms whatever it holds): the
    // shadow rays of bounce b then run on a side stream BESIDE the extension rays of bounce b + 1.  Both need only shade b; shade
    // b + 1 waits for both, so the accumulator sees its deposits in the same order as in the serial schedule (bit-identical images).
    // Large batches fill the machine with one kernel; two traversal kernels side by side only evict each other's nodes (measured: slower).
    const bool overlap = PT_OVERLAP_SMALL && entries <= (4u << 20) && !c->profile && !(c->packetUse & 2u);
    // One sample in flight (RayTracer::rayTrace's frames): every entry deposits into the accumulator proper, so the shade launch of bounce b + 1 had to wait
    // for the shadow rays of bounce b (same words, same order as the serial schedule) -- and the shadow passes, the longer ones, were the frame's critical path.
    // There the shadow rays get an accumulator of their own (added to the other at the end of pt_render, in either schedule: the images stay bit-identical
    // between them) and a queue per bounce: a shadow pass then waits for its own shade launch only.
    const bool split = splitShadowAccum(c, c->capacity);
    if (split) { // (its buffers were set aside with the queues: ensureQueues)
        if (!c->accumShadow.p || !c->shadowQ[bounces - 1].o.p)
            return fail(c, PT_ERR_STATE, "the buffers of the one-sample-in-flight schedule are missing");
        c->mergePending = true;
    }
    int in = 0, out = 1;
    for (uint32_t b = 0; b < bounces; b++) {
        prof.begin(1);
        const bool coherent = b == 0 && coherentFirst;
        if (c->profile && coherent && c->dyn[c->active].packetOk && (c->packetUse & 1u))
            prof.marks.back().first = 4; // timed apart from the per-ray kernel (ms_packet)
        launchIntersect(c, in, b, coherent, fused && b == 0 ? &fp : nullptr, derived && b == 0);
        prof.end();
        if (overlap && !split && b > 0)
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->evShadowed[b - 1], 0)); // the deposits of bounce b - 1's shadow rays come first
        prof.begin(2);
        launchShade(c, fp, in, out, b, entries, split ? &c->shadowQ[b] : nullptr, derived && b == 0);
        if (b == 0 && (c->capExt < c->capacity || c->capShadow < c->capacity)) // (later passes emit at most what they were handed: only the first can outgrow a queue)
            hipLaunchKernelGGL(k_clamp_counts, dim3(1), dim3(64), 0, c->stream, c->control.p, 0u, c->capExt, c->capShadow, c->overflowPinned);
        prof.end();
        prof.begin(3);
        if (overlap && b + 1u == bounces && PT_LAST_SHADOW_ON_MAIN) {
            // the last bounce's shadow rays have no extension pass to run beside: on the render stream itself, behind their shade launch -- the wait for a
            // side stream's event that has only just fired was a 17 us hole in front of k_end_sample in every frame's trace
            launchShadow(c, b, coherent, nullptr, split ? &c->shadowQ[b] : nullptr);
        } else if (overlap) {
            // (with a plane and a queue per bounce the shadow passes depend on nothing but their own shade launch: two side streams take them in turn, so
            // that the pass of bounce b does not queue behind the longer one of bounce b - 1 -- the side stream had become a frame's critical path)
            hipStream_t side = split && (b & 1u) ? c->sideStream2 : c->sideStream;
            HIPCHK(c, hipEventRecord(c->evShaded[b], c->stream));
            HIPCHK(c, hipStreamWaitEvent(side, c->evShaded[b], 0));
            launchShadow(c, b, coherent, side, split ? &c->shadowQ[b] : nullptr);
            HIPCHK(c, hipEventRecord(c->evShadowed[b], side));
        } else {
            launchShadow(c, b, coherent, nullptr, split ? &c->shadowQ[b] : nullptr);
        }
        prof.end();
        std::swap(in, out);
    }
    if (overlap && PT_LAST_SHADOW_ON_MAIN) { // the side streams' last passes (long done, as a rule)
        if (split && bounces > 1)
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->evShadowed[bounces - 2], 0));
        if (split && bounces > 2)
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->evShadowed[bounces - 3], 0));
        // (one accumulator: the shade launch of the last bounce has waited for the shadow rays of the bounce before it already)
    } else if (overlap) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->evShadowed[bounces - 1], 0));
        if (split && bounces > 1)
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->evShadowed[bounces - 2], 0)); // the other side stream's last pass
    }
    // the pass counters go to pinned memory from inside k_end_sample (a report still unread keeps its slot: the host reads it only once its event has fired)
    const bool report = c->passCountsPinned && !c->passCountsPending;
    hipLaunchKernelGGL(k_end_sample, dim3(1), dim3(64), 0, c->stream, c->control.p, c->totals.p, bounces, report ? c->passCountsPinned : nullptr);
    if (report) {
        HIPCHK(c, hipEventRecord(c->passCountsCopied, c->stream));
        c->passCountsPending = entries;
        c->passCountsEpoch = c->epoch;
    }
    c->batchSamples = batch;
    c->batchEntries = 0;
    c->foldPlanes = std::max(c->foldPlanes, batch); // folded once per pt_render (foldPlanesNow)
    HIPCHK(c, hipGetLastError());
    return PT_OK;
}

// Sum the extra accumulator planes into the accumulator proper and clear them.  Plane p holds sample p of every batch since
// the last fold -- still one live path per (plane, pixel) at any time -- so this runs once per pt_render call, not once per
// batch: 2 x 16 B x planes x owned pixels of traffic each time (3.7 ms at 1080p x 256 planes).
void foldPlanesNow(pt_ctx* c)
{
    if (c->mergePending) { // the shadow rays' own accumulator (one sample in flight) into the accumulator proper
        const uint32_t n = c->cfg.width * c->cfg.height;
        hipLaunchKernelGGL(k_merge_accum, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, c->accum, c->accumShadow.p, n, maxBounces(c));
        c->mergePending = false;
    }
    if (c->foldPlanes > 1) {
        const uint32_t n = c->numOwned;
        hipLaunchKernelGGL(k_fold_planes, dim3((uint32_t)(((uint64_t)n * kFoldLanes + 255) / 256)), dim3(256), 0, c->stream, accumView(c), c->foldPlanes,
            c->identityPixels ? nullptr : c->pixelList.p, n);
    }
    c->foldPlanes = 0;
}

// General schedule with slot refill (queue smaller than the number of owned pixels) and, in parity
// mode, exactly the reference's queue bookkeeping (raytracer.cpp:323-427): finished entries stay in
// the queue for one more pass, the loop ends when shade emits nothing and every pixel was issued.
// One 4-byte count read-back per pass, as the reference does.
int renderSampleRefill(pt_ctx* c, uint32_t sample)
{
    const FrameParams fp = frameParams(c, sample);
    Control* ctl = c->control.p;
    const uint32_t cap = c->capacity;
    uint32_t issued = 0, surviving = 0, pass = 0;
    int in = 0, out = 1;
    Control zero {};
    while (true) {
        // every pass reuses index 0/1 of the control block
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(64), 0, c->stream, &ctl->extCount[0], surviving);
        HIPCHK(c, hipMemsetAsync(&ctl->extCount[1], 0, sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(&ctl->shadowCount[0], 0, sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(&ctl->extCursor[0], 0, sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(&ctl->shadowCursor[0], 0, sizeof(uint32_t), c->stream));
        uint32_t newRays = 0;
        if (surviving != cap) {
            newRays = std::min(cap - surviving, c->numOwned - issued);
            if (newRays)
                launchGen(c, fp, in, issued, newRays, surviving, 0);
        }
        const uint32_t entries = surviving + newRays;
        launchIntersect(c, in, 0);
        launchShade(c, fp, in, out, 0, entries);
        uint32_t counts[2] = { 0, 0 };
        HIPCHK(c, hipMemcpyAsync(&counts[0], &ctl->extCount[1], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&counts[1], &ctl->shadowCount[0], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        issued += newRays;
        surviving = counts[0];
        pass++;
        if (counts[1] != 0)
            launchShadow(c, 0);
        // fold this pass into the totals (entries/ shadow counted on the host side of the loop)
        hipLaunchKernelGGL(k_end_sample, dim3(1), dim3(64), 0, c->stream, c->control.p, c->totals.p, 0u, (uint32_t*)nullptr);
        if (surviving == 0 && issued >= c->numOwned)
            break;
        std::swap(in, out);
        if (pass > 100000)
            return fail(c, PT_ERR_STATE, "refill loop did not terminate");
    }
    (void)zero;
    HIPCHK(c, hipGetLastError());
    return PT_OK;
}

} // namespace

// =================================================================================================
extern "C" {

const char* pt_version(void) { return "ptamd 0.1 (gfx950)"; }

#ifdef PT_TRACE_STATS
// diagnostic builds only: read and clear the traversal-loop counters
int pt_debug_trace_stats(unsigned long long* out, unsigned int n) // n <= 64 counters (pt_trace.h, g_traceStats)
{
    n = std::min(n, 64u);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceStats), sizeof(unsigned long long) * n) != hipSuccess)
        return -1;
    unsigned long long zero[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_traceStats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_createError.c_str(); }

// Test hook, no device needed: the host side of quantiseWideNode (pt_bake.h) -- up to four child boxes (lo / hi: 4 x 3 floats; empty[k] != 0:
// unused slot) into the 64-byte node the traversal kernels read.  tests/test_abi_and_layout.py checks the planes it makes against the
// expression the kernels evaluate (origin + 2^exp * q): every child box must lie inside its quantised box.
int pt_debug_quantise_node(const float* lo12, const float* hi12, const uint32_t* refs4, const uint8_t* empty4, uint32_t emptyRef, void* out64)
{
    if (!lo12 || !hi12 || !refs4 || !empty4 || !out64)
        return PT_ERR_INVALID;
    float lo[4][3], hi[4][3];
    bool empty[4];
    for (int k = 0; k < 4; k++) {
        empty[k] = empty4[k] != 0;
        for (int a = 0; a < 3; a++)
            lo[k][a] = lo12[k * 3 + a], hi[k][a] = hi12[k * 3 + a];
    }
    WideNode w;
    quantiseWideNode(lo, hi, refs4, empty, emptyRef, &w);
    std::memcpy(out64, &w, sizeof(w));
    return PT_OK;
}

int pt_create(const pt_config* cfg, pt_ctx** out)
{
    return guarded(nullptr, "pt_create", [&]() -> int {
    if (!cfg || !out)
        return fail(nullptr, PT_ERR_INVALID, "pt_create: null argument");
    if (cfg->width == 0 || cfg->height == 0 || (uint64_t)cfg->width * cfg->height > 0x7FFFFFFFull)
        return fail(nullptr, PT_ERR_INVALID, "pt_create: bad image size %ux%u", cfg->width, cfg->height);
    if (cfg->max_bounces > (uint32_t)kMaxPasses - 1)
        return fail(nullptr, PT_ERR_INVALID, "pt_create: max_bounces > %d", kMaxPasses - 1);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, PT_ERR_HIP, "pt_create: no HIP device (%s) -- this library has no CPU fallback", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= count)
        return fail(nullptr, PT_ERR_INVALID, "pt_create: device %d out of range (%d devices)", cfg->device, count);
    pt_ctx* c = new pt_ctx();
    c->cfg = *cfg;
    c->packetUse = (cfg->flags & PT_FLAG_NO_PACKETS) ? 0u : kPacketUseDefault;
    if (cfg->flags & PT_FLAG_PACKET_INTERSECT)
        c->packetUse |= 4u;
    if (const char* hs = getenv("PTAMD_SHADE_HEAD_SHIFT"))
        c->shadeHeadShift = (uint32_t)std::max(0, atoi(hs));
    if (const char* pk = getenv("PTAMD_PACKET")) // diagnostics: which launches may use k_trace_packet (bit 0 primary, 1 shadow, 2 pt_intersect)
        if (!(cfg->flags & PT_FLAG_NO_PACKETS))
            c->packetUse = (uint32_t)atoi(pk) | (c->packetUse & 4u);
    c->device = cfg->device;
    auto bail = [&](hipError_t err, const char* what) {
        int rc = fail(nullptr, PT_ERR_HIP, "pt_create: %s: %s", what, hipGetErrorString(err));
        delete c;
        return rc;
    };
    if ((e = hipSetDevice(c->device)) != hipSuccess)
        return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, c->device)) != hipSuccess)
        return bail(e, "hipGetDeviceProperties");
    c->numCUs = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate");
    c->ownStream = true;
    if ((e = hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate (copy stream)");
    if ((e = hipEventCreate(&c->evStart)) != hipSuccess || (e = hipEventCreate(&c->evStop)) != hipSuccess)
        return bail(e, "hipEventCreate");
    if ((e = hipStreamCreateWithFlags(&c->sideStream, hipStreamNonBlocking)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->sideStream2, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate (side stream)");
    for (int k = 0; k < kMaxPasses; k++)
        if ((e = hipEventCreateWithFlags(&c->evShaded[k], hipEventDisableTiming)) != hipSuccess
            || (e = hipEventCreateWithFlags(&c->evShadowed[k], hipEventDisableTiming)) != hipSuccess)
            return bail(e, "hipEventCreate");
    for (auto& d : c->dyn)
        if ((e = hipEventCreateWithFlags(&d.uploaded, hipEventDisableTiming)) != hipSuccess
            || (e = hipEventCreateWithFlags(&d.lastUse, hipEventDisableTiming)) != hipSuccess
            || (e = hipEventCreateWithFlags(&d.stageRead, hipEventDisableTiming)) != hipSuccess)
            return bail(e, "hipEventCreate");
    if ((e = hipHostMalloc((void**)&c->passCountsPinned, sizeof(c->passCountsHint), hipHostMallocDefault)) != hipSuccess
        || (e = hipEventCreateWithFlags(&c->passCountsCopied, hipEventDisableTiming)) != hipSuccess)
        return bail(e, "pinned counters");
    std::memset(c->passCountsPinned, 0, sizeof(c->passCountsHint));
    if ((e = hipHostMalloc((void**)&c->overflowPinned, sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
        return bail(e, "pinned overflow word");
    *c->overflowPinned = 0u;
    if ((e = c->totals.alloc(1)) != hipSuccess || (e = hipMemsetAsync(c->totals.p, 0, sizeof(Totals), c->stream)) != hipSuccess)
        return bail(e, "alloc totals");
    if ((e = c->accumOwn.alloc((size_t)cfg->width * cfg->height)) != hipSuccess
        || (e = hipMemsetAsync(c->accumOwn.p, 0, (size_t)cfg->width * cfg->height * sizeof(float4), c->stream)) != hipSuccess)
        return bail(e, "alloc accumulator");
    c->accum = c->accumOwn.p;
    c->numOwned = cfg->width * cfg->height;
    c->identityPixels = true;
    *out = c;
    int rc = pt_set_tiles(c, nullptr, 0);
    if (rc) {
        g_createError = c->error;
        pt_destroy(c);
        *out = nullptr;
        return rc;
    }
    return PT_OK;
    });
}

void pt_destroy(pt_ctx* c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    if (c->copyStream) // a never-adopted upload may still be copying into the sets released below
        (void)hipStreamSynchronize(c->copyStream);
    if (c->sideStream)
        (void)hipStreamSynchronize(c->sideStream);
    if (c->sideStream2)
        (void)hipStreamSynchronize(c->sideStream2);
    c->accumShadow.release();
    for (ShadowQueueBuf& q : c->shadowQ)
        q.o.release(), q.d.release(), q.c.release();
    DevBuf<float4>* f4[] = { &c->accumOwn, &c->hitH, &c->rays[0].o, &c->rays[0].d, &c->rays[0].thr, &c->rays[1].o,
        &c->rays[1].d, &c->rays[1].thr, &c->stagedRays.o, &c->stagedRays.d, &c->stagedRays.thr, &c->shadow.o, &c->shadow.d, &c->shadow.c,
        &c->stagedShadow.o, &c->stagedShadow.d, &c->stagedShadow.c };
    for (auto* b : f4)
        b->release();
    c->texMaterial.release(), c->texSky.release();
    c->nodes.release();
    for (StaticScene& sc : c->stat) {
        StaticScene::StaticGeom& g = sc.sg;
        sc.materials.release(), sc.triShade.release();
        g.dWide.release(), g.dBoxes.release(), g.dLeafOfs.release(), g.dRefTri.release(), g.dTris.release(), g.dFat.release();
        g.dVerts.release(), g.dNodes.release(), g.dKidBoxNode.release(), g.dExtra.release(), g.dParent.release(), g.dNeed.release(), g.dArrived.release();
        if (g.stage) (void)hipHostFree(g.stage);
        if (g.stageRead) (void)hipEventDestroy(g.stageRead);
    }
    for (auto& d : c->dyn) {
        d.wide.release(), d.tris.release(), d.fat.release(), d.instances.release(), d.lights.release(), d.jobs.release(), d.instFold.release(), d.instRootSrc.release();
        if (d.stage) (void)hipHostFree(d.stage);
        if (d.stageRead) (void)hipEventDestroy(d.stageRead);
        if (d.uploaded) (void)hipEventDestroy(d.uploaded);
        if (d.lastUse) (void)hipEventDestroy(d.lastUse);
    }
    if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
    c->pixelList.release(), c->hitInst.release();
    c->accumPlanes.release(), c->pixelOrdinal.release(), c->resolveTmp.release(), c->activeFlag.release(), c->streams.release(), c->control.release(), c->totals.release(), c->spill.release();
    for (hipEvent_t ev : c->profEvents)
        (void)hipEventDestroy(ev);
    if (c->sideStream) (void)hipStreamDestroy(c->sideStream);
    if (c->sideStream2) (void)hipStreamDestroy(c->sideStream2);
    for (int k = 0; k < kMaxPasses; k++) {
        if (c->evShaded[k]) (void)hipEventDestroy(c->evShaded[k]);
        if (c->evShadowed[k]) (void)hipEventDestroy(c->evShadowed[k]);
    }
    if (c->passCountsPinned) (void)hipHostFree(c->passCountsPinned);
    if (c->overflowPinned) (void)hipHostFree(c->overflowPinned);
    if (c->passCountsCopied) (void)hipEventDestroy(c->passCountsCopied);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->ownStream && c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

int pt_set_stream(pt_ctx* c, void* hip_stream)
{
    if (!c)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ownStream && c->stream)
        (void)hipStreamDestroy(c->stream);
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->ownStream = false;
    } else {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->ownStream = true;
    }
    return PT_OK;
}

// `async`: into c->st without touching the render stream or the dynamic state (pt_upload_static_async: c->st is the scene that is not current)
static int uploadStaticImpl(pt_ctx* c, const pt_vertex* verts, uint32_t nV, const pt_triangle* tris, uint32_t nT, const pt_material* mats,
    uint32_t nM, const pt_sub_bvh_node* nodes, uint32_t nN, bool async = false)
{
    {
    if (!c)
        return PT_ERR_INVALID;
    if (!verts || !tris || !mats || !nodes || nV == 0 || nT == 0 || nM == 0 || nN == 0)
        return fail(c, PT_ERR_INVALID, "pt_upload_static: empty or null scene array");
    if (nT > kRefIndexMask || nN > kRefIndexMask)
        return fail(c, PT_ERR_UNSUPPORTED, "pt_upload_static: more than 2^27 triangle references or nodes");
    HIPCHK(c, hipSetDevice(c->device));
    StageTimer tm("uploadStatic");
    // ---- validate: every index in range, children after their parent (rules out cycles) -----
    for (uint32_t t = 0; t < nT; t++) {
        if (tris[t].indices[0] >= nV || tris[t].indices[1] >= nV || tris[t].indices[2] >= nV)
            return fail(c, PT_ERR_INVALID, "triangle %u: vertex index out of range", t);
        if (tris[t].materialIndex >= nM)
            return fail(c, PT_ERR_INVALID, "triangle %u: material index out of range", t);
    }
    // An inner node whose children do not lie strictly after it is an unused pad (the reference's pair
    // allocator leaves one next to every root); pads may not be referenced.
    auto isPad = [&](uint32_t i) {
        const uint32_t l = nodes[i].leftChildOrFirstTriangle;
        return nodes[i].triangleCount == 0 && (l <= i || (uint64_t)l + 1 >= nN);
    };
    for (uint32_t i = 0; i < nN; i++) {
        const pt_sub_bvh_node& n = nodes[i];
        if (n.triangleCount != 0 && (uint64_t)n.leftChildOrFirstTriangle + n.triangleCount > nT)
            return fail(c, PT_ERR_INVALID, "sub-BVH leaf %u: triangle range out of bounds", i);
    }

    tm.lap("validate");
    // ---- triangles / vertices / materials ------------------------------------------------------
    std::vector<TriIsect> hTris(nT);
    std::vector<TriShade> hShade(nT);
    auto P = [&](uint32_t vi) { return mk(verts[vi].vertex[0], verts[vi].vertex[1], verts[vi].vertex[2]); };
    for (uint32_t t = 0; t < nT; t++) {
        const V3 v0 = P(tris[t].indices[0]);
        const V3 e1 = P(tris[t].indices[1]) - v0, e2 = P(tris[t].indices[2]) - v0; // shapes.cl:37-38
        hTris[t].a = make_float4(v0.x, v0.y, v0.z, e1.x);
        hTris[t].b = make_float4(e1.y, e1.z, e2.x, e2.y);
        hTris[t].c = make_float4(e2.z, 0.f, 0.f, 0.f);
        hShade[t] = { tris[t].indices[0], tris[t].indices[1], tris[t].indices[2], tris[t].materialIndex };
    }
    std::vector<VertexShade> hVerts(nV);
    for (uint32_t v = 0; v < nV; v++) {
        hVerts[v].n_u = make_float4(verts[v].normal[0], verts[v].normal[1], verts[v].normal[2], verts[v].texCoord[0]);
        hVerts[v].v_pad = make_float4(verts[v].texCoord[1], 0.f, 0.f, 0.f);
    }
    std::vector<Material> hMats(nM);
    static_assert(sizeof(Material) == sizeof(pt_material), "material record is copied verbatim");
    std::memcpy(hMats.data(), mats, (size_t)nM * sizeof(pt_material));

    tm.lap("records");
    // ---- pair nodes ----------------------------------------------------------------------------
    std::vector<uint32_t> dense(nN, 0xFFFFFFFFu);
    uint32_t numInner = 0;
    for (uint32_t i = 0; i < nN; i++)
        if (nodes[i].triangleCount == 0 && !isPad(i))
            dense[i] = numInner++;
    std::vector<PairNode> hNodes(numInner);
    auto triBox = [&](uint32_t t, V3& lo, V3& hi) {
        for (int k = 0; k < 3; k++) {
            const V3 p = P(tris[t].indices[k]);
            lo = mk(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
            hi = mk(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
        }
    };
    // Leaves larger than `maxLeaf` triangles become a small subtree over their triangle range.  Round 5: maxLeaf = 2, not the 30 a reference can
    // address -- the reference's builders stop at <= 3 triangles (src/bvh/bvh_build.cpp:15: 60 % of the benchmark meshes' leaves hold two, 39 % three),
    // and a leaf step of the traversal kernels runs to the LONGEST leaf among its lanes: with the three-triangle leaves cut into 1 + 2 at the cheaper
    // of the two places (the pieces go into free slots of the 4-wide nodes where there are any: 27 k -> 37 k nodes for 82 k triangles) every leaf step
    // is two trips at most.  Measured on the benchmark (one box, A / B / A): 10 868 -> 11 012 -> 10 832 Mrays/s (+1.5 %; leaves of ONE triangle: -0.7 %;
    // merging subtrees into leaves of up to 4 / 6 / 8 instead: -1.0 / -3.3 / -3.4 %, profiles/round5/r5_tree_shape.txt).  Parity mode keeps the caller's
    // leaves (its order of triangle tests is the reference's).  PTAMD_MAX_LEAF=n overrides (diagnostics).
#ifndef PT_MAX_LEAF
#define PT_MAX_LEAF 2
#endif
    uint32_t maxLeaf = parityMode(c) ? kMaxLeafTris : std::min<uint32_t>(PT_MAX_LEAF, kMaxLeafTris);
    if (const char* e = getenv("PTAMD_MAX_LEAF"))
        maxLeaf = std::max(1u, std::min((uint32_t)atoi(e), kMaxLeafTris));
    struct Range {
        uint32_t first, count;
    };
    auto rangeBox = [&](Range r, V3& lo, V3& hi) {
        lo = mk(FLT_MAX), hi = mk(-FLT_MAX);
        for (uint32_t t = 0; t < r.count; t++)
            triBox(r.first + t, lo, hi);
    };
    auto halfArea = [](V3 lo, V3 hi) {
        const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return dx * dy + dy * dz + dz * dx;
    };
    std::function<uint32_t(Range, V3&, V3&)> leafRef = [&](Range r, V3& lo, V3& hi) -> uint32_t {
        lo = mk(FLT_MAX), hi = mk(-FLT_MAX);
        if (r.count <= maxLeaf) {
            for (uint32_t t = 0; t < r.count; t++)
                triBox(r.first + t, lo, hi);
            return makeRef(r.first, r.count);
        }
        // the range stays in the caller's order (a leaf is a run of it): cut where the two runs' surface-area cost is smallest
        uint32_t half = r.count / 2;
        if (r.count <= 8u) {
            float best = FLT_MAX;
            for (uint32_t cut = 1; cut < r.count; cut++) {
                V3 alo, ahi, blo, bhi;
                rangeBox({ r.first, cut }, alo, ahi);
                rangeBox({ r.first + cut, r.count - cut }, blo, bhi);
                const float cost = halfArea(alo, ahi) * (float)cut + halfArea(blo, bhi) * (float)(r.count - cut);
                if (cost < best)
                    best = cost, half = cut;
            }
        }
        V3 llo, lhi, rlo, rhi;
        const uint32_t l = leafRef({ r.first, half }, llo, lhi);
        const uint32_t rr = leafRef({ r.first + half, r.count - half }, rlo, rhi);
        PairNode pn {};
        pn.bx = make_float4(llo.x, lhi.x, rlo.x, rhi.x);
        pn.by = make_float4(llo.y, lhi.y, rlo.y, rhi.y);
        pn.bz = make_float4(llo.z, lhi.z, rlo.z, rhi.z);
        pn.left = l;
        pn.right = rr;
        lo = mk(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z));
        hi = mk(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z));
        hNodes.push_back(pn);
        return makeRef((uint32_t)hNodes.size() - 1, 0);
    };
    c->st->nodeRef.assign(nN, kRefNone);
    for (uint32_t i = 0; i < nN; i++) {
        if (nodes[i].triangleCount != 0) {
            if (nodes[i].triangleCount <= maxLeaf) {
                c->st->nodeRef[i] = makeRef(nodes[i].leftChildOrFirstTriangle, nodes[i].triangleCount);
            } else {
                V3 lo, hi;
                c->st->nodeRef[i] = leafRef({ nodes[i].leftChildOrFirstTriangle, nodes[i].triangleCount }, lo, hi);
            }
        } else if (dense[i] != 0xFFFFFFFFu) {
            c->st->nodeRef[i] = makeRef(dense[i], 0);
        }
    }
    for (uint32_t i = 0; i < nN; i++) {
        if (dense[i] == 0xFFFFFFFFu)
            continue;
        const uint32_t l = nodes[i].leftChildOrFirstTriangle;
        const pt_sub_bvh_node& L = nodes[l];
        const pt_sub_bvh_node& R = nodes[l + 1];
        PairNode pn {};
        pn.bx = make_float4(L.min[0], L.max[0], R.min[0], R.max[0]);
        pn.by = make_float4(L.min[1], L.max[1], R.min[1], R.max[1]);
        pn.bz = make_float4(L.min[2], L.max[2], R.min[2], R.max[2]);
        pn.left = c->st->nodeRef[l];
        pn.right = c->st->nodeRef[l + 1];
        if (pn.left == kRefNone || pn.right == kRefNone)
            return fail(c, PT_ERR_INVALID, "sub-BVH node %u: child is an unused pad node", i);
        hNodes[dense[i]] = pn;
    }
    // depth of every subtree (children have larger indices: one reverse sweep), for the stack bound
    c->st->subtreeDepth.assign(nN, 0);
    for (uint32_t i = nN; i-- > 0;) {
        if (dense[i] == 0xFFFFFFFFu) {
            uint32_t extra = 0;
            for (uint32_t cnt = nodes[i].triangleCount; cnt > maxLeaf; cnt = cnt > 8u ? (cnt + 1) / 2 : cnt - 1) // (the cost-driven cut of a short run may peel one triangle off per level)
                extra++;
            c->st->subtreeDepth[i] = extra;
        } else {
            const uint32_t l = nodes[i].leftChildOrFirstTriangle;
            c->st->subtreeDepth[i] = 1 + std::max(c->st->subtreeDepth[l], c->st->subtreeDepth[l + 1]);
        }
    }
    if (hNodes.size() > kRefIndexMask)
        return fail(c, PT_ERR_UNSUPPORTED, "too many BVH nodes");

    tm.lap("pairNodes");
    int rc;
    c->st->hostTris = hTris;
    c->st->hostBottomNodes = std::move(hNodes);
    c->st->rawVerts.assign(verts, verts + nV);
    c->st->hostGeomStale = false;
    c->st->sg.latestInStage = false;
    c->st->denseOfNode = dense;
    c->st->numDensePairs = numInner;
    if (!async)
        HIPCHK(c, hipStreamSynchronize(c->stream)); // renders in flight read these buffers
    if ((rc = uploadVec(c, c->st->materials, hMats)))
        return rc;
    c->st->hostTriShade = std::move(hShade);
    c->st->hostMaterials.assign(mats, mats + nM);
    c->st->hostSubNodes.assign(nodes, nodes + nN);
    c->st->hostVerts = std::move(hVerts);
    c->st->numVerts = nV;
    c->st->numRefNodes = nN;
    c->st->numTris = nT;
    c->st->have = true;
    c->st->hostNodeBoxesStale = false;
    if (!async) {
        c->haveStatic = true;
        c->haveDynamic = false; // top-level leaves reference sub-BVH roots: must be re-uploaded
        c->pending = -1;
        c->statPending = -1; // (a rebuilt scene that was waiting for its frame tick is dropped with the dynamic state)
    }
    {   // material types in use (emissive surfaces end a path in a few instructions: they do not count)
        uint32_t types = 0;
        for (uint32_t t = 0; t < nT; t++) {
            uint32_t ty;
            std::memcpy(&ty, (const char*)&mats[tris[t].materialIndex] + 32, 4); // the type word of the 48-byte record (Material::typeAndPad.x)
            types |= 1u << std::min(ty, 31u);
        }
        types &= ~(1u << MAT_EMISSIVE);
        c->st->materialBins = (types & (types - 1u)) != 0u && (c->cfg.flags & PT_FLAG_MATERIAL_BINS) != 0u; // opt-in: measured slower (pt_shade.h)
    }
    c->st->sg.extraRoots.clear();
    tm.lap("mirrors");
    if ((rc = buildStaticGeom(c)))
        return rc;
    tm.lap("buildStaticGeom");
    if (!async)
        refreshSceneView(c);
    return PT_OK;
    }
}

// A REBUILT scene per frame as a frame-loop citizen (the other branch of MeshSequence::buildBvh, reference src/model/mesh_sequence.cpp:89-96, whose
// result transferDynamicData uploads every tick, src/raytracer.cpp:510-568): the same arrays as pt_upload_static, converted into the context's
// SECOND static scene and copied to the device without touching the render stream -- the frames enqueued so far, and any enqueued before the flip,
// keep rendering the old trees.  Follow with pt_upload_dynamic_async (lights and top level OF THE NEW SCENE: its leaves name the new sub-BVH roots)
// and pt_frame_tick, which adopts both.  pt_update_geometry / pt_refit_vertices keep addressing the current scene until then.
int pt_upload_static_async(pt_ctx* c, const pt_vertex* verts, uint32_t nV, const pt_triangle* tris, uint32_t nT, const pt_material* mats,
    uint32_t nM, const pt_sub_bvh_node* nodes, uint32_t nN)
{
    return guarded(c, "pt_upload_static_async", [&]() -> int {
        if (!c)
            return PT_ERR_INVALID;
        if (!c->haveStatic || !c->haveDynamic) // nothing is rendering yet: the plain upload does
            return uploadStaticImpl(c, verts, nV, tris, nT, mats, nM, nodes, nN);
        HIPCHK(c, hipSetDevice(c->device));
        // the other scene's master copies may still be read by copies the copy stream has queued (a dynamic set refreshing itself from them,
        // world-space copy kernels): that stream only -- the render stream is never waited for
        HIPCHK(c, hipStreamSynchronize(c->copyStream));
        const int target = 1 - c->statCur;
        if (c->pending >= 0 && c->dyn[c->pending].staticIndex == target)
            c->pending = -1; // a dynamic state built on the scene that is about to be replaced: never adopted
        c->statPending = -1;
        StaticScene* const cur = c->st;
        c->st = &c->stat[target];
        int rc = uploadStaticImpl(c, verts, nV, tris, nT, mats, nM, nodes, nN, true);
        if (rc == PT_OK)
            rc = uploadStaticGeom(c);
        if (rc != PT_OK)
            c->st->have = false;
        c->st = cur;
        if (rc == PT_OK)
            c->statPending = target;
        return rc;
    });
}

int pt_upload_static(pt_ctx* c, const pt_vertex* verts, uint32_t nV, const pt_triangle* tris, uint32_t nT, const pt_material* mats,
    uint32_t nM, const pt_sub_bvh_node* nodes, uint32_t nN)
{
    return guarded(c, "pt_upload_static", [&]() -> int { return uploadStaticImpl(c, verts, nV, tris, nT, mats, nM, nodes, nN); });
}

// New vertex data and refitted boxes for an UNCHANGED topology (what refitBVH, src/bvh/refit_bvh.cpp:6-34, leaves of a deforming
// mesh; the reference rewrites the dynamic tail of its vertex and sub-BVH buffers every tick, src/raytracer.cpp:510-568).  The
// arrays are the caller's whole vertex and sub-BVH arrays again; triangles, materials and node links must be the ones uploaded.
// Takes effect with the next pt_upload_dynamic(_async) + pt_frame_tick; renders in flight are not disturbed.
//
// A refit cannot change what the conversion of pt_upload_static decided -- which descendants of a binary node became the children of
// its 4-wide node (the collapse's split choices), the breadth-first packing, the leaves' triangle references, the stack bound -- so all
// of that is kept (StaticScene::StaticGeom::kidSrc / kidBoxNode / kidEmpty / leafOfs / refTri / stackNeed) and only what moves is re-made, ON THE
// DEVICE: the caller's vertex and node arrays travel as they are through pinned staging on the copy stream (an event of its own guards
// the staging memory: no stream is synchronised), k_refit_nodes gathers every packed node's child boxes from the caller's nodes and
// re-quantises it (quantiseWideNode: the host's routine), k_refit_tris re-makes the triangles' intersection and shading records.  The
// host's half is the topology check and two copies into pinned memory; its own mirrors of the converted arrays go stale and are
// refreshed from the staging memory only if the whole conversion ever runs again.  (Round 3 re-ran the whole conversion here --
// collapse, packing, 128 bytes of shading record per triangle on one host thread -- and re-uploaded everything behind a
// synchronisation of the copy stream.)
int pt_update_geometry(pt_ctx* c, const pt_vertex* verts, uint32_t nV, const pt_sub_bvh_node* nodes, uint32_t nN)
{
    return guarded(c, "pt_update_geometry", [&]() -> int {
        if (!c)
            return PT_ERR_INVALID;
        if (!c->haveStatic)
            return fail(c, PT_ERR_STATE, "pt_update_geometry: call pt_upload_static first");
        if (c->statPending >= 0) // (the caller's offsets and counts are those of the rebuilt scene already; the scene that still renders is the old one)
            return fail(c, PT_ERR_STATE, "pt_update_geometry: a rebuilt scene (pt_upload_static_async) is waiting for pt_frame_tick: refit after the tick");
        if (!verts || !nodes || nV != c->st->numVerts || nN != c->st->numRefNodes)
            return fail(c, PT_ERR_INVALID, "pt_update_geometry: the vertex and node counts must be the uploaded ones (%u, %u)", c->st->numVerts, c->st->numRefNodes);
        for (uint32_t i = 0; i < nN; i++)
            if (nodes[i].leftChildOrFirstTriangle != c->st->hostSubNodes[i].leftChildOrFirstTriangle || nodes[i].triangleCount != c->st->hostSubNodes[i].triangleCount)
                return fail(c, PT_ERR_INVALID, "pt_update_geometry: node %u changed its links: a refit keeps the topology (use pt_upload_static for a rebuilt tree)", i);
        HIPCHK(c, hipSetDevice(c->device));
        StaticScene::StaticGeom& g = c->st->sg;
        if (!g.onDevice) {
            // nothing of the old geometry is on the device yet: everything on the host, the first upload takes it from the host arrays
            refreshHostGeometry(c); // (an earlier device-side refit may have left the host mirrors behind)
            refitPairBoxes(c, verts, nodes, false);
            refitWideOnHost(c);
            c->st->rawVerts.assign(verts, verts + nV);
            c->st->hostSubNodes.assign(nodes, nodes + nN);
            g.latestInStage = false;
            c->st->hostGeomStale = true; // hostTris / hostVerts / sg.fat follow at the upload
            g.version = ++c->staticVersions;
            return PT_OK;
        }
        // ---- the device's master copy: the caller's vertices and nodes AS THEY ARE through pinned staging on the copy stream; the packed nodes
        // re-quantised (k_refit_nodes) and the triangle records re-made (k_refit_tris) there.  The staging memory doubles as the host's copy of the
        // caller's latest arrays (refreshHostGeometry reads it if the whole conversion ever runs again).
        static_assert(sizeof(VertexIn) == sizeof(pt_vertex), "k_refit_tris reads the caller's vertex records as they are");
        static_assert(sizeof(SubNodeIn) == sizeof(pt_sub_bvh_node), "k_refit_nodes reads the caller's node records as they are");
        std::vector<float> extra; // boxes of the pair nodes that split an oversized leaf (rare): recomputed here
        if (c->st->hostBottomNodes.size() > c->st->numDensePairs) {
            refitPairBoxes(c, verts, nodes, true);
            for (size_t j = c->st->numDensePairs; j < c->st->hostBottomNodes.size(); j++) {
                const PairNode& n = c->st->hostBottomNodes[j];
                const float b[12] = { n.bx.x, n.by.x, n.bz.x, n.bx.y, n.by.y, n.bz.y, n.bx.z, n.by.z, n.bz.z, n.bx.w, n.by.w, n.bz.w };
                extra.insert(extra.end(), b, b + 12);
            }
        }
        const size_t bytesV = (size_t)nV * sizeof(pt_vertex), bytesN = (size_t)nN * sizeof(pt_sub_bvh_node), bytesE = extra.size() * sizeof(float);
        const size_t total = bytesV + bytesN + bytesE;
        if (!g.stageRead)
            HIPCHK(c, hipEventCreateWithFlags(&g.stageRead, hipEventDisableTiming));
        if (g.stageBusy) { // the previous refit's copies out of the staging memory (normally long done)
            HIPCHK(c, hipEventSynchronize(g.stageRead));
            g.stageBusy = false;
        }
        if (g.stageBytes < total) {
            if (g.latestInStage)
                refreshHostGeometry(c); // (the staging memory holds the host's only copy of the latest arrays: take it over before it goes)
            if (g.stage)
                (void)hipHostFree(g.stage);
            g.stage = nullptr;
            g.stageBytes = total + total / 8;
            HIPCHK(c, hipHostMalloc(&g.stage, g.stageBytes, hipHostMallocDefault));
        }
        unsigned char* st = (unsigned char*)g.stage;
        std::memcpy(st, verts, bytesV);
        std::memcpy(st + bytesV, nodes, bytesN);
        g.latestInStage = true;
        c->st->hostGeomStale = true;
        g.version = ++c->staticVersions;
        HIPCHK(c, hipMemcpyAsync(g.dVerts.p, st, bytesV, hipMemcpyHostToDevice, c->copyStream));
        HIPCHK(c, hipMemcpyAsync(g.dNodes.p, st + bytesV, bytesN, hipMemcpyHostToDevice, c->copyStream));
        if (bytesE) {
            std::memcpy(st + bytesV + bytesN, extra.data(), bytesE);
            if (g.dExtra.n < extra.size())
                HIPCHK(c, g.dExtra.alloc(extra.size()));
            HIPCHK(c, hipMemcpyAsync(g.dExtra.p, st + bytesV + bytesN, bytesE, hipMemcpyHostToDevice, c->copyStream));
        }
        HIPCHK(c, hipEventRecord(g.stageRead, c->copyStream));
        g.stageBusy = true;
        if (!g.wide.empty()) {
            RefitNodeArgs rn {};
            rn.nodes = (const SubNodeIn*)g.dNodes.p, rn.kidBoxNode = g.dKidBoxNode.p, rn.extra = g.dExtra.p, rn.wide = g.dWide.p, rn.boxes = g.dBoxes.p;
            rn.emptyRef = g.emptyRef, rn.n = (uint32_t)g.wide.size();
            hipLaunchKernelGGL(k_refit_nodes, dim3((rn.n + 127u) / 128u), dim3(128), 0, c->copyStream, rn);
        }
        RefitArgs ra {};
        ra.verts = (const VertexIn*)g.dVerts.p, ra.tri = c->st->triShade.p, ra.mats = c->st->materials.p, ra.tris = g.dTris.p, ra.fat = g.dFat.p, ra.n = c->st->numTris;
        hipLaunchKernelGGL(k_refit_tris, dim3((c->st->numTris + 255u) / 256u), dim3(256), 0, c->copyStream, ra);
        HIPCHK(c, hipGetLastError());
        return PT_OK;
    });
}

// Who a packed node reports to in a bottom-up pass, and how many arrivals complete it (k_refit_tree).  Made once per topology.
static int ensureRefitTables(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    if (g.refitTablesFor == g.topology)
        return PT_OK;
    const size_t n = g.wide.size();
    std::vector<uint32_t> parent(n, 0xFFFFFFFFu), need(n, 1u);
    bool ok = true;
    for (size_t q = 0; q < n; q++)
        for (int k = 0; k < 4; k++) {
            const uint32_t r = g.wide[q].child[k];
            if (r == g.emptyRef || refCount(r) != 0u)
                continue;
            const uint32_t ch = refIndex(r);
            if (ch >= n || ch == q || parent[ch] != 0xFFFFFFFFu) { // (two parents: roots that share a subtree -- the bottom-up pass would complete the child once and leave one parent waiting)
                ok = false;
                continue;
            }
            parent[ch] = ((uint32_t)q << 2) | (uint32_t)k;
            need[q]++;
        }
    g.refitTablesOk = ok;
    g.refitTablesFor = g.topology;
    if (!ok)
        return PT_OK;
    HIPCHK(c, hipStreamSynchronize(c->copyStream)); // (an earlier refit may still be walking the old tables)
    int rc;
    if ((rc = uploadVec(c, g.dParent, parent)) || (rc = uploadVec(c, g.dNeed, need)))
        return rc;
    if (!g.dArrived.p || g.dArrived.n < std::max<size_t>(n, 1))
        HIPCHK(c, g.dArrived.alloc(std::max<size_t>(n + n / 8, 1)));
    HIPCHK(c, hipMemset(g.dArrived.p, 0, g.dArrived.n * sizeof(uint32_t)));
    return PT_OK;
}

// A deformed frame of the same topology, refitted ON THE DEVICE: the caller hands over the vertices of the mesh that moved -- `nV` records that
// replace [firstVertex, firstVertex + nV) of the vertex array pt_upload_static took -- and nothing else.  The device re-makes the triangles'
// intersection and shading records (k_refit_tris) and recomputes every box of its packed trees bottom-up from the triangles (k_refit_tree:
// what refitBVH does on the host in the reference, src/bvh/refit_bvh.cpp:6-34, and what pt_update_geometry expects the caller to have done),
// on the copy stream, nothing synchronised.  Takes effect with the next pt_upload_dynamic(_async) + pt_frame_tick, like pt_update_geometry.
// The host's share of a tick is one copy of the moved vertices into pinned memory.
int pt_refit_vertices(pt_ctx* c, uint32_t firstVertex, const pt_vertex* verts, uint32_t nV)
{
    return guarded(c, "pt_refit_vertices", [&]() -> int {
        if (!c)
            return PT_ERR_INVALID;
        if (!c->haveStatic)
            return fail(c, PT_ERR_STATE, "pt_refit_vertices: call pt_upload_static first");
        if (c->statPending >= 0) // (a refit between a rebuild and its tick addressed the OLD scene with the NEW scene's offsets)
            return fail(c, PT_ERR_STATE, "pt_refit_vertices: a rebuilt scene (pt_upload_static_async) is waiting for pt_frame_tick: refit after the tick");
        if (!verts || nV == 0 || (uint64_t)firstVertex + nV > c->st->numVerts)
            return fail(c, PT_ERR_INVALID, "pt_refit_vertices: [%u, %u + %u) is not a range of the %u uploaded vertices", firstVertex, firstVertex, nV, c->st->numVerts);
        HIPCHK(c, hipSetDevice(c->device));
        StaticScene::StaticGeom& g = c->st->sg;
        int rc;
        if ((rc = uploadStaticGeom(c)) || (rc = ensureRefitTables(c))) // (the master copy reaches the device with the first dynamic upload at the latest)
            return rc;
        if (!g.refitTablesOk)
            return fail(c, PT_ERR_UNSUPPORTED, "pt_refit_vertices: two roots of the sub-BVH array share a subtree: refit on the host and use pt_update_geometry");
        // the host's copy of the caller's arrays: rawVerts takes the new vertices; the node boxes go stale (nobody refits them here) and are
        // recomputed from the vertices only if the whole conversion ever runs again (refreshHostGeometry)
        if (g.latestInStage) { // an earlier pt_update_geometry left the latest arrays in the staging memory: take them over first
            if (g.stageBusy) {
                HIPCHK(c, hipEventSynchronize(g.stageRead));
                g.stageBusy = false;
            }
            const pt_vertex* sv = latestVerts(c);
            const pt_sub_bvh_node* sn = latestNodes(c);
            c->st->rawVerts.assign(sv, sv + c->st->numVerts);
            c->st->hostSubNodes.assign(sn, sn + c->st->numRefNodes);
            g.latestInStage = false;
        }
        std::memcpy(c->st->rawVerts.data() + firstVertex, verts, (size_t)nV * sizeof(pt_vertex));
        c->st->hostNodeBoxesStale = true;
        c->st->hostGeomStale = true;
        const size_t bytesV = (size_t)nV * sizeof(pt_vertex);
        if (!g.stageRead)
            HIPCHK(c, hipEventCreateWithFlags(&g.stageRead, hipEventDisableTiming));
        if (g.stageBusy) { // the previous refit's copy out of the staging memory (normally long done)
            HIPCHK(c, hipEventSynchronize(g.stageRead));
            g.stageBusy = false;
        }
        if (g.stageBytes < bytesV) {
            if (g.stage)
                (void)hipHostFree(g.stage);
            g.stage = nullptr;
            g.stageBytes = bytesV + bytesV / 8;
            HIPCHK(c, hipHostMalloc(&g.stage, g.stageBytes, hipHostMallocDefault));
        }
        std::memcpy(g.stage, verts, bytesV);
        g.version = ++c->staticVersions;
        HIPCHK(c, hipMemcpyAsync(g.dVerts.p + firstVertex, g.stage, bytesV, hipMemcpyHostToDevice, c->copyStream));
        HIPCHK(c, hipEventRecord(g.stageRead, c->copyStream));
        g.stageBusy = true;
        static_assert(sizeof(VertexIn) == sizeof(pt_vertex), "the refit kernels read the caller's vertex records as they are");
        if (!g.wide.empty()) {
            RefitTreeArgs rt {};
            rt.verts = (const VertexIn*)g.dVerts.p, rt.tri = c->st->triShade.p, rt.wide = g.dWide.p, rt.boxes = g.dBoxes.p;
            rt.parent = g.dParent.p, rt.need = g.dNeed.p, rt.arrived = g.dArrived.p, rt.emptyRef = g.emptyRef, rt.n = (uint32_t)g.wide.size();
            hipLaunchKernelGGL(k_refit_tree, dim3((rt.n + 127u) / 128u), dim3(128), 0, c->copyStream, rt);
        }
        RefitArgs ra {};
        ra.verts = (const VertexIn*)g.dVerts.p, ra.tri = c->st->triShade.p, ra.mats = c->st->materials.p, ra.tris = g.dTris.p, ra.fat = g.dFat.p, ra.n = c->st->numTris;
        hipLaunchKernelGGL(k_refit_tris, dim3((c->st->numTris + 255u) / 256u), dim3(256), 0, c->copyStream, ra);
        HIPCHK(c, hipGetLastError());
        return PT_OK;
    });
}

} // extern "C"

namespace {

// What the host-side conversion of one dynamic state produces (no device call in it): the top level, the instance table, the lights,
// and the list of world-space copies the device is to make.  Everything below the top level is static (StaticScene::StaticGeom).
struct DynamicHost {
    std::vector<WideNode> topWide; // goes to wide[staticNodes ...]
    std::vector<Instance> instances;
    std::vector<Light> lights;
    std::vector<BakeJob> jobs;
    std::vector<uint32_t> instanceTopNode;
    uint32_t numLights = 0, rootRef = 0, rootRefFolded = 0;
    uint32_t foldedInstances = 0; // instances the per-ray kernels walk without parking (translation + uniform scale, pt_trace.h)
    std::vector<WideNode> instRoots; // their copies of their meshes' root nodes (one slot per instance), stored at instRootBase: the last run of the node array
    std::vector<uint32_t> instRootSrc; // per instance: the packed node its root copy is made from ON THE DEVICE (k_inst_roots), ~0: instRoots[k] holds it already
    std::vector<float4> instFold; // entry 1 + k: (1 / s, w) of instance k; entry 0 and the instances on the general route: the identity
    uint32_t instRootBase = 0;
    uint32_t topSlots = 0; // node slots reserved for the top level (the copies start behind them)
    uint32_t bakedNodes = 0, bakedTris = 0;
    bool packetOk = false, hasInstances = false, generalRoute = false;
    uint32_t stackNeed = 0, enteredInstances = 0;
};

int convertDynamic(pt_ctx* c, const pt_emissive_triangle* lights, uint32_t nL, const pt_top_bvh_node* topNodes, uint32_t nTop, uint32_t topRoot, DynamicHost& out)
{
    StaticScene::StaticGeom& sg = c->st->sg;
    // a top-level leaf may name any node of the caller's sub-BVH array; the ones that are not mesh roots become roots of their own
    {
        bool grown = false;
        for (uint32_t i = 0; i < nTop; i++) {
            const pt_top_bvh_node& n = topNodes[i];
            if (!n.isLeaf)
                continue;
            if (n.a >= c->st->numRefNodes || c->st->nodeRef[n.a] == kRefNone)
                return fail(c, PT_ERR_INVALID, "top-level leaf %u: sub-BVH root %u is not a valid node", i, n.a);
            if (sg.rootOfNode[n.a] < 0 && std::find(sg.extraRoots.begin(), sg.extraRoots.end(), n.a) == sg.extraRoots.end()) {
                sg.extraRoots.push_back(n.a);
                grown = true;
            }
        }
        if (grown) {
            int rc = buildStaticGeom(c);
            if (rc)
                return rc;
        }
    }
    const uint32_t staticNodes = (uint32_t)sg.wide.size();
    const uint32_t staticTris = c->st->numTris + 1u; // the caller's triangles + the all-zero one
    // ---- instances (one per top-level leaf) and top-level pair nodes (one per top-level inner node)
    std::vector<Instance>& hInst = out.instances;
    hInst.clear();
    std::vector<uint32_t> topRef(nTop, kRefNone); // reference of top node i as a child
    std::vector<int32_t> instRoot; // instance -> roots[] slot
    out.instanceTopNode.clear();
    out.jobs.clear();
    uint32_t numTopInner = 0, maxBottomDepth = 0;
    for (uint32_t i = 0; i < nTop; i++) {
        const pt_top_bvh_node& n = topNodes[i];
        if (n.isLeaf) {
            maxBottomDepth = std::max(maxBottomDepth, c->st->subtreeDepth[n.a] + 1);
            const StaticScene::StaticGeom::Root& root = sg.roots[sg.rootOfNode[n.a]];
            Instance in {};
            const float* m = n.invTransform; // column-major
            in.r0 = make_float4(m[0], m[4], m[8], m[12]);
            in.r1 = make_float4(m[1], m[5], m[9], m[13]);
            in.r2 = make_float4(m[2], m[6], m[10], m[14]);
            in.rootRef = root.ref;
            in.topNode = i;
            {   // a translation + uniform scale?  (parity mode follows the reference's route to the letter)
                const float a = in.r0.x;
                in.simple = (!parityMode(c) && !(c->cfg.flags & PT_FLAG_PARKED_INSTANCES) && a > 0.f && std::isfinite(a) && in.r1.y == a && in.r2.z == a && in.r0.y == 0.f
                                && in.r0.z == 0.f && in.r1.x == 0.f && in.r1.z == 0.f && in.r2.x == 0.f && in.r2.y == 0.f && std::isfinite(in.r0.w) && std::isfinite(in.r1.w)
                                && std::isfinite(in.r2.w))
                    ? 1u : 0u;
            }
            if (hInst.size() >= kSpecialLeaveInstance)
                return fail(c, PT_ERR_UNSUPPORTED, "too many instances");
            topRef[i] = makeRef((uint32_t)hInst.size(), kRefSpecial);
            hInst.push_back(in);
            instRoot.push_back(sg.rootOfNode[n.a]);
            out.instanceTopNode.push_back(i);
        } else {
            if (n.a >= nTop || n.b >= nTop)
                return fail(c, PT_ERR_INVALID, "top-level node %u: child out of range", i);
            topRef[i] = makeRef(staticNodes + numTopInner, 0u);
            numTopInner++;
        }
    }
    // node slots of the top level: [the top level with instance references (<= numTopInner nodes) | the same top level for the per-ray kernels, which
    // walk translated + uniformly scaled instances without parking (<= numTopInner)]; the world-space copies start behind them, the instances' root
    // copies (one slot per instance) come last
    const uint32_t foldedBase = staticNodes + numTopInner;
    out.topSlots = 2u * numTopInner;
    // ---- instances copied to world space --------------------------------------------------------------------
    // An instance costs every ray that enters it a transform in and a restore out on top of the traversal proper.  With 288 GB of
    // HBM the instanced geometry of scenes like the benchmark's (12 x 82 k triangles: ~110 MB of nodes and triangles) simply fits
    // as world-space copies, so instances are copied while a byte budget lasts -- single-leaf meshes (a ground quad, an area light)
    // first, they cost almost nothing -- and the rest stay two-level.  (t,u,v) are the same in both spaces (the reference never
    // renormalises the transformed direction, scene.cl:118-121); the traversal kernels map a hit on a copy back to (original
    // triangle, instance).  The copies themselves are made on the device (pt_bake.h); this only lays them out.
    {
        uint64_t budgetBytes = 2ull << 30;
        if (const char* e = getenv("PTAMD_BAKE_BUDGET_GB")) // diagnostics (bench.py, two_level_general: the copied scene as the yardstick of the entered one)
            budgetBytes = (uint64_t)std::max(0.0, atof(e) * (double)(1ull << 30));
        uint64_t usedBytes = 0;
        uint32_t nextNode = staticNodes + out.topSlots, nextTri = staticTris;
        auto tryBake = [&](uint32_t instIndex, bool wholeTrees) {
            const StaticScene::StaticGeom::Root& root = sg.roots[instRoot[instIndex]];
            const bool single = refCount(root.ref) != 0u; // the mesh is one leaf
            if (single != !wholeTrees)
                return;
            if (!single && (!root.bakeable || root.numNodes == 0u || (c->cfg.flags & PT_FLAG_TWO_LEVEL_ONLY) || parityMode(c))) // parity mode follows the reference to the letter
                return;
            double w[4][8]; // [r][4..7] = row r of the world transform
            if (!invertTransform(topNodes[hInst[instIndex].topNode].invTransform, w))
                return; // singular: stays an instance
            const uint64_t bytes = (uint64_t)root.numNodes * sizeof(WideNode) + (uint64_t)root.numRefs * sizeof(TriIsect);
            if ((!single && usedBytes + bytes > budgetBytes) || (uint64_t)nextNode + root.numNodes >= kRefIndexMask - 4u
                || (uint64_t)nextTri + root.numRefs >= kRefIndexMask - 4u)
                return;
            usedBytes += bytes;
            BakeJob j {};
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 4; k++)
                    j.m[r * 4 + k] = w[r][4 + k];
            j.srcNode = root.nodeBase, j.numNodes = root.numNodes, j.dstNode = nextNode;
            j.srcRef = root.refBase, j.numRefs = root.numRefs, j.dstTri = nextTri;
            j.instance = instIndex;
            out.jobs.push_back(j);
            topRef[hInst[instIndex].topNode] = single ? makeRef(nextTri, refCount(root.ref)) : makeRef(nextNode, 0u);
            nextNode += root.numNodes;
            nextTri += root.numRefs;
        };
        if (!(c->cfg.flags & PT_FLAG_NO_BAKED_INSTANCES)) {
            for (uint32_t k = 0; k < hInst.size(); k++) // single leaves first
                tryBake(k, false);
            // Whole trees: ALL of them or none (round 6).  A scene that is partly copied pays for both: every ray runs the kernels that can enter instances,
            // and the copies' bytes push the shared trees out of the caches (432 instances of the 82 k-triangle meshes, 421 copied + 13 entered: 8 481 Mrays/s
            // against 9 089 with all of them entered and 9 017 with all of them copied: profiles/round6/).
            uint64_t allBytes = 0;
            for (uint32_t k = 0; k < hInst.size(); k++) {
                const StaticScene::StaticGeom::Root& root = sg.roots[instRoot[k]];
                if (refCount(root.ref) == 0u)
                    allBytes += (uint64_t)root.numNodes * sizeof(WideNode) + (uint64_t)root.numRefs * sizeof(TriIsect);
            }
            if (allBytes <= budgetBytes)
                for (uint32_t k = 0; k < hInst.size(); k++)
                    tryBake(k, true);
        }
        out.bakedNodes = nextNode - (staticNodes + out.topSlots);
        out.bakedTris = nextTri - staticTris;
    }
    uint32_t topDepth = 0;
    { // depth / cycle check from the root
        std::vector<std::pair<uint32_t, uint32_t>> st { { topRoot, 1u } };
        size_t visited = 0;
        while (!st.empty()) {
            auto [ni, depth] = st.back();
            st.pop_back();
            if (++visited > nTop)
                return fail(c, PT_ERR_INVALID, "top-level BVH is not a tree");
            topDepth = std::max(topDepth, depth);
            if (!topNodes[ni].isLeaf) {
                st.push_back({ topNodes[ni].a, depth + 1 });
                st.push_back({ topNodes[ni].b, depth + 1 });
            }
        }
    }
    // one pending entry per level of either tree + the leave-instance sentinel
    if (topDepth + 1 + maxBottomDepth > (uint32_t)kTraversalStackMax)
        return fail(c, PT_ERR_UNSUPPORTED, "BVH depth %u (top) + %u (bottom) exceeds the traversal stack (%d)", topDepth, maxBottomDepth, kTraversalStackMax);
    if ((uint64_t)staticNodes + out.topSlots + out.bakedNodes + hInst.size() > kRefIndexMask)
        return fail(c, PT_ERR_UNSUPPORTED, "too many BVH nodes");
    // ---- the top level: pair nodes -> 4-wide, packed breadth-first into the slots behind the static nodes -----------------
    std::vector<PairNode> topPairs(numTopInner);
    auto local = [&](uint32_t ref) { return refIndex(ref) - staticNodes; }; // top-level inner reference -> index into topPairs
    auto isTopInner = [&](uint32_t ref) { return ref != kRefNone && refCount(ref) == 0u && refIndex(ref) >= staticNodes && refIndex(ref) < staticNodes + numTopInner; };
    for (uint32_t i = 0; i < nTop; i++) {
        const pt_top_bvh_node& n = topNodes[i];
        if (n.isLeaf)
            continue;
        const pt_top_bvh_node& L = topNodes[n.a];
        const pt_top_bvh_node& R = topNodes[n.b];
        PairNode pn {};
        pn.bx = make_float4(L.min[0], L.max[0], R.min[0], R.max[0]);
        pn.by = make_float4(L.min[1], L.max[1], R.min[1], R.max[1]);
        pn.bz = make_float4(L.min[2], L.max[2], R.min[2], R.max[2]);
        // inside the collapse the top-level children are indices into topPairs; every other reference is opaque to it (instance
        // references and leaves by their count, the roots of world-space copies by an index beyond the array: they start behind the
        // top level's slots)
        pn.left = isTopInner(topRef[n.a]) ? makeRef(local(topRef[n.a]), 0u) : topRef[n.a];
        pn.right = isTopInner(topRef[n.b]) ? makeRef(local(topRef[n.b]), 0u) : topRef[n.b];
        topPairs[local(topRef[i])] = pn;
    }
    const std::vector<WideKids> kids = collapseKids(topPairs);
    // breadth-first packing of the top-level nodes the collapse kept
    uint32_t rootRef = topRef[topRoot];
    constexpr uint32_t kUnset = 0xFFFFFFFFu;
    std::vector<uint32_t> newIndex(numTopInner, kUnset), order;
    auto isKept = [&](uint32_t r) { return r != kRefNone && refCount(r) == 0u && refIndex(r) < numTopInner; };
    if (isTopInner(rootRef)) {
        newIndex[local(rootRef)] = 0;
        order.push_back(local(rootRef));
        for (size_t q = 0; q < order.size(); q++)
            for (int k = 0; k < 4; k++) {
                const uint32_t r = kids[order[q]].ref[k];
                if (!kids[order[q]].empty[k] && isKept(r) && newIndex[refIndex(r)] == kUnset) {
                    newIndex[refIndex(r)] = (uint32_t)order.size();
                    order.push_back(refIndex(r));
                }
            }
        rootRef = makeRef(staticNodes, 0u);
    }
    out.topWide.resize(order.size());
    out.hasInstances = refCount(rootRef) == kRefSpecial;
    for (size_t q = 0; q < order.size(); q++) {
        const WideKids& wk = kids[order[q]];
        uint32_t refs[4];
        for (int k = 0; k < 4; k++) {
            refs[k] = wk.empty[k] ? sg.emptyRef : (isKept(wk.ref[k]) ? makeRef(staticNodes + newIndex[refIndex(wk.ref[k])], 0u) : wk.ref[k]);
            if (!wk.empty[k] && refCount(refs[k]) == kRefSpecial)
                out.hasInstances = true;
        }
        quantiseWideNode(wk.lo, wk.hi, refs, wk.empty, sg.emptyRef, &out.topWide[q]);
    }
    // ---- worst-case traversal stack: the top level on top of the deepest thing below it (an entered instance adds its sentinel)
    std::vector<uint32_t> topNeed(order.size(), 0u);
    // the copies' roots are looked up by node index: a map for scenes with many of them
    std::vector<std::pair<uint32_t, uint32_t>> copyRoots;
    for (const BakeJob& j : out.jobs)
        if (j.numNodes)
            copyRoots.push_back({ j.dstNode, sg.stackNeed[j.srcNode] });
    std::sort(copyRoots.begin(), copyRoots.end());
    auto needOf = [&](uint32_t ref) -> uint32_t {
        if (refCount(ref) == 0u && refIndex(ref) >= staticNodes && refIndex(ref) < staticNodes + order.size())
            return topNeed[refIndex(ref) - staticNodes];
        if (refCount(ref) == 0u) {
            auto it = std::lower_bound(copyRoots.begin(), copyRoots.end(), std::make_pair(refIndex(ref), 0u));
            return it != copyRoots.end() && it->first == refIndex(ref) ? it->second : 0u;
        }
        if (refCount(ref) == kRefSpecial) { // an entered instance: its sentinel + its mesh tree
            const uint32_t rr = hInst[refIndex(ref)].rootRef;
            return 1u + (refCount(rr) == 0u ? sg.stackNeed[refIndex(rr)] : 0u);
        }
        return 0u;
    };
    for (size_t q = order.size(); q-- > 0;) {
        uint32_t n = 0, deepest = 0;
        for (uint32_t r : out.topWide[q].child)
            if (r != sg.emptyRef)
                n++, deepest = std::max(deepest, needOf(r));
        topNeed[q] = (n > 0 ? n - 1 : 0u) + deepest;
    }
    const uint32_t stackNeed = needOf(rootRef);
    if (stackNeed > (uint32_t)kTraversalStackMax)
        return fail(c, PT_ERR_UNSUPPORTED, "BVH needs %u traversal stack entries, %d are available", stackNeed, kTraversalStackMax);
    // k_trace_packet keeps its stack in the 64 lanes of a register (instance references are entered there too, pt_packet.h)
    out.packetOk = stackNeed <= kPacketStack;
    out.stackNeed = stackNeed;
    // ---- the top level once more, for the per-ray kernels: instances whose transform is a translation + uniform scale (the reference's own scenes,
    // BASELINE configs 4 / 5) are walked WITHOUT parking (pt_trace.h).  In this copy of the top level such an instance is an ordinary inner reference
    // -- to the instance's own copy of its mesh's ROOT node (object space, 64 bytes; the copies are the LAST run of the node array, copy k = instance
    // k) -- and (1 / s, w = -t / s) of its inverse transform sits in a table the kernel keeps in LDS.  Same pairs, same boxes, hence the same
    // collapse and a worst-case stack no larger than the one computed above (no sentinel).
    out.rootRefFolded = rootRef;
    out.foldedInstances = 0;
    out.instRoots.clear();
    out.instRootSrc.clear();
    out.instFold.clear();
    out.instRootBase = staticNodes + out.topSlots + out.bakedNodes;
    {
        static const bool envNoFold = getenv("PTAMD_NO_FOLDED_INSTANCES") != nullptr; // diagnostics: every entered instance takes the parked route (rounds 2-4)
        const bool parked = envNoFold || (c->cfg.flags & PT_FLAG_PARKED_INSTANCES) != 0u || parityMode(c); // (parity mode follows the reference to the letter)
        auto simple = [](const Instance& in) { return in.simple != 0u; };
        // Which route for the instances that are entered?  Every one a translation + uniform scale and few enough for the LDS table: folded (no entry step at
        // all).  Otherwise -- a rotation, a non-uniform scale, a shear, or instance number 96 -- the general route (round 6): every instance is entered as a
        // leaf-kind step, nothing is parked (pt_trace.h, LEVELS 2).  PTAMD_GENERAL_ROUTE=1 / 0 (diagnostics): the general route for every scene with entered
        // instances / never (rounds 2-5: such scenes park).
        uint32_t entered = 0, enteredGeneral = 0;
        for (size_t k = 0; k < hInst.size(); k++)
            if (refCount(topRef[hInst[k].topNode]) == kRefSpecial)
                entered++, enteredGeneral += simple(hInst[k]) ? 0u : 1u;
        out.enteredInstances = entered;
        // A scene with FEW such instances among many translated + uniformly scaled ones (at most a quarter) that fits the table keeps the folded route
        // for those -- no entry step at all -- and parks the few.
        static const char* envGeneral = getenv("PTAMD_GENERAL_ROUTE");
        const bool tableHolds = hInst.size() + 1 <= kInstFoldTable;
        const bool mostlySimple = enteredGeneral * 4u <= entered;
        out.generalRoute = !parked && entered > 0u && (envGeneral ? atoi(envGeneral) != 0 : (!mostlySimple || !tableHolds));
        const bool noFold = parked || out.generalRoute || !tableHolds;
        std::vector<uint8_t> folded(hInst.size(), 0);
        for (size_t k = 0; k < hInst.size() && !noFold; k++)
            if (refCount(topRef[hInst[k].topNode]) == kRefSpecial && simple(hInst[k]))
                folded[k] = 1, out.foldedInstances++;
        if (out.foldedInstances) {
            const uint32_t instRootBase = out.instRootBase;
            auto foldRef = [&](uint32_t r) { return refCount(r) == kRefSpecial && refIndex(r) < hInst.size() && folded[refIndex(r)] ? makeRef(instRootBase + refIndex(r), 0u) : r; };
            std::vector<PairNode> pairsB = topPairs;
            for (PairNode& pn : pairsB)
                pn.left = foldRef(pn.left), pn.right = foldRef(pn.right);
            const std::vector<WideKids> kidsB = collapseKids(pairsB);
            std::vector<uint32_t> newB(numTopInner, kUnset), orderB;
            uint32_t rootB = foldRef(topRef[topRoot]);
            if (isTopInner(topRef[topRoot])) {
                newB[local(topRef[topRoot])] = 0;
                orderB.push_back(local(topRef[topRoot]));
                for (size_t q = 0; q < orderB.size(); q++)
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r = kidsB[orderB[q]].ref[k];
                        if (!kidsB[orderB[q]].empty[k] && isKept(r) && newB[refIndex(r)] == kUnset) {
                            newB[refIndex(r)] = (uint32_t)orderB.size();
                            orderB.push_back(refIndex(r));
                        }
                    }
                rootB = makeRef(foldedBase, 0u);
            }
            out.topWide.resize((size_t)numTopInner + orderB.size()); // (the gap behind the first top level stays zero: never referenced)
            for (size_t q = 0; q < orderB.size(); q++) {
                const WideKids& wk = kidsB[orderB[q]];
                uint32_t refs[4];
                for (int k = 0; k < 4; k++)
                    refs[k] = wk.empty[k] ? sg.emptyRef : (isKept(wk.ref[k]) ? makeRef(foldedBase + newB[refIndex(wk.ref[k])], 0u) : wk.ref[k]);
                quantiseWideNode(wk.lo, wk.hi, refs, wk.empty, sg.emptyRef, &out.topWide[(size_t)numTopInner + q]);
            }
            out.rootRefFolded = rootB;
            // the instances' root copies and the table of their transforms (entry 0: the identity; instances on the general route: the identity too --
            // their lanes hold the instance-space ray in registers)
            out.instRoots.assign(hInst.size(), WideNode {});
            out.instRootSrc.assign(hInst.size(), 0xFFFFFFFFu);
            out.instFold.assign(hInst.size() + 1, make_float4(1.f, 0.f, 0.f, 0.f));
            for (size_t k = 0; k < hInst.size(); k++) {
                if (!folded[k])
                    continue;
                Instance& in = hInst[k];
                out.instFold[k + 1] = make_float4(in.r0.x, in.r0.w, in.r1.w, in.r2.w);
                in.folded = 1u;
                if (refCount(in.rootRef) == 0u) {
                    out.instRootSrc[k] = refIndex(in.rootRef); // the mesh's packed root node as the device holds it (the host's mirror goes stale with a refit): object space, children in the shared tree
                } else { // the mesh is a single leaf: a one-child node around it -- the top-level leaf's box taken into object space, a few ulps outwards
                    const pt_top_bvh_node& leaf = topNodes[in.topNode];
                    float lo[4][3], hi[4][3];
                    const uint32_t refs[4] = { in.rootRef, sg.emptyRef, sg.emptyRef, sg.emptyRef };
                    const bool empty[4] = { false, true, true, true };
                    const float w[3] = { in.r0.w, in.r1.w, in.r2.w };
                    for (int a = 0; a < 3; a++) {
                        const double l = (double)leaf.min[a] * in.r0.x + w[a], h = (double)leaf.max[a] * in.r0.x + w[a];
                        lo[0][a] = nextafterf(nextafterf((float)l, -INFINITY), -INFINITY), hi[0][a] = nextafterf(nextafterf((float)h, INFINITY), INFINITY);
                        for (int q = 1; q < 4; q++)
                            lo[q][a] = 1.f, hi[q][a] = -1.f;
                    }
                    quantiseWideNode(lo, hi, refs, empty, sg.emptyRef, &out.instRoots[k]);
                }
            }
        }
    }
    std::vector<Light>& hLights = out.lights;
    hLights.resize(nL);
    for (uint32_t i = 0; i < nL; i++) {
        const pt_emissive_triangle& e = lights[i];
        const V3 v0 = mk(e.vertices[0][0], e.vertices[0][1], e.vertices[0][2]);
        const V3 v1 = mk(e.vertices[1][0], e.vertices[1][1], e.vertices[1][2]);
        const V3 v2 = mk(e.vertices[2][0], e.vertices[2][1], e.vertices[2][2]);
        // Heron's formula (shading_helper.cl:204-214)
        const V3 A = v1 - v0, B = v2 - v1, C = v0 - v2;
        const float la = sqrtf(dot(A, A)), lb = sqrtf(dot(B, B)), lc = sqrtf(dot(C, C));
        const float s = (la + lb + lc) / 2.0f;
        const float area = sqrtf(s * (s - la) * (s - lb) * (s - lc));
        const V3 nrm = normalize(cross(v1 - v0, v2 - v0));
        hLights[i].v0 = make_float4(v0.x, v0.y, v0.z, area);
        hLights[i].v1 = make_float4(v1.x, v1.y, v1.z, 0.f);
        hLights[i].v2 = make_float4(v2.x, v2.y, v2.z, 0.f);
        hLights[i].normal = make_float4(nrm.x, nrm.y, nrm.z, 0.f);
        hLights[i].colour = make_float4(e.material.u.emissive.emissiveColour[0], e.material.u.emissive.emissiveColour[1], e.material.u.emissive.emissiveColour[2], 0.f);
    }
    out.numLights = nL;
    out.rootRef = rootRef;
    return PT_OK;
}

template <typename T>
int growTo(pt_ctx* c, DevBuf<T>& buf, size_t count)
{
    if (buf.n >= std::max<size_t>(count, 1))
        return PT_OK;
    HIPCHK(c, buf.alloc(std::max<size_t>(count + count / 8, 1))); // some headroom: the number of world-space copies varies from state to state
    return PT_OK;
}

} // namespace

extern "C" {

// Convert the next dynamic state on the host -- the top level, the instance table, the lights: the trees below are static -- and
// start putting it into the INACTIVE set on the copy stream (a few KB of copies and, where instances are copied to world space, two
// kernels); returns without waiting for the device.  Renders already enqueued (and any enqueued before the next pt_frame_tick)
// keep using the active set.
int pt_upload_dynamic_async(pt_ctx* c, const pt_emissive_triangle* lights, uint32_t nL, const pt_top_bvh_node* topNodes, uint32_t nTop, uint32_t topRoot)
{
    return guarded(c, "pt_upload_dynamic_async", [&]() -> int {
    if (!c)
        return PT_ERR_INVALID;
    if (!c->haveStatic)
        return fail(c, PT_ERR_STATE, "pt_upload_dynamic: call pt_upload_static first");
    if (!topNodes || nTop == 0 || topRoot >= nTop)
        return fail(c, PT_ERR_INVALID, "pt_upload_dynamic: bad top-level BVH");
    if (nL > 0 && !lights)
        return fail(c, PT_ERR_INVALID, "pt_upload_dynamic: null light array");
    HIPCHK(c, hipSetDevice(c->device));
    // a rebuilt scene is waiting (pt_upload_static_async): this state is built on IT -- its top-level leaves name the new sub-BVH roots
    struct UseScene {
        pt_ctx* c;
        StaticScene* saved;
        ~UseScene() { c->st = saved; }
    } useScene { c, c->st };
    if (c->statPending >= 0)
        c->st = &c->stat[c->statPending];
    StageTimer tm("uploadDynamicAsync");
    DynamicHost h;
    int rc = convertDynamic(c, lights, nL, topNodes, nTop, topRoot, h);
    tm.lap("convertDynamic");
    if (rc || (rc = uploadStaticGeom(c)))
        return rc;
    tm.lap("uploadStaticGeom");
    StaticScene::StaticGeom& sg = c->st->sg;
    const int target = c->haveDynamic ? 1 - c->active : c->active; // nothing active yet: fill the active set itself
    pt_ctx::DynamicSet& d = c->dyn[target];
    // the copies may not start before the renders that still read this set are done (it was the active set until the last tick),
    // nor before an earlier, never-adopted upload into it has finished (same stream: ordered)
    if (d.used)
        HIPCHK(c, hipStreamWaitEvent(c->copyStream, d.lastUse, 0));
    const size_t staticNodes = sg.wide.size(), staticTris = (size_t)c->st->numTris + 1;
    // (+ 1: the SMALL traversal instantiations fetch 96 bytes from wherever a lane stands -- 32 beyond a node, 48 beyond a one-triangle leaf)
    const size_t needWide = staticNodes + h.topSlots + h.bakedNodes + h.instRoots.size() + 1, needTris = staticTris + h.bakedTris + 1;
    // the traversal kernels address nodes and triangle records by base + 32-bit byte offset (pt_trace.h, PT_OFFSET32)
    uint64_t offsetLimit = 0xFFFFFFFFull;
    if (const char* e = getenv("PTAMD_OFFSET_LIMIT")) // (tests: the refusal below without a 4 GB scene)
        offsetLimit = std::min<uint64_t>(offsetLimit, strtoull(e, nullptr, 10));
    if (needWide * sizeof(WideNode) > offsetLimit || needTris * sizeof(TriIsect) > offsetLimit)
        return fail(c, PT_ERR_UNSUPPORTED, "scene of %zu packed nodes and %zu triangle records (world-space copies included): more than 4 GB of either; enter the instances instead (PT_FLAG_NO_BAKED_INSTANCES)",
            needWide, needTris);
    const size_t bytes[7] = { h.topWide.size() * sizeof(WideNode), h.instances.size() * sizeof(Instance), h.lights.size() * sizeof(Light),
        h.jobs.size() * sizeof(BakeJob), h.instRoots.size() * sizeof(WideNode), h.instFold.size() * sizeof(float4), h.instRootSrc.size() * sizeof(uint32_t) };
    const size_t total = bytes[0] + bytes[1] + bytes[2] + bytes[3] + bytes[4] + bytes[5] + bytes[6];
    if (d.wide.n < needWide || d.tris.n < needTris || d.fat.n < sg.fat.size() || d.instances.n < std::max<size_t>(h.instances.size(), 1)
        || d.lights.n < std::max<size_t>(h.lights.size(), 1) || d.jobs.n < std::max<size_t>(h.jobs.size(), 1) || d.instFold.n < std::max<size_t>(h.instFold.size(), 1)
        || d.instRootSrc.n < std::max<size_t>(h.instRootSrc.size(), 1) || d.stageBytes < std::max<size_t>(total, 1)) {
        // growing frees device memory, which the runtime only does once nothing uses it: wait for both streams (rare: the first
        // uploads, or a state with more world-space copies than any before)
        HIPCHK(c, hipStreamSynchronize(c->copyStream));
        if (d.used)
            HIPCHK(c, hipEventSynchronize(d.lastUse));
        const bool regrown = d.wide.n < needWide || d.tris.n < needTris || d.fat.n < sg.fat.size();
        if ((rc = growTo(c, d.wide, needWide)) || (rc = growTo(c, d.tris, needTris)) || (rc = growTo(c, d.fat, sg.fat.size()))
            || (rc = growTo(c, d.instances, h.instances.size())) || (rc = growTo(c, d.lights, h.lights.size())) || (rc = growTo(c, d.jobs, h.jobs.size()))
            || (rc = growTo(c, d.instFold, h.instFold.size())) || (rc = growTo(c, d.instRootSrc, h.instRootSrc.size())))
            return rc;
        if (regrown)
            d.staticVersion = 0; // fresh buffers: the static arrays have to be put in again
        if (d.stageBytes < std::max<size_t>(total, 1)) {
            if (d.stage)
                (void)hipHostFree(d.stage);
            d.stage = nullptr;
            d.stageBytes = total + total / 8 + 4096;
            HIPCHK(c, hipHostMalloc(&d.stage, d.stageBytes, hipHostMallocDefault));
            d.stageBusy = false;
        }
    }
    if (d.stageBusy) { // the staging memory of this set is about to be rewritten: its last copies (two ticks ago) must have been read
        HIPCHK(c, hipEventSynchronize(d.stageRead));
        d.stageBusy = false;
    }
    tm.lap("grow");
    if (d.staticVersion != sg.version) { // device to device, from the master copy
        if (staticNodes)
            HIPCHK(c, hipMemcpyAsync(d.wide.p, sg.dWide.p, staticNodes * sizeof(WideNode), hipMemcpyDeviceToDevice, c->copyStream));
        HIPCHK(c, hipMemcpyAsync(d.tris.p, sg.dTris.p, staticTris * sizeof(TriIsect), hipMemcpyDeviceToDevice, c->copyStream));
        if (!sg.fat.empty())
            HIPCHK(c, hipMemcpyAsync(d.fat.p, sg.dFat.p, sg.fat.size() * sizeof(TriFat), hipMemcpyDeviceToDevice, c->copyStream));
        d.staticVersion = sg.version;
    }
    unsigned char* st = (unsigned char*)d.stage;
    const void* src[7] = { h.topWide.data(), h.instances.data(), h.lights.data(), h.jobs.data(), h.instRoots.data(), h.instFold.data(), h.instRootSrc.data() };
    void* dst[7] = { d.wide.p + staticNodes, d.instances.p, d.lights.p, d.jobs.p, d.wide.p + h.instRootBase, d.instFold.p, d.instRootSrc.p };
    for (int k = 0; k < 7; k++) {
        if (bytes[k] == 0)
            continue;
        std::memcpy(st, src[k], bytes[k]);
        HIPCHK(c, hipMemcpyAsync(dst[k], st, bytes[k], hipMemcpyHostToDevice, c->copyStream));
        st += bytes[k];
    }
    if (total) {
        HIPCHK(c, hipEventRecord(d.stageRead, c->copyStream));
        d.stageBusy = true;
    }
    if (!h.instRootSrc.empty()) { // the folded instances' copies of their meshes' root nodes, from the set's own (just refreshed) copy of the static nodes
        const uint32_t n = (uint32_t)h.instRootSrc.size();
        hipLaunchKernelGGL(k_inst_roots, dim3((n + 63u) / 64u), dim3(64), 0, c->copyStream, d.wide.p, d.instRootSrc.p, d.wide.p + h.instRootBase, n);
    }
    if (!h.jobs.empty()) { // the world-space copies: one thread per (copy, node) and per (copy, triangle reference)
        BakeArgs ba {};
        ba.srcWide = sg.dWide.p, ba.srcBoxes = sg.dBoxes.p, ba.srcLeafOfs = sg.dLeafOfs.p, ba.refTri = sg.dRefTri.p, ba.srcTris = sg.dTris.p;
        ba.dstWide = d.wide.p, ba.dstTris = d.tris.p, ba.emptyRef = sg.emptyRef;
        for (size_t first = 0; first < h.jobs.size(); first += 32768) {
            const uint32_t n = (uint32_t)std::min<size_t>(32768, h.jobs.size() - first);
            uint32_t maxNodes = 0, maxRefs = 0;
            for (uint32_t k = 0; k < n; k++)
                maxNodes = std::max(maxNodes, h.jobs[first + k].numNodes), maxRefs = std::max(maxRefs, h.jobs[first + k].numRefs);
            ba.jobs = d.jobs.p + first;
            if (maxNodes)
                hipLaunchKernelGGL(k_bake_nodes, dim3((maxNodes + 127) / 128, n), dim3(128), 0, c->copyStream, ba);
            if (maxRefs)
                hipLaunchKernelGGL(k_bake_tris, dim3((maxRefs + 255) / 256, n), dim3(256), 0, c->copyStream, ba);
        }
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(d.uploaded, c->copyStream));
    tm.lap("enqueue");
    d.numLights = h.numLights;
    d.rootRef = h.rootRef;
    d.staticIndex = (int)(c->st - c->stat);
    d.numTris = c->st->numTris, d.firstWorldNode = (uint32_t)sg.wide.size();
    d.rootRefFolded = h.rootRefFolded;
    d.foldedInstances = h.foldedInstances;
    d.instRootBase = h.instRootBase, d.numInstRoots = (uint32_t)h.instRoots.size(), d.instFoldCount = (uint32_t)h.instFold.size();
    d.packetOk = h.packetOk;
    d.stackNeed = h.stackNeed;
    d.hasInstances = h.hasInstances;
    d.generalRoute = h.generalRoute && h.hasInstances, d.enteredInstances = h.enteredInstances;
    d.instanceTopNode = std::move(h.instanceTopNode);
    c->pending = target;
    return PT_OK;
    });
}

// RayTracer::frameTick's flip (src/raytracer.cpp:183-189) with the barrier of transferDynamicData (:593): renders enqueued from
// now on wait for the pending upload and read the new set; nothing waits on the host.
int pt_frame_tick(pt_ctx* c)
{
    if (!c)
        return PT_ERR_INVALID;
    if (c->pending < 0)
        return PT_OK; // nothing was uploaded since the last tick
    HIPCHK(c, hipSetDevice(c->device));
    pt_ctx::DynamicSet& next = c->dyn[c->pending];
    HIPCHK(c, hipStreamWaitEvent(c->stream, next.uploaded, 0));
    if (c->pending != c->active) {
        pt_ctx::DynamicSet& old = c->dyn[c->active];
        HIPCHK(c, hipEventRecord(old.lastUse, c->stream)); // everything enqueued so far may still read the old set
        old.used = true;
    }
    next.used = true;
    HIPCHK(c, hipEventRecord(next.lastUse, c->stream));
    c->active = c->pending;
    newEpoch(c); // (what a batch's first pass emits belongs to the scene state it was measured on)
    c->pending = -1;
    c->haveDynamic = true;
    if (next.staticIndex != c->statCur) { // the state was built on a rebuilt scene (pt_upload_static_async): that scene is the current one from here on
        c->statCur = next.staticIndex;
        c->st = &c->stat[c->statCur];
    }
    if (c->statPending == c->statCur)
        c->statPending = -1;
    refreshSceneView(c);
    return PT_OK;
}

// the synchronous-looking form: the next pt_render sees the new state (still no host-side wait for the device)
int pt_upload_dynamic(pt_ctx* c, const pt_emissive_triangle* lights, uint32_t nL, const pt_top_bvh_node* topNodes, uint32_t nTop, uint32_t topRoot)
{
    const int rc = pt_upload_dynamic_async(c, lights, nL, topNodes, nTop, topRoot);
    return rc ? rc : pt_frame_tick(c);
}

int pt_upload_texture_array(pt_ctx* c, int kind, uint32_t width, uint32_t height, uint32_t layers, int format, const void* data)
{
    return guarded(c, "pt_upload_texture_array", [&]() -> int {
    if (!c)
        return PT_ERR_INVALID;
    if ((kind != 0 && kind != 1) || !data || width == 0 || height == 0 || layers == 0 || (format != PT_TEX_RGBA32F && format != PT_TEX_BGRA8_UNORM))
        return fail(c, PT_ERR_INVALID, "pt_upload_texture_array: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DevBuf<uint8_t>& buf = kind == 0 ? c->texMaterial : c->texSky;
    const size_t bytes = (size_t)width * height * layers * (format == PT_TEX_RGBA32F ? sizeof(float4) : 4u);
    HIPCHK(c, buf.alloc(bytes));
    HIPCHK(c, hipMemcpy(buf.p, data, bytes, hipMemcpyHostToDevice));
    Texture& t = kind == 0 ? c->scene.materialTex : c->scene.sky;
    t.width = (int)width;
    t.height = (int)height;
    t.layers = (int)layers;
    t.format = format;
    refreshSceneView(c);
    return PT_OK;
    });
}

int pt_set_camera(pt_ctx* c, const pt_camera* cam)
{
    if (!c || !cam)
        return PT_ERR_INVALID;
    auto f4 = [](const float* p) { return make_float4(p[0], p[1], p[2], 0.f); };
    c->camera.eye = f4(cam->eyePoint);
    c->camera.screen = f4(cam->screenPoint);
    c->camera.u = f4(cam->u);
    c->camera.v = f4(cam->v);
    c->camera.uN = f4(cam->uNormalized);
    c->camera.vN = f4(cam->vNormalized);
    c->camera.focalDistance = cam->focalDistance;
    c->camera.apertureRadius = cam->apertureRadius;
    c->camera.relativeAperture = cam->relativeAperture;
    c->camera.shutterTime = cam->shutterTime;
    c->camera.ISO = cam->ISO;
    c->camera.thinLens = cam->thinLensEnabled ? 1u : 0u;
    c->haveCamera = true;
    newEpoch(c);
    return PT_OK;
}

int pt_set_tiles(pt_ctx* c, const pt_rect* rects, uint32_t n)
{
    return guarded(c, "pt_set_tiles", [&]() -> int {
    if (!c)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t W = c->cfg.width, H = c->cfg.height;
    pt_rect whole { 0, 0, W, H };
    if (n == 0 || !rects) {
        rects = &whole;
        n = 1;
    }
    const bool rowMajor = (c->cfg.flags & PT_FLAG_ROWMAJOR_PIXELS) || parityMode(c);
    std::vector<uint32_t> list;
    // position of every pixel in the list; doubles as the overlap check: the kernels deposit with a plain
    // read-modify-write that relies on ONE live path per (plane, pixel), so a pixel listed twice would race
    constexpr uint32_t kUnowned = 0xFFFFFFFFu;
    std::vector<uint32_t> ordinal((size_t)W * H, kUnowned);
    auto add = [&](uint32_t x, uint32_t y) {
        const uint32_t px = y * W + x;
        if (ordinal[px] != kUnowned)
            return false;
        ordinal[px] = (uint32_t)list.size();
        list.push_back(px);
        return true;
    };
    for (uint32_t r = 0; r < n; r++) {
        const pt_rect& q = rects[r];
        if (q.x0 >= q.x1 || q.y0 >= q.y1 || q.x1 > W || q.y1 > H)
            return fail(c, PT_ERR_INVALID, "pt_set_tiles: rect %u out of bounds", r);
        bool ok = true;
        if (rowMajor) {
            for (uint32_t y = q.y0; y < q.y1; y++)
                for (uint32_t x = q.x0; x < q.x1; x++)
                    ok = add(x, y) && ok;
        } else { // 8x8 pixel blocks: a 64-lane wave starts on a compact screen-space tile
            for (uint32_t by = q.y0; by < q.y1; by += 8)
                for (uint32_t bx = q.x0; bx < q.x1; bx += 8)
                    for (uint32_t y = by; y < std::min(by + 8, q.y1); y++)
                        for (uint32_t x = bx; x < std::min(bx + 8, q.x1); x++)
                            ok = add(x, y) && ok;
        }
        if (!ok)
            return fail(c, PT_ERR_INVALID, "pt_set_tiles: rect %u overlaps an earlier one", r);
    }
    if (c->queuesReady && list == c->hostPixelList)
        return PT_OK; // the same pixels in the same order: queues, planes and ordinals stay as they are
    c->identityPixels = rowMajor && n == 1 && rects[0].x0 == 0 && rects[0].y0 == 0 && rects[0].x1 == W && rects[0].y1 == H;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = uploadVec(c, c->pixelList, list);
    if (rc)
        return rc;
    if (list.size() == (size_t)W * H) { // whole frame: the extra accumulator planes are indexed by the pixel itself
        c->pixelOrdinal.release();
    } else if ((rc = uploadVec(c, c->pixelOrdinal, ordinal)))
        return rc;
    c->numOwned = (uint32_t)list.size();
    c->hostPixelList = std::move(list);
    c->queuesReady = false; // the ordinal of a pixel may have changed: queues and planes are re-made lazily (ensureQueues)
    return PT_OK;
    });
}

int pt_set_accum_buffer(pt_ctx* c, void* device_float4)
{
    if (!c)
        return PT_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->accum = device_float4 ? (float4*)device_float4 : c->accumOwn.p;
    return PT_OK;
}

int pt_clear(pt_ctx* c)
{
    return guarded(c, "pt_clear", [&]() -> int {
    if (!c)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->accum, 0, (size_t)c->cfg.width * c->cfg.height * sizeof(float4), c->stream));
    if (c->foldPlanes > 1 && c->accumPlanes.p) // samples of a pt_render that failed half-way: not to be folded into the fresh accumulator
        HIPCHK(c, hipMemsetAsync(c->accumPlanes.p, 0, c->accumPlanes.n * sizeof(float4), c->stream));
    c->foldPlanes = 0;
    c->spp = 0;
    if (c->overflowPinned && *c->overflowPinned) { // a reported overflow is cleared with the image it spoiled (whatever set it has run: the word is only read after a synchronisation)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *c->overflowPinned = 0u;
    }
    if (c->accumShadow.p) // (zero between pt_render calls unless one failed half-way)
        HIPCHK(c, hipMemsetAsync(c->accumShadow.p, 0, c->accumShadow.n * sizeof(float4), c->stream));
    c->mergePending = false;
    if (parityMode(c) && c->queuesReady) {
        int rc = resetStreams(c);
        if (rc)
            return rc;
    }
    return PT_OK;
    });
}

int pt_render(pt_ctx* c, uint32_t spp)
{
    return guarded(c, "pt_render", [&]() -> int {
    if (!c)
        return PT_ERR_INVALID;
    if (!c->haveStatic || !c->haveDynamic || !c->haveCamera)
        return fail(c, PT_ERR_STATE, "pt_render: scene (static + dynamic) and camera must be set first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = checkOverflow(c)))
        return rc;
    if (c->queuesReady && c->q0Small && !derivedPrimariesCapable(c))
        c->queuesReady = false; // (the first queue was sized for a pinhole's bundles: this camera / scene state queues whole camera rays)
    if ((rc = ensureQueues(c)) || (rc = ensureSpill(c)))
        return rc;
    Prof prof { c };
    HIPCHK(c, hipEventRecord(c->evStart, c->stream));
    const bool fixedSchedule = !parityMode(c) && c->cfg.max_active_rays == 0;
    for (uint32_t s = 0; s < spp;) {
        uint32_t batch = fixedSchedule ? std::min(c->planes, spp - s) : 1u;
        bool probe = false;
        if (fixedSchedule && smallQueues(c)) {
            // Queues smaller than the batch: how much of a batch goes on after the first hit, and how many shadow rays leave from there, is measured -- a report
            // of an earlier batch of this epoch, or a short PROBE batch whose report is waited for (16 samples per pixel, fewer where even that might not
            // fit: its samples count like any others) -- and the batch is cut so that it fits with 3 % to spare.
            if (c->passCountsPending && hipEventQuery(c->passCountsCopied) == hipSuccess)
                adoptPassCounts(c);
            else
                (void)hipGetLastError();
            if (!c->ratiosKnown) {
                probe = true;
                const uint64_t room = std::min(c->capExt, c->capShadow);
                batch = std::min<uint32_t>(batch, (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, room / std::max(c->numOwned, 1u))));
                if (c->passCountsPending) { // an older report is still on its way: take it first, the probe needs the slot
                    HIPCHK(c, hipEventSynchronize(c->passCountsCopied));
                    adoptPassCounts(c);
                    probe = !c->ratiosKnown;
                }
            }
            if (!probe)
                batch = std::min(batch, safeBatch(c));
        }
        if (fixedSchedule) {
            // Up to kGenInterleave samples of a pixel are queue neighbours only if the batch is a multiple of that power of two (k_gen,
            // primaryEntry): a batch of 2 046 would keep TWO together and lose the bundle kernel and every coherent launch behind it.
            // Cut the batch at the largest such multiple; what is left follows as smaller batches.
            uint32_t g = kGenInterleave;
            while (g > 1u && batch < g)
                g >>= 1;
            batch -= batch % g;
        }
        rc = fixedSchedule ? renderSampleFixed(c, c->spp, batch, prof) : renderSampleRefill(c, c->spp);
        if (rc) {
            foldPlanesNow(c); // what the earlier batches of this call deposited belongs to the samples already counted
            return rc;
        }
        c->spp += batch;
        s += batch;
        if (probe) { // wait for the probe's counters (a few ms of rendering) and learn the ratios
            c->probeBatches++;
            if (c->passCountsPending) {
                HIPCHK(c, hipEventSynchronize(c->passCountsCopied));
                adoptPassCounts(c);
            }
            if (!c->ratiosKnown) // (no report slot: cannot happen -- the probe took care of it above)
                return fail(c, PT_ERR_STATE, "the probe batch did not report its counters");
        }
    }
    foldPlanesNow(c); // the accumulator is complete when pt_render's work on the stream is: callers read it with their own tools
    HIPCHK(c, hipEventRecord(c->evStop, c->stream));
    if (c->profile) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double fam[5] = { 0, 0, 0, 0, 0 };
        for (auto& m : prof.marks) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->profEvents[m.second], c->profEvents[m.second + 1]);
            fam[m.first] += ms;
        }
        c->msGen = fam[0], c->msIntersect = fam[1] + fam[4], c->msShade = fam[2], c->msShadow = fam[3], c->msPacket = fam[4];
    }
    return PT_OK;
    });
}

int pt_synchronize(pt_ctx* c)
{
    if (!c)
        return PT_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return checkOverflow(c);
}

int pt_resolve(pt_ctx* c, float* rgba_out)
{
    return guarded(c, "pt_resolve", [&]() -> int {
    if (!c || !rgba_out)
        return PT_ERR_INVALID;
    if (!c->haveCamera || c->spp == 0)
        return fail(c, PT_ERR_STATE, "pt_resolve: nothing rendered yet");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = c->cfg.width * c->cfg.height;
    if (c->resolveTmp.n != n) // once per context: this is the interactive getOutput() path
        HIPCHK(c, c->resolveTmp.alloc(n));
    hipLaunchKernelGGL(k_resolve, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->accum, c->resolveTmp.p, n, (float)c->spp,
        c->camera.relativeAperture, c->camera.shutterTime, c->camera.ISO);
    HIPCHK(c, hipMemcpyAsync(rgba_out, c->resolveTmp.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return checkOverflow(c);
    });
}

int pt_resolve_device(pt_ctx* c, void* device_rgba)
{
    return guarded(c, "pt_resolve_device", [&]() -> int {
    if (!c)
        return PT_ERR_INVALID;
    if (!c->haveCamera || c->spp == 0)
        return fail(c, PT_ERR_STATE, "pt_resolve_device: nothing rendered yet");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = c->cfg.width * c->cfg.height;
    if (!device_rgba) { // the context's own image (pt_resolve_device_ptr)
        if (c->resolveTmp.n != n)
            HIPCHK(c, c->resolveTmp.alloc(n));
        device_rgba = c->resolveTmp.p;
    }
    hipLaunchKernelGGL(k_resolve, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->accum, (float4*)device_rgba, n, (float)c->spp,
        c->camera.relativeAperture, c->camera.shutterTime, c->camera.ISO);
    HIPCHK(c, hipGetLastError());
    return PT_OK;
    });
}

// the image pt_resolve_device(ctx, NULL) writes: width * height float4, valid once the render stream has passed that call; NULL before
// the first one
void* pt_resolve_device_ptr(pt_ctx* c) { return c ? (void*)c->resolveTmp.p : nullptr; }

int pt_read_accum(pt_ctx* c, float* out)
{
    if (!c || !out)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->accum, (size_t)c->cfg.width * c->cfg.height * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return checkOverflow(c);
}

int pt_write_accum(pt_ctx* c, const float* in, uint32_t spp)
{
    if (!c || !in)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->accum, in, (size_t)c->cfg.width * c->cfg.height * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->spp = spp;
    return PT_OK;
}

void* pt_accum_device_ptr(pt_ctx* c) { return c ? (void*)c->accum : nullptr; }

uint32_t pt_samples_per_pixel(const pt_ctx* c) { return c ? c->spp : 0; }

int pt_stats_get(pt_ctx* c, pt_stats* out)
{
    if (!c || !out)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    Totals t {};
    HIPCHK(c, hipMemcpy(&t, c->totals.p, sizeof(Totals), hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    out->rays_extension = t.raysExtension;
    out->rays_shadow = t.raysShadow;
    out->rays_generated = t.raysGenerated;
    out->shade_hits = t.shadeHits;
    out->deposits = t.deposits;
    out->deposits_shadow = t.depositsShadow;
    out->samples = c->spp;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->evStart, c->evStop) == hipSuccess)
        c->msLastRender = ms;
    else
        (void)hipGetLastError(); // nothing rendered yet: the events were never recorded; do not leave that error for the next call to find
    out->ms_last_render = c->msLastRender;
    out->ms_intersect = c->msIntersect;
    out->ms_shade = c->msShade;
    out->ms_shadow = c->msShadow;
    out->ms_gen = c->msGen;
    out->packet_launches = c->packetLaunches;
    out->gen_launches = c->genLaunches;
    out->bundle_launches = c->bundleLaunches;
    out->ms_packet = c->msPacket;
    out->stack_need = c->dyn[c->active].stackNeed;
    out->folded_instances = c->dyn[c->active].foldedInstances;
    out->entered_instances = c->dyn[c->active].enteredInstances;
    out->batch_samples = c->batchSamples;
    out->probe_batches = c->probeBatches;
    out->first_pass_ext_ratio = (float)c->ratioExt, out->first_pass_shadow_ratio = (float)c->ratioShadow;
    out->general_route = c->dyn[c->active].generalRoute ? 1u : 0u;
    out->team_launches = c->teamLaunches;
    return PT_OK;
}

int pt_stats_reset(pt_ctx* c)
{
    if (!c)
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->totals.p, 0, sizeof(Totals), c->stream));
    c->packetLaunches = c->genLaunches = c->bundleLaunches = c->teamLaunches = 0;
    return PT_OK;
}

int pt_profile_kernels(pt_ctx* c, int enable)
{
    if (!c)
        return PT_ERR_INVALID;
    c->profile = enable != 0;
    return PT_OK;
}

int pt_reduce_accum(pt_ctx* c, void* nccl_comm, int root)
{
    if (!c || !nccl_comm)
        return PT_ERR_INVALID;
    // RCCL is bound at first use, and to the copy the process has ALREADY loaded if there is one (the
    // communicator comes from the caller's RCCL; a second copy of the library would not know it).
    // PTAMD_RCCL_LIB names a library explicitly.
    typedef int (*reduce_fn)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
    static reduce_fn reduce = nullptr;
    if (!reduce) {
        void* lib = nullptr;
        if (const char* path = getenv("PTAMD_RCCL_LIB"))
            lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        const char* names[] = { "librccl.so.1", "librccl.so" };
        for (int pass = 0; pass < 2 && !lib; pass++)
            for (const char* n : names)
                if (!lib)
                    lib = dlopen(n, RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : RTLD_GLOBAL));
        if (lib)
            reduce = (reduce_fn)dlsym(lib, "ncclReduce");
        if (!reduce)
            return fail(c, PT_ERR_UNSUPPORTED, "pt_reduce_accum: librccl not found (%s)", dlerror() ? dlerror() : "no ncclReduce symbol");
    }
    HIPCHK(c, hipSetDevice(c->device));
    float* buf = (float*)c->accum; // in place on the root (sendbuff == recvbuff is allowed there), send-only elsewhere
    const size_t count = (size_t)c->cfg.width * c->cfg.height * 4;
#ifdef PT_HAVE_RCCL_HEADER
    constexpr int kFloat = (int)ncclFloat, kSum = (int)ncclSum;
    static_assert(kFloat == 7 && kSum == 0, "the values this file falls back to where rccl.h is absent at build time");
#else
    constexpr int kFloat = 7, kSum = 0; // ncclFloat, ncclSum (rccl.h: ncclDataType_t / ncclRedOp_t)
#endif
    // (all four floats of every pixel travel, although w is unused: the accumulator is one contiguous float4 array and a strided reduce
    // would be three collectives instead of one; 33 MB at 1080p is 0.2 ms on one xGMI link)
    const int rc = reduce(buf, buf, count, kFloat, kSum, root, nccl_comm, c->stream);
    if (rc != 0)
        return fail(c, PT_ERR_HIP, "pt_reduce_accum: ncclReduce returned %d", rc);
    return PT_OK;
}

// ---- kernel-granular hooks ---------------------------------------------------------------------

int pt_intersect(pt_ctx* c, const pt_rays_soa* rays, uint32_t n, int any_hit, pt_hits_soa* hits, uint32_t repeat, float* ms_out)
{
    return guarded(c, "pt_intersect", [&]() -> int {
    if (!c || !rays || !hits || n == 0)
        return PT_ERR_INVALID;
    if (!c->haveStatic || !c->haveDynamic)
        return fail(c, PT_ERR_STATE, "pt_intersect: upload the scene first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensureSpill(c);
    if (rc)
        return rc;
    std::vector<float4> hO(n), hD(n), hC(n);
    for (uint32_t i = 0; i < n; i++) {
        hO[i] = make_float4(rays->ox[i], rays->oy[i], rays->oz[i], any_hit ? rays->tmax[i] : 0.f);
        hD[i] = make_float4(rays->dx[i], rays->dy[i], rays->dz[i], 0.f); // pixel / flags = 0
        hC[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    DevBuf<float4> dO, dD, dC, dH, dAcc;
    DevBuf<int32_t> dI;
    DevBuf<uint32_t> dOcc;
    DevBuf<Control> dCtl;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x) {
        if (e == hipSuccess)
            e = x;
    };
    chk(dO.alloc(n)), chk(dD.alloc(n)), chk(dC.alloc(n)), chk(dH.alloc(n)), chk(dI.alloc(n)), chk(dOcc.alloc(n)), chk(dAcc.alloc(1)), chk(dCtl.alloc(1));
    if (e == hipSuccess) {
        chk(hipMemcpy(dO.p, hO.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemcpy(dD.p, hD.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemcpy(dC.p, hC.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemsetAsync(dAcc.p, 0, sizeof(float4), c->stream));
    }
    float msTotal = 0;
    repeat = std::max(repeat, 1u);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    chk(hipEventCreate(&e0));
    chk(hipEventCreate(&e1));
    for (uint32_t r = 0; r < repeat && e == hipSuccess; r++) {
        Control ctl {}; // a control block of its own: n entries in pass 0's queue, cursors at zero
        ctl.extCount[0] = ctl.shadowCount[0] = n;
        chk(hipMemcpyAsync(dCtl.p, &ctl, sizeof(ctl), hipMemcpyHostToDevice, c->stream));
        chk(hipStreamSynchronize(c->stream)); // `ctl` is a stack object
        TraceArgs a = traceArgsBase(c);
        a.parityShadow = 0;
        a.rayO = dO.p, a.rayD = dD.p, a.rayC = dC.p;
        a.hit = dH.p, a.inst = dI.p, a.accum = AccumView { dAcc.p, nullptr, nullptr, 0u }, a.occluded = dOcc.p;
        a.ctl = dCtl.p, a.pass = 0;
        chk(hipEventRecord(e0, c->stream));
        if (c->dyn[c->active].packetOk && (c->packetUse & 4u))
            launchPacket(c, any_hit != 0, a);
        else
            launchTrace(c, any_hit != 0, a);
        chk(hipEventRecord(e1, c->stream));
        chk(hipStreamSynchronize(c->stream));
        chk(hipGetLastError());
        float ms = 0;
        chk(hipEventElapsedTime(&ms, e0, e1));
        msTotal += ms;
    }
    if (e == hipSuccess) {
        if (any_hit) {
            std::vector<uint32_t> occ(n);
            chk(hipMemcpy(occ.data(), dOcc.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; i++)
                hits->prim[i] = (int32_t)occ[i];
        } else {
            std::vector<float4> h(n);
            std::vector<int32_t> in(n);
            chk(hipMemcpy(h.data(), dH.p, n * sizeof(float4), hipMemcpyDeviceToHost));
            chk(hipMemcpy(in.data(), dI.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; i++) {
                hits->t[i] = h[i].x, hits->u[i] = h[i].y, hits->v[i] = h[i].z;
                int32_t prim;
                std::memcpy(&prim, &h[i].w, 4);
                hits->prim[i] = prim;
                hits->inst[i] = (in[i] >= 0 && (size_t)in[i] < c->dyn[c->active].instanceTopNode.size()) ? (int32_t)c->dyn[c->active].instanceTopNode[in[i]] : -1;
            }
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dO.release(), dD.release(), dC.release(), dH.release(), dI.release(), dOcc.release(), dAcc.release(), dCtl.release();
    if (ms_out)
        *ms_out = msTotal / (float)repeat;
    HIPCHK(c, e);
    return PT_OK;
    });
}

int pt_gen_rays(pt_ctx* c, uint32_t sample, uint32_t n, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz, uint32_t* pixel)
{
    return guarded(c, "pt_gen_rays", [&]() -> int {
    if (!c || n == 0)
        return PT_ERR_INVALID;
    if (!c->haveCamera)
        return fail(c, PT_ERR_STATE, "pt_gen_rays: set the camera first");
    if (n > c->numOwned)
        return fail(c, PT_ERR_INVALID, "pt_gen_rays: n exceeds the owned pixels");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensureQueues(c);
    if (rc)
        return rc;
    if (n > (c->q0Small ? c->capExt : c->capacity))
        return fail(c, PT_ERR_INVALID, "pt_gen_rays: n exceeds the queue capacity");
    FrameParams fp = frameParams(c, sample);
    launchGen(c, fp, 0, 0, n, 0, 0);
    std::vector<float4> o(n), d(n);
    HIPCHK(c, hipMemcpyAsync(o.data(), c->rays[0].o.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(d.data(), c->rays[0].d.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->control.p, 0, sizeof(Control), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; i++) {
        if (ox) ox[i] = o[i].x;
        if (oy) oy[i] = o[i].y;
        if (oz) oz[i] = o[i].z;
        if (dx) dx[i] = d[i].x;
        if (dy) dy[i] = d[i].y;
        if (dz) dz[i] = d[i].z;
        if (pixel)
            std::memcpy(&pixel[i], &o[i].w, 4);
    }
    return PT_OK;
    });
}

// The first pass of one batch exactly as pt_render issues it -- camera rays generated inside the traversal kernel where pt_render does
// that, walked as bundles where it does that -- with the rays and the hit records read back in queue order.
int pt_primary_pass(pt_ctx* c, uint32_t sample, uint32_t batch, uint32_t n, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz, uint32_t* pixel,
    pt_hits_soa* hits)
{
    return guarded(c, "pt_primary_pass", [&]() -> int {
    if (!c || !hits || batch == 0 || n == 0)
        return PT_ERR_INVALID;
    if (!c->haveStatic || !c->haveDynamic || !c->haveCamera)
        return fail(c, PT_ERR_STATE, "pt_primary_pass: scene (static + dynamic) and camera must be set first");
    if (parityMode(c) || c->cfg.max_active_rays != 0)
        return fail(c, PT_ERR_UNSUPPORTED, "pt_primary_pass: the fixed schedule only");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensureQueues(c)) || (rc = ensureSpill(c)))
        return rc;
    if (batch > c->planes || n != c->numOwned * batch)
        return fail(c, PT_ERR_INVALID, "pt_primary_pass: batch exceeds samples_in_flight, or n != owned pixels * batch");
    if (c->q0Small)
        return fail(c, PT_ERR_UNSUPPORTED, "pt_primary_pass: the hook queues whole camera rays; this context's first queue holds directions only (pt_config.ext_queue_fraction)");
    if (n > (1u << 26)) // every entry comes back to the host (68 B each): a hook for tests, not for 531 M-entry batches
        return fail(c, PT_ERR_INVALID, "pt_primary_pass: %u entries -- the hook reads everything back, use it on small frames (<= 64 M entries)", n);
    FrameParams fp = batchFrameParams(c, sample, batch);
    const bool coherentFirst = firstPassCoherent(c, fp, batch);
    const bool packetsFirst = coherentFirst && c->dyn[c->active].packetOk && (c->packetUse & 1u);
    const bool fused = packetsFirst && PT_FUSED_PRIMARY && !(c->cfg.flags & PT_FLAG_QUEUE_PRIMARY_RAYS);
    if (fused)
        hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, c->stream, &c->control.p->extCount[0], &c->control.p->generated, n);
    else
        launchGen(c, fp, 0, 0, n, 0, 0);
    launchIntersect(c, 0, 0, coherentFirst, fused ? &fp : nullptr);
    std::vector<float4> o(n), d(n), h(n);
    std::vector<int32_t> in(n);
    HIPCHK(c, hipMemcpyAsync(o.data(), c->rays[0].o.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(d.data(), c->rays[0].d.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h.data(), c->hitH.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(in.data(), c->hitInst.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->control.p, 0, sizeof(Control), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const std::vector<uint32_t>& top = c->dyn[c->active].instanceTopNode;
    for (uint32_t i = 0; i < n; i++) {
        if (ox) ox[i] = o[i].x;
        if (oy) oy[i] = o[i].y;
        if (oz) oz[i] = o[i].z;
        if (dx) dx[i] = d[i].x;
        if (dy) dy[i] = d[i].y;
        if (dz) dz[i] = d[i].z;
        if (pixel)
            std::memcpy(&pixel[i], &o[i].w, 4);
        hits->t[i] = h[i].x, hits->u[i] = h[i].y, hits->v[i] = h[i].z;
        int32_t prim;
        std::memcpy(&prim, &h[i].w, 4);
        hits->prim[i] = prim;
        hits->inst[i] = (in[i] >= 0 && (size_t)in[i] < top.size()) ? (int32_t)top[in[i]] : -1;
    }
    return PT_OK;
    });
}

int pt_shade_batch(pt_ctx* c, pt_shade_batch_io* io)
{
    return guarded(c, "pt_shade_batch", [&]() -> int {
    if (!c || !io || io->n == 0)
        return PT_ERR_INVALID;
    if (!c->haveStatic || !c->haveDynamic)
        return fail(c, PT_ERR_STATE, "pt_shade_batch: upload the scene first");
    if (parityMode(c))
        return fail(c, PT_ERR_UNSUPPORTED, "pt_shade_batch uses the counter PRNG keying");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = io->n;
    // instance index from the reported top-level leaf
    std::vector<int32_t> topToInst;
    const std::vector<uint32_t>& instanceTopNode = c->dyn[c->active].instanceTopNode;
    for (size_t k = 0; k < instanceTopNode.size(); k++) {
        if (instanceTopNode[k] >= topToInst.size())
            topToInst.resize(instanceTopNode[k] + 1, -1);
        topToInst[instanceTopNode[k]] = (int32_t)k;
    }
    std::vector<float4> hO(n), hD(n), hT(n), hH(n);
    std::vector<int32_t> hI(n);
    for (uint32_t i = 0; i < n; i++) {
        float fp, ff, fprim;
        uint32_t px = io->pixel[i], fl = (io->flags[i] & 0xFFu) | (io->bounce[i] << 8);
        int32_t prim = io->prim[i];
        std::memcpy(&fp, &px, 4), std::memcpy(&ff, &fl, 4), std::memcpy(&fprim, &prim, 4);
        hO[i] = make_float4(io->ox[i], io->oy[i], io->oz[i], fp);
        hD[i] = make_float4(io->dx[i], io->dy[i], io->dz[i], ff);
        hT[i] = make_float4(io->thr_r[i], io->thr_g[i], io->thr_b[i], 0.f);
        hH[i] = make_float4(prim >= 0 ? io->t[i] : INFINITY, io->u[i], io->v[i], fprim);
        int32_t ti = io->inst[i];
        hI[i] = (ti >= 0 && (size_t)ti < topToInst.size()) ? topToInst[ti] : -1;
        if (prim >= 0 && (hI[i] < 0 || (uint32_t)prim >= c->st->numTris))
            return fail(c, PT_ERR_INVALID, "pt_shade_batch: entry %u has an invalid prim/inst", i);
    }
    RayQueueBuf in, out, stagedDummy;
    ShadowQueueBuf sh;
    DevBuf<float4> dH, dAcc;
    DevBuf<int32_t> dI;
    DevBuf<uint32_t> dCtl;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x) {
        if (e == hipSuccess)
            e = x;
    };
    chk(in.o.alloc(n)), chk(in.d.alloc(n)), chk(in.thr.alloc(n)), chk(out.o.alloc(n)), chk(out.d.alloc(n)), chk(out.thr.alloc(n));
    chk(sh.o.alloc(n)), chk(sh.d.alloc(n)), chk(sh.c.alloc(n)), chk(dH.alloc(n)), chk(dI.alloc(n)), chk(dCtl.alloc(5));
    const size_t npix = (size_t)c->cfg.width * c->cfg.height;
    chk(dAcc.alloc(npix));
    // per-entry outputs are needed, so every entry is shaded as its own 1-entry queue slice
    (void)stagedDummy;
    if (e == hipSuccess) {
        chk(hipMemcpy(in.o.p, hO.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemcpy(in.d.p, hD.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemcpy(in.thr.p, hT.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemcpy(dH.p, hH.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        chk(hipMemcpy(dI.p, hI.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    std::vector<float4> acc(npix);
    FrameParams fp = frameParams(c, io->sample);
    for (uint32_t i = 0; i < n && e == hipSuccess; i++) {
        // pixel-indexed accumulator: clear only the touched pixel
        uint32_t ctl[5] = { 1, 0, 0, 0, 0 }; // inCount, outCount, shadowCount, shadeHits, deposits
        chk(hipMemcpyAsync(dCtl.p, ctl, sizeof(ctl), hipMemcpyHostToDevice, c->stream));
        chk(hipMemsetAsync(dAcc.p + io->pixel[i], 0, sizeof(float4), c->stream));
        ShadeArgs a {};
        a.sc = c->scene;
        a.fp = fp;
        a.in = { in.o.p + i, in.d.p + i, in.thr.p + i };
        a.hits = { dH.p + i, dI.p + i };
        a.out = { out.o.p + i, out.d.p + i, out.thr.p + i };
        a.shadow = { sh.o.p + i, sh.d.p + i, sh.c.p + i };
        a.accum = AccumView { dAcc.p, nullptr, nullptr, 0u };
        a.inCount = dCtl.p, a.outCount = dCtl.p + 1, a.shadowCount = dCtl.p + 2, a.shadeHits = dCtl.p + 3, a.deposits = dCtl.p + 4;
        if (generalShading(c))
            hipLaunchKernelGGL((k_shade<false, true>), dim3(1), dim3(64), 0, c->stream, a);
        else
            hipLaunchKernelGGL((k_shade<false, false>), dim3(1), dim3(64), 0, c->stream, a);
        uint32_t back[4];
        float4 px;
        chk(hipMemcpyAsync(back, dCtl.p, sizeof(back), hipMemcpyDeviceToHost, c->stream));
        chk(hipMemcpyAsync(&px, dAcc.p + io->pixel[i], sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        chk(hipStreamSynchronize(c->stream));
        io->out_alive[i] = back[1];
        io->shadow_alive[i] = back[2];
        io->radiance[3 * i] = px.x, io->radiance[3 * i + 1] = px.y, io->radiance[3 * i + 2] = px.z;
    }
    if (e == hipSuccess) {
        std::vector<float4> o(n), d(n), t(n), so(n), sd(n), sc(n);
        chk(hipMemcpy(o.data(), out.o.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        chk(hipMemcpy(d.data(), out.d.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        chk(hipMemcpy(t.data(), out.thr.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        chk(hipMemcpy(so.data(), sh.o.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        chk(hipMemcpy(sd.data(), sh.d.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        chk(hipMemcpy(sc.data(), sh.c.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            if (io->out_alive[i]) {
                io->nox[i] = o[i].x, io->noy[i] = o[i].y, io->noz[i] = o[i].z;
                io->ndx[i] = d[i].x, io->ndy[i] = d[i].y, io->ndz[i] = d[i].z;
                io->nthr_r[i] = t[i].x, io->nthr_g[i] = t[i].y, io->nthr_b[i] = t[i].z;
                uint32_t fl;
                std::memcpy(&fl, &d[i].w, 4);
                io->nflags[i] = fl & 0xFFu;
            }
            if (io->shadow_alive[i]) {
                io->sox[i] = so[i].x, io->soy[i] = so[i].y, io->soz[i] = so[i].z, io->slen[i] = so[i].w;
                io->sdx[i] = sd[i].x, io->sdy[i] = sd[i].y, io->sdz[i] = sd[i].z;
                io->sc_r[i] = sc[i].x, io->sc_g[i] = sc[i].y, io->sc_b[i] = sc[i].z;
            }
        }
    }
    in.o.release(), in.d.release(), in.thr.release(), out.o.release(), out.d.release(), out.thr.release();
    sh.o.release(), sh.d.release(), sh.c.release(), dH.release(), dI.release(), dCtl.release(), dAcc.release();
    HIPCHK(c, e);
    return PT_OK;
    });
}

} // extern "C"
