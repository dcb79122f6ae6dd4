// The context of the C-ABI (include/ptamd.h): device buffers, the two static scenes and the two dynamic sets a context holds, error reporting.
// Included by ptamd.hip alone (one translation unit: pt_context.h, pt_convert.h, pt_schedule.h, then the C-ABI shell).
#pragma once

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
        // pt_upload_static_async (a rebuilt tree per frame): the host makes the topology only -- child references, the slots' box sources, the triangle
        // references -- and the device makes the records from the caller's own arrays, as after a refit (k_refit_nodes: exact boxes and quantised planes;
        // k_refit_tris: intersection and shading records).  The host's `wide` planes, `boxes` and `fat` are then not filled in (nothing reads them: a
        // conversion that runs again makes everything anew, refitWideOnHost / buildFat re-make them from the pair boxes where a host-side refit needs them).
        bool deviceMakesRecords = false;
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

} // namespace
