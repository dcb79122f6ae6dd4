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

#include "pt_context.h"
#include "pt_convert.h"
#include "pt_schedule.h"


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

// Diagnostic (bench.py): GB/s (read + written) of a float4 grid-stride device copy of `bytes` bytes, the best of `repeat` timed runs after a warm-up.
int pt_debug_copy_bandwidth(pt_ctx* c, size_t bytes, uint32_t repeat, float* gbps_out)
{
    return guarded(c, "pt_debug_copy_bandwidth", [&]() -> int {
    if (!c || !gbps_out || bytes < (1u << 20))
        return PT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = bytes / sizeof(float4);
    DevBuf<float4> src, dst;
    HIPCHK(c, src.alloc(n));
    HIPCHK(c, dst.alloc(n));
    HIPCHK(c, hipMemsetAsync(src.p, 0x3c, n * sizeof(float4), c->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    const uint32_t blocks = (uint32_t)c->numCUs * 4u; // 1 024 workgroups of 256 on an MI355X (pt_bake.h: the best of the shapes tried)
    float best = 0.f;
    for (uint32_t r = 0; r <= std::max(repeat, 1u); r++) { // (run 0: warm-up)
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(k_copy_probe, dim3(blocks), dim3(256), 0, c->stream, src.p, dst.p, n);
        (void)hipEventRecord(e1, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms > 0.f)
            best = std::max(best, (float)(2.0 * (double)n * sizeof(float4) / (ms * 1e-3) / 1e9));
    }
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    src.release(), dst.release();
    *gbps_out = best;
    return PT_OK;
    });
}

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
    d.instRootBase = h.instRootBase, d.numInstRoots = (uint32_t)h.instRoots.size(), d.instFoldCount = h.generalRoute ? h.enteredGeneral : (uint32_t)h.instFold.size(); // (the general route: how many entered instances want their 3 x 4 rows)
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
            // of an earlier batch of this epoch, or a short PROBE batch whose report is waited for (16 samples per pixel: the queues hold at least what those
            // can emit, ensureQueues; its samples count like any others) -- and the batch is cut so that it fits with 3 % to spare.
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
            if (!probe) // (16 samples per pixel always fit: ensureQueues' floor)
                batch = std::min(batch, std::max(safeBatch(c), std::min(16u, c->planes)));
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
        a.outCap = a.shadowCap = 0xFFFFFFFFu; // (the hook's own queues: one slot per entry)
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
