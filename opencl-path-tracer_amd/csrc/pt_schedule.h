// The launch schedule of the C-ABI: what replaces the reference's RayTracer::traceRays (src/raytracer.cpp:289-430) -- queues and their sizes, the choice of
// traversal kernel per launch, one batch of the fixed no-read-back schedule (renderSampleFixed), the reference-style refill loop (renderSampleRefill), the
// probe / batch sizing of queues smaller than a batch.  pt_render (ptamd.hip) drives it.  Included by ptamd.hip (one translation unit).
#pragma once

namespace {

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
            // ... but never less than the probe batch may emit: 16 samples per pixel, every path going on (16: the smallest batch whose camera rays go through
            // the bundle kernel, which is what queues them as directions only)
            const uint64_t floorEntries = std::min<uint64_t>(cap, ((16ull * c->numOwned + 63u) & ~63ull));
            c->capExt = (uint32_t)std::max<uint64_t>(c->capExt, floorEntries);
            c->capShadow = (uint32_t)std::max<uint64_t>(c->capShadow, floorEntries);
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
    // (the first queue's origin / throughput planes hold capExt >= 16 samples' worth of entries: a batch that queues whole camera rays -- fewer than 16 samples
    // per pixel: no bundles -- fits them; a larger one that does so all the same is a change of camera or scene state pt_render has re-made the queues for)
    if (c->q0Small && !derived && entries > c->capExt)
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
