// Persistent-wave two-level BVH traversal for gfx950 (closest-hit and any-hit).
//
// What it computes is the reference's traceRay (assets/cl/scene.cl:61-271): top-level stack walk
// with the child whose box centre is nearer to the ray origin visited first, instance transform
// without renormalisation (so t is shared between spaces), the zero-component fix-up, bottom-level
// ordered traversal with Moeller-Trumbore leaves, `t < closestT` updates and any-hit early out.
//
// How it runs is CDNA4-specific:
//  * persistent waves: the grid is sized to the machine (blocks/CU x 256 CUs); each 64-lane wave
//    pulls rays from the queue with ONE atomicAdd per refill (ballot + popcount ranks the idle
//    lanes) and refills when enough lanes have gone idle, so long rays do not hold 63 lanes hostage;
//  * both traversal stacks live in LDS, laid out [entry][lane] so a wave's push/pop touches 64
//    consecutive dwords (conflict-free ds_read/ds_write_b32); deeper entries spill to a
//    lane-interleaved region in global memory (never reached by the benchmark scenes);
//  * the top-level BVH and the instance table are staged in LDS once per workgroup when they fit
//    (TOP_LDS): a ray visits ~13 top-level nodes, each a dependent pop -> node -> children chain that
//    would otherwise be three global-memory round trips;
//  * one 64-byte PairNode fetch per bottom-level step (both child boxes), reciprocal directions
//    computed once per ray / instance entry, 48-byte pre-digested triangles.
#pragma once
#include "pt_math.h"

namespace ptd {

#ifndef PT_LEAF_BATCH
#define PT_LEAF_BATCH 4
#endif
#ifndef PT_REFILL_IDLE
#define PT_REFILL_IDLE 20
#endif
#ifndef PT_TRACE_MIN_WAVES
#define PT_TRACE_MIN_WAVES 1
#endif
constexpr int kLdsStack = 20; // bottom-level entries kept in LDS per lane
constexpr int kSpillStack = 52; // further bottom-level entries in global memory
constexpr int kTopLdsStack = 8; // top-level entries kept in LDS per lane
constexpr int kTopStack = 64; // top-level entries overall (LDS + global spill)
constexpr int kTraceBlock = 256;
constexpr int kRefillIdleLanes = PT_REFILL_IDLE; // refill the wave once this many lanes are idle
constexpr int kTopLdsNodes = 128; // top-level nodes / instances staged in LDS (TOP_LDS variant)
constexpr int kTopLdsInstances = 64;
constexpr int kLeafBatch = PT_LEAF_BATCH; // triangles fetched together in a leaf

struct TraceArgs {
    SceneDev sc;
    // closest-hit: rays from (rayO, rayD), results to (hit, inst)
    // any-hit: rays from (shO, shD, shC); unoccluded contributions are added to accum
    const float4* rayO;
    const float4* rayD;
    const float4* rayC;
    float4* hit;
    int32_t* inst;
    AccumView accum;
    uint32_t* occluded; // optional (test hook): 1/0 per shadow ray
    const uint32_t* count; // number of queue entries (device word)
    uint32_t* cursor; // fetch cursor (device word, zero at launch)
    uint32_t* spill; // (kSpillStack + kTopStack) * totalThreads dwords
    uint32_t totalThreads;
    uint32_t parityShadow; // any-hit: entries carry a FINISHED flag in rayC.w (reference semantics)
    uint32_t numTopNodes, numInstances;
};

template <bool ANY_HIT, bool TOP_LDS>
__global__ void __launch_bounds__(kTraceBlock, PT_TRACE_MIN_WAVES) k_trace(TraceArgs a)
{
    __shared__ uint32_t ldsStack[kTraceBlock / 64][kLdsStack][64];
    __shared__ uint32_t ldsTopStack[kTraceBlock / 64][kTopLdsStack][64];
    __shared__ TopNode sTop[TOP_LDS ? kTopLdsNodes : 1];
    __shared__ Instance sInst[TOP_LDS ? kTopLdsInstances : 1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
    uint32_t* const spillBottom = a.spill + gtid; // entry e at spillBottom[e * totalThreads]
    uint32_t* const spillTop = a.spill + (size_t)kSpillStack * a.totalThreads + gtid;
    const uint32_t total = a.totalThreads;
    const uint32_t count = *a.count;
    const SceneDev& sc = a.sc;

    if (TOP_LDS) { // stage the top level once per workgroup (16-byte copies)
        const float4* srcN = (const float4*)sc.top;
        float4* dstN = (float4*)sTop;
        for (uint32_t i = threadIdx.x; i < a.numTopNodes * 2u; i += kTraceBlock)
            dstN[i] = srcN[i];
        const float4* srcI = (const float4*)sc.instances;
        float4* dstI = (float4*)sInst;
        for (uint32_t i = threadIdx.x; i < a.numInstances * 4u; i += kTraceBlock)
            dstI[i] = srcI[i];
        __syncthreads();
    }
    const TopNode* const topNodes = TOP_LDS ? sTop : sc.top;
    const Instance* const instances = TOP_LDS ? sInst : sc.instances;

    auto pushTop = [&](int slot, uint32_t v) {
        if (slot < kTopLdsStack)
            ldsTopStack[wave][slot][lane] = v;
        else
            spillTop[(size_t)(slot - kTopLdsStack) * total] = v;
    };
    auto popTop = [&](int slot) -> uint32_t {
        return slot < kTopLdsStack ? ldsTopStack[wave][slot][lane] : spillTop[(size_t)(slot - kTopLdsStack) * total];
    };
    auto popBottom = [&](int slot) -> uint32_t {
        return slot < kLdsStack ? ldsStack[wave][slot][lane] : spillBottom[(size_t)(slot - kLdsStack) * total];
    };

    bool active = false;
    bool exhausted = false; // wave-uniform: queue has no more rays
    uint32_t rayIdx = 0;
    V3 o = mk(0.f), d = mk(0.f), id = mk(0.f); // world-space ray and 1/d (only used where d != 0)
    V3 to = mk(0.f), td = mk(0.f), itd = mk(0.f); // instance-space origin, direction, 1/direction
    float tClosest = 0.f, tMax = 0.f, hu = 0.f, hv = 0.f;
    int hprim = -1, hinst = -1, curInst = -1;
    uint32_t cur = kRefNone;
    int sp = 0, tsp = 0;
    float4 contrib = make_float4(0, 0, 0, 0);
    uint32_t pixel = 0;

    // ---- per-wave ray packets ------------------------------------------------------------------
    // A wave claims 64 consecutive queue entries with one atomicAdd and loads them with fully coalesced
    // 16 B/lane reads into registers (lane i holds entry i).  Idle lanes are handed their next ray by
    // ballot rank through a cross-lane read (ds_bpermute) of those registers, so a refill costs no
    // memory round trip; as soon as a packet has been handed out completely the next one is requested,
    // and its loads are in flight while the rays just handed out are traversed.
    float4 poolO = make_float4(0, 0, 0, 0), poolD = poolO, poolC = poolO;
    uint32_t poolBase = 0, poolNext = 0, poolEnd = 0; // wave-uniform
    auto requestPacket = [&]() {
        uint32_t base = 0;
        if (lane == 0)
            base = atomicAdd(a.cursor, 64u);
        base = __shfl(base, 0);
        poolBase = base;
        poolNext = 0;
        poolEnd = base < count ? min(64u, count - base) : 0u;
        if (lane < poolEnd) {
            poolO = a.rayO[base + lane];
            poolD = a.rayD[base + lane];
            if (ANY_HIT)
                poolC = a.rayC[base + lane];
        }
    };
    requestPacket();

    while (true) {
        // ---- hand rays to idle lanes ----------------------------------------------------------
        if (!exhausted) {
            const unsigned long long idle = __ballot(!active);
            const int nIdle = __popcll(idle);
            if (nIdle >= kRefillIdleLanes) {
                const uint32_t avail = poolEnd - poolNext;
                if (avail == 0u) {
                    exhausted = true; // the request issued after the last hand-out came back empty
                } else {
                    const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    const int e = (int)min(poolNext + rank, 63u);
                    float4 ro, rd, rc;
                    ro.x = __shfl(poolO.x, e), ro.y = __shfl(poolO.y, e), ro.z = __shfl(poolO.z, e), ro.w = __shfl(poolO.w, e);
                    rd.x = __shfl(poolD.x, e), rd.y = __shfl(poolD.y, e), rd.z = __shfl(poolD.z, e), rd.w = __shfl(poolD.w, e);
                    if (ANY_HIT)
                        rc.x = __shfl(poolC.x, e), rc.y = __shfl(poolC.y, e), rc.z = __shfl(poolC.z, e), rc.w = __shfl(poolC.w, e);
                    if (!active && rank < avail) {
                        const uint32_t idx = poolBase + (uint32_t)e;
                        bool live = true;
                        if (ANY_HIT) {
                            contrib = rc;
                            pixel = asU(rd.w);
                            tMax = ro.w;
                            if (a.parityShadow && (asU(contrib.w) & FLAG_FINISHED))
                                live = false;
                        } else {
                            tMax = INFINITY;
                            if (asU(rd.w) & FLAG_FINISHED) { // parity mode keeps finished rays in the queue
                                live = false;
                                a.hit[idx] = make_float4(INFINITY, 0.f, 0.f, asF(0xFFFFFFFFu));
                                a.inst[idx] = -1;
                            }
                        }
                        if (live) {
                            rayIdx = idx;
                            o = xyz(ro);
                            d = xyz(rd);
                            id = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                            tClosest = tMax;
                            hprim = -1;
                            hinst = -1;
                            hu = hv = 0.f;
                            cur = kRefNone;
                            sp = 0;
                            tsp = 1;
                            pushTop(0, sc.topRoot);
                            active = true;
                        }
                    }
                    poolNext += min((uint32_t)nIdle, avail);
                    if (poolNext == poolEnd)
                        requestPacket();
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted)
                break;
            continue;
        }

        // ---- traverse until enough lanes are idle again -----------------------------------
        // Each iteration runs ONE of the three step kinds -- the one most lanes are waiting for (ballot
        // majority vote) -- instead of serialising all three bodies for a third of the lanes each.
        while (true) {
            const bool wantTop = active && cur == kRefNone;
            const bool wantInner = active && cur != kRefNone && refCount(cur) == 0u;
            const bool wantLeaf = active && cur != kRefNone && refCount(cur) != 0u;
            const int nTop = __popcll(__ballot(wantTop)), nInner = __popcll(__ballot(wantInner)), nLeaf = __popcll(__ballot(wantLeaf));
            const int kind = (nInner >= nTop && nInner >= nLeaf) ? 1 : (nLeaf >= nTop ? 2 : 0);
            {
                if (kind == 0 && wantTop) {
                    // ---------------- top level (scene.cl:105-159) ----------------
                    if (tsp == 0) {
                        // ray finished: closestT != maxT decides hit/miss (scene.cl:257)
                        if (ANY_HIT) {
                            if (a.occluded)
                                a.occluded[rayIdx] = 0u;
                            float4* ap = a.accum.at(asU(contrib.w) >> 16, pixel); // one live path per entry: plain RMW
                            float4 px = *ap;
                            px.x += contrib.x, px.y += contrib.y, px.z += contrib.z;
                            *ap = px;
                        } else {
                            a.hit[rayIdx] = make_float4(hprim >= 0 ? tClosest : INFINITY, hu, hv, asF((uint32_t)hprim));
                            a.inst[rayIdx] = hinst;
                        }
                        active = false;
                    } else {
                        const uint32_t ni = popTop(--tsp);
                        const TopNode tn = topNodes[ni];
                        // slab test in world space with the per-axis d != 0 guard (bvh.cl:36-73)
                        float tmin = -INFINITY, tmax = INFINITY;
                        if (d.x != 0.0f) {
                            const float t1 = (tn.lo.x - o.x) * id.x, t2 = (tn.hi.x - o.x) * id.x;
                            tmin = fmaxf(tmin, fminf(t1, t2));
                            tmax = fminf(tmax, fmaxf(t1, t2));
                        }
                        if (d.y != 0.0f) {
                            const float t1 = (tn.lo.y - o.y) * id.y, t2 = (tn.hi.y - o.y) * id.y;
                            tmin = fmaxf(tmin, fminf(t1, t2));
                            tmax = fminf(tmax, fmaxf(t1, t2));
                        }
                        if (d.z != 0.0f) {
                            const float t1 = (tn.lo.z - o.z) * id.z, t2 = (tn.hi.z - o.z) * id.z;
                            tmin = fmaxf(tmin, fminf(t1, t2));
                            tmax = fminf(tmax, fmaxf(t1, t2));
                        }
                        if (tmax >= tmin && tmax >= 0.f && tmin < tClosest) {
                            const uint32_t ca = asU(tn.lo.w), cb = asU(tn.hi.w);
                            if (cb == 0xFFFFFFFFu) { // leaf: enter the instance
                                const Instance in = instances[ca];
                                to = mk(in.r0.x * o.x + in.r0.y * o.y + in.r0.z * o.z + in.r0.w,
                                    in.r1.x * o.x + in.r1.y * o.y + in.r1.z * o.z + in.r1.w,
                                    in.r2.x * o.x + in.r2.y * o.y + in.r2.z * o.z + in.r2.w);
                                td = mk(in.r0.x * d.x + in.r0.y * d.y + in.r0.z * d.z,
                                    in.r1.x * d.x + in.r1.y * d.y + in.r1.z * d.z,
                                    in.r2.x * d.x + in.r2.y * d.y + in.r2.z * d.z);
                                // NO_PARALLEL_RAYS fix-up (scene.cl:123-137)
                                if (td.x == 0.0f) td.x = FLT_MIN;
                                if (td.y == 0.0f) td.y = FLT_MIN;
                                if (td.z == 0.0f) td.z = FLT_MIN;
                                if (to.x == 0.0f) to.x = -FLT_MIN;
                                if (to.y == 0.0f) to.y = -FLT_MIN;
                                if (to.z == 0.0f) to.z = -FLT_MIN;
                                itd = mk(1.0f / td.x, 1.0f / td.y, 1.0f / td.z);
                                cur = in.rootRef;
                                curInst = (int)ca; // instance index; pt_intersect reports the top-level leaf
                                sp = 0;
                            } else { // inner: nearer box centre is visited first (scene.cl:141-157)
                                const TopNode l = topNodes[ca];
                                const TopNode r = topNodes[cb];
                                const V3 lv = (xyz(l.lo) + xyz(l.hi)) / 2.0f - o;
                                const V3 rv = (xyz(r.lo) + xyz(r.hi)) / 2.0f - o;
                                const bool leftFirst = dot(lv, lv) < dot(rv, rv);
                                pushTop(tsp, leftFirst ? cb : ca);
                                pushTop(tsp + 1, leftFirst ? ca : cb);
                                tsp += 2;
                            }
                        }
                    }
                } else if (kind == 1 && wantInner) {
                    // ---------------- bottom level, inner step (scene.cl:197-231) ----------------
                    const PairNode* np = &sc.nodes[refIndex(cur)];
                    const float4 bx = np->bx, by = np->by, bz = np->bz;
                    const uint32_t lref = np->left, rref = np->right;
                    const float lx0 = (bx.x - to.x) * itd.x, lx1 = (bx.y - to.x) * itd.x;
                    const float rx0 = (bx.z - to.x) * itd.x, rx1 = (bx.w - to.x) * itd.x;
                    const float ly0 = (by.x - to.y) * itd.y, ly1 = (by.y - to.y) * itd.y;
                    const float ry0 = (by.z - to.y) * itd.y, ry1 = (by.w - to.y) * itd.y;
                    const float lz0 = (bz.x - to.z) * itd.z, lz1 = (bz.y - to.z) * itd.z;
                    const float rz0 = (bz.z - to.z) * itd.z, rz1 = (bz.w - to.z) * itd.z;
                    const float ltmin = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fminf(lz0, lz1));
                    const float ltmax = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fmaxf(lz0, lz1));
                    const float rtmin = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fminf(rz0, rz1));
                    const float rtmax = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fmaxf(rz0, rz1));
                    const bool lvis = ltmax >= ltmin && ltmax >= 0.f && ltmin < tClosest;
                    const bool rvis = rtmax >= rtmin && rtmax >= 0.f && rtmin < tClosest;
                    if (lvis && rvis) {
                        const bool leftFirst = ltmin < rtmin;
                        const uint32_t farRef = leftFirst ? rref : lref;
                        if (sp < kLdsStack)
                            ldsStack[wave][sp][lane] = farRef;
                        else
                            spillBottom[(size_t)(sp - kLdsStack) * total] = farRef;
                        sp++;
                        cur = leftFirst ? lref : rref;
                    } else if (lvis) {
                        cur = lref;
                    } else if (rvis) {
                        cur = rref;
                    } else if (sp > 0) {
                        cur = popBottom(--sp);
                    } else {
                        cur = kRefNone;
                    }
                } else if (kind == 2 && wantLeaf) {
                    // ---------------- bottom level, leaf (scene.cl:168-195, shapes.cl:20-72) -------
                    const uint32_t first = refIndex(cur), n = refCount(cur);
                    bool done = false;
                    // Triangles are fetched kLeafBatch at a time so that their 3 x 16 B loads are all in
                    // flight together (one memory round trip per batch instead of one per triangle); the
                    // tests then run in index order against the running closestT, as the reference does.
                    for (uint32_t k0 = 0; k0 < n && !done; k0 += kLeafBatch) {
                        float4 ta[kLeafBatch], tb[kLeafBatch];
                        float tc[kLeafBatch];
#pragma unroll
                        for (int j = 0; j < kLeafBatch; j++) {
                            const TriIsect* tp = &sc.tris[first + min(k0 + (uint32_t)j, n - 1u)];
                            ta[j] = tp->a, tb[j] = tp->b, tc[j] = tp->c.x;
                        }
#pragma unroll
                        for (int j = 0; j < kLeafBatch; j++) {
                            if (k0 + (uint32_t)j < n && !done) {
                                const V3 v0 = mk(ta[j].x, ta[j].y, ta[j].z), e1 = mk(ta[j].w, tb[j].x, tb[j].y), e2 = mk(tb[j].z, tb[j].w, tc[j]);
                                const V3 P = cross(td, e2);
                                const float det = dot(e1, P);
                                const float inv = 1.f / det;
                                const V3 T = to - v0;
                                const float u = dot(T, P) * inv;
                                const V3 Q = cross(T, e1);
                                const float v = dot(td, Q) * inv;
                                const float t = dot(e2, Q) * inv;
                                const bool hit = !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f)
                                    && t > 0.f && t < tClosest;
                                if (hit) {
                                    if (ANY_HIT) {
                                        done = true;
                                    } else {
                                        tClosest = t;
                                        hu = u;
                                        hv = v;
                                        hprim = (int)(first + k0 + (uint32_t)j);
                                        hinst = curInst;
                                    }
                                }
                            }
                        }
                    }
                    if (ANY_HIT && done) { // occluded: nothing to deposit
                        if (a.occluded)
                            a.occluded[rayIdx] = 1u;
                        active = false;
                    } else if (sp > 0) {
                        cur = popBottom(--sp);
                    } else {
                        cur = kRefNone;
                    }
                }
            }
            const int nActive = __popcll(__ballot(active));
            if (nActive == 0 || (!exhausted && nActive <= 64 - kRefillIdleLanes))
                break;
        }
    }
}

} // namespace ptd
