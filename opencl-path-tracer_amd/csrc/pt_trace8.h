// Persistent-wave traversal of the 8-wide compressed BVH (pt_wide8.h) for gfx950: closest-hit and any-hit.
//
// Same contract as k_trace (pt_trace.h): the reference's traceRay (assets/cl/scene.cl:61-271) -- instance
// transform without renormalisation, zero-component fix-up, two-sided Moeller-Trumbore with the reference's
// accept tests, any-hit early out.  What differs is the node: one visit is FIVE 16-byte loads for EIGHT
// children, the pending work of a visited node is kept as one 8-byte group (base index + bit mask) instead of
// one stack entry per child, and children are taken in `slot ^ octant` order instead of being sorted.
//
// Per lane: G = node group  (x: index of the first inner child, y: pending-children bits 31..24 | imask 7..0)
//           T = item group  (x: index of the first item,        y: pending-item bits 23..0)
// A step of the hot loop is either a NODE step (take the next child of G, fetch it, test its 8 boxes -> new G
// and T, the rest of the old G goes onto the stack) or a TRIANGLE step (all triangles of T); which one runs is
// decided per iteration by ballot majority, as in k_trace.  Everything rare -- top-level items (enter an
// instance / test a baked world-space triangle), leaving an instance, writing the result -- parks the lane
// until the hot loop breaks and is served in batches outside it.
//
// STATUS: parity-green (all -m gpu tests pass with -DPT_BVH8=3) but NOT the default: on the benchmark scene it
// visits 11.0 / 12.8 nodes per primary / secondary ray against 14.0 / 17.1 for the 4-wide tree, i.e. 55 / 64
// 16-byte node loads per ray against 56 / 68 -- no relief for the vector-memory bottleneck -- while a node visit
// costs ~220 VALU instructions instead of ~120 and top-level items add special passes (5.0 vs 3.6 per ray):
// 4.9 Grays/s in-kernel against 6.9 for k_trace.
#pragma once
#include "pt_trace.h"
#include "pt_wide8.h"

namespace ptd {

#ifndef PT_BVH8
#define PT_BVH8 0 // bit 0: closest-hit launches use the 8-wide tree, bit 1: any-hit launches (0: 4-wide k_trace, the default)
#endif
#ifndef PT_TRACE8_MIN_WAVES
#define PT_TRACE8_MIN_WAVES 5 // 95 VGPRs without spills
#endif
#ifndef PT_LDS_STACK8
#define PT_LDS_STACK8 10
#endif
constexpr int kLdsStack8 = PT_LDS_STACK8; // 8-byte groups kept in LDS per lane
constexpr int kSpillStack8 = kSpillStack / 2; // further groups in global memory (two dwords each: the region k_trace uses)
constexpr uint32_t kGroupFinish = 0xFFFFFFFEu;

template <bool ANY_HIT>
__global__ void __launch_bounds__(kTraceBlock, PT_TRACE8_MIN_WAVES) k_trace8(TraceArgs a)
{
    __shared__ uint2 ldsStack[kTraceBlock / 64][kLdsStack8][64];
    __shared__ float ldsWorld[kTraceBlock / 64][6][64]; // world-space origin and direction per lane
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
    uint2* const spill = (uint2*)a.spill + gtid; // group e at spill[e * totalThreads]
    const uint32_t total = a.totalThreads;
    const uint32_t count = *a.count;
    const SceneDev& sc = a.sc;

#ifdef PT_TRACE_STATS
    unsigned long long statAcc[24] = {};
    PT_TIC(tKernel);
#endif
    bool active = false;
    bool exhausted = false; // wave-uniform: queue has no more rays
    uint32_t rayIdx = 0;
    V3 co = mk(0.f), cd = mk(0.f), cid = mk(0.f), coid = mk(0.f);
    uint32_t octinv4 = 0; // (7 ^ negative-direction mask) replicated into 4 bytes
    float tClosest = 0.f, hu = 0.f, hv = 0.f;
    int hprim = -1, hinst = -1, curInst = -1;
    uint32_t Gx = 0u, Gy = 0u, Tx = 0u, Ty = 0u; // the two groups, as scalars: uint2 objects ended up in scratch memory
    int sp = 0;

    auto push = [&](uint32_t x, uint32_t y) __attribute__((always_inline)) {
        if (sp < kLdsStack8)
            ldsStack[wave][sp][lane] = make_uint2(x, y);
        else
            spill[(size_t)(sp - kLdsStack8) * total] = make_uint2(x, y);
        sp++;
    };
    // next pending group: a node group goes to G, an item group or the leave-instance sentinel to T
    auto popNext = [&]() __attribute__((always_inline)) {
        uint32_t ex = kGroupFinish, ey = 0u; // nothing left: the end-of-traversal marker
        if (sp > 0) {
            sp--;
            const uint2 e = ldsStack[wave][min(sp, kLdsStack8 - 1)][lane];
            ex = e.x, ey = e.y;
            if (sp >= kLdsStack8) { // volatile: keeps this (rare) global read apart from the LDS read above
                const volatile uint32_t* sv = (const volatile uint32_t*)&spill[(size_t)(sp - kLdsStack8) * total];
                ex = sv[0], ey = sv[1];
            }
        }
        // One straight-line update of all four registers with value selects.  (Written as `if (node) G = e; else
        // T = e;` hipcc selected between the ADDRESSES of the variables and kept them in scratch memory.)
        const bool node = (ey & 0xFF000000u) != 0u;
        Gx = node ? ex : Gx;
        Gy = node ? ey : 0u;
        Tx = node ? 0u : ex;
        Ty = node ? 0u : ey;
    };
    auto setRay = [&](V3 o, V3 d) __attribute__((always_inline)) {
        co = o;
        cd = d;
        cid = mk(rcpSlab(d.x), rcpSlab(d.y), rcpSlab(d.z));
        coid = mk(-o.x * cid.x, -o.y * cid.y, -o.z * cid.z);
        const uint32_t oct = (cid.x < 0.f ? 1u : 0u) | (cid.y < 0.f ? 2u : 0u) | (cid.z < 0.f ? 4u : 0u);
        octinv4 = (7u ^ oct) * 0x01010101u;
    };
    // Moeller-Trumbore (shapes.cl:20-72) in the current ray space; returns true when the ANY_HIT ray is done
    auto testTriangle = [&](uint32_t idx) __attribute__((always_inline)) -> bool {
        const TriIsect* tp = &sc.tris8[idx];
        const float4 ta = tp->a, tb = tp->b;
        const float tcx = tp->c.x;
        const V3 v0 = mk(ta.x, ta.y, ta.z), e1 = mk(ta.w, tb.x, tb.y), e2 = mk(tb.z, tb.w, tcx);
        const V3 P = cross(cd, e2);
        const float det = dot(e1, P);
        const float inv = rcpFast(det);
        const V3 Tv = co - v0;
        const float u = dot(Tv, P) * inv;
        const V3 Q = cross(Tv, e1);
        const float v = dot(cd, Q) * inv;
        const float t = dot(e2, Q) * inv;
        const bool hit = !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f && t < tClosest;
        if (hit) {
            if (ANY_HIT)
                return true;
            tClosest = t;
            hu = u;
            hv = v;
            hprim = (int)idx;
            hinst = curInst;
        }
        return false;
    };

    // ---- claiming queue entries: as in k_trace ---------------------------------------------------
    uint32_t poolBase = 0, poolNext = 0, poolEnd = 0; // wave-uniform
    const uint32_t totalWaves = total >> 6, gwave = gtid >> 6;
    const uint32_t spanSize = 64u * min(8u, max(1u, count / (totalWaves * 64u * 8u)));
    uint32_t spanNext = gwave * 64u, spanEnd = spanNext + 64u;
    auto requestPacket = [&]() {
        if (spanNext >= spanEnd) {
            uint32_t base = 0xFFFFFFC0u;
            if (gwave * 64u < count) {
                if (lane == 0)
                    base = atomicAdd(a.cursor, spanSize);
                base = totalWaves * 64u + __shfl(base, 0);
            }
            spanNext = base;
            spanEnd = base + spanSize;
        }
        const uint32_t base = spanNext;
        spanNext += 64u;
        poolBase = base;
        poolNext = 0;
        poolEnd = base < count ? min(64u, count - base) : 0u;
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
                    exhausted = true;
                } else {
                    const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    if (!active && rank < avail) {
                        const uint32_t idx = poolBase + poolNext + rank;
                        // State word first, ray afterwards and only for live entries.  (Do not fold the ray loads back in
                        // front of the flag test: with `tMax = ro.w` ahead of the parity-mode flag load, hipcc 7.2 let that
                        // load overwrite the register of ro.w and dropped the tClosest assignment on that path.)
                        const float4 rd = a.rayD[idx];
                        uint32_t state = asU(rd.w); // closest-hit: parity mode keeps finished rays in the queue
                        if (ANY_HIT) // contribution and pixel stay in the queue until the ray turns out unoccluded
                            state = a.parityShadow ? asU(a.rayC[idx].w) : 0u;
                        const bool live = (state & FLAG_FINISHED) == 0u;
                        if (!ANY_HIT && !live) {
                            a.hit[idx] = make_float4(INFINITY, 0.f, 0.f, asF(0xFFFFFFFFu));
                            a.inst[idx] = -1;
                        }
                        if (live) {
                            const float4 ro = a.rayO[idx];
                            float tMax = ANY_HIT ? ro.w : INFINITY;
                            asm volatile("" : "+v"(tMax)); // pins the value in a register of its own
                            rayIdx = idx;
                            ldsWorld[wave][0][lane] = ro.x, ldsWorld[wave][1][lane] = ro.y, ldsWorld[wave][2][lane] = ro.z;
                            ldsWorld[wave][3][lane] = rd.x, ldsWorld[wave][4][lane] = rd.y, ldsWorld[wave][5][lane] = rd.z;
                            setRay(xyz(ro), xyz(rd));
                            tClosest = tMax;
                            hprim = -1;
                            hinst = -1;
                            curInst = -1;
                            hu = hv = 0.f;
                            Gx = sc.root8, Gy = 1u << 24; // "child 0 of a group that starts at the root"
                            Tx = 0u, Ty = 0u;
                            sp = 0;
                            active = true;
                        }
                    }
                    PT_STAT(8, 1);
                    PT_STAT(9, min((uint32_t)nIdle, avail));
                    poolNext += min((uint32_t)nIdle, avail);
                    if (poolNext == poolEnd)
                        requestPacket();
                }
            }
        }
        // ---- parked lanes: top-level items, leaving an instance, end of traversal ----------------------
        while (true) {
            const bool nodeReady = (Gy & 0xFF000000u) != 0u && Ty == 0u;
            const bool triReady = Ty != 0u && curInst >= 0;
            const bool wantSpecial = active && !nodeReady && !triReady;
            if (__ballot(wantSpecial) == 0ull)
                break;
            PT_STAT(4, 1);
            PT_STAT(7, __popcll(__ballot(wantSpecial)));
            if (wantSpecial) {
                if (Ty != 0u) {
                    // -------- one item of a top-level node (world space) ------------------------------------
                    const int bit = 31 - __clz((int)Ty);
                    Ty &= ~(1u << bit);
                    const uint32_t item = sc.items[Tx + (uint32_t)bit];
                    if (item & kItemBakedTriangle) {
                        if (testTriangle(item & ~kItemBakedTriangle)) { // occluded (ANY_HIT only)
                            if (a.occluded)
                                a.occluded[rayIdx] = 1u;
                            active = false;
                        } else if (Ty == 0u && (Gy & 0xFF000000u) == 0u) {
                            popNext();
                        }
                    } else {
                        // -------- enter instance `item` (scene.cl:116-139) ---------------------------------
                        if (Gy & 0xFF000000u)
                            push(Gx, Gy);
                        if (Ty != 0u)
                            push(Tx, Ty);
                        push(kGroupLeaveInstance, 0u);
                        const Instance in = sc.instances[item];
                        const V3 o = co, d = cd; // instances are only ever entered from world space
                        V3 to = mk(in.r0.x * o.x + in.r0.y * o.y + in.r0.z * o.z + in.r0.w, in.r1.x * o.x + in.r1.y * o.y + in.r1.z * o.z + in.r1.w,
                            in.r2.x * o.x + in.r2.y * o.y + in.r2.z * o.z + in.r2.w);
                        V3 td = mk(in.r0.x * d.x + in.r0.y * d.y + in.r0.z * d.z, in.r1.x * d.x + in.r1.y * d.y + in.r1.z * d.z,
                            in.r2.x * d.x + in.r2.y * d.y + in.r2.z * d.z);
                        // NO_PARALLEL_RAYS fix-up (scene.cl:123-137)
                        if (td.x == 0.0f) td.x = FLT_MIN;
                        if (td.y == 0.0f) td.y = FLT_MIN;
                        if (td.z == 0.0f) td.z = FLT_MIN;
                        if (to.x == 0.0f) to.x = -FLT_MIN;
                        if (to.y == 0.0f) to.y = -FLT_MIN;
                        if (to.z == 0.0f) to.z = -FLT_MIN;
                        setRay(to, td);
                        curInst = (int)item;
                        Gx = in.root8, Gy = 1u << 24;
                        Tx = 0u, Ty = 0u;
                    }
                } else if (Tx == kGroupLeaveInstance) {
                    // -------- back to world space ---------------------------------------------------------
                    setRay(mk(ldsWorld[wave][0][lane], ldsWorld[wave][1][lane], ldsWorld[wave][2][lane]),
                        mk(ldsWorld[wave][3][lane], ldsWorld[wave][4][lane], ldsWorld[wave][5][lane]));
                    curInst = -1;
                    popNext();
                } else if (Tx == kGroupFinish) {
                    // -------- ray finished: closestT != maxT decides hit/miss (scene.cl:257) ------------
                    if (ANY_HIT) {
                        if (a.occluded)
                            a.occluded[rayIdx] = 0u;
                        const float4 contrib = a.rayC[rayIdx];
                        const uint32_t pixel = asU(a.rayD[rayIdx].w);
                        float4* ap = a.accum.at(asU(contrib.w) >> 16, pixel); // one live path per entry: plain RMW
                        float4 px = *ap;
                        px.x += contrib.x, px.y += contrib.y, px.z += contrib.z;
                        *ap = px;
                    } else {
                        if (hprim >= 0) { // triangles were re-emitted in node order: back to the caller's numbering
                            const float4 tc = sc.tris8[hprim].c;
                            hprim = (int)asU(tc.y);
                            if (asU(tc.z) != 0xFFFFFFFFu)
                                hinst = (int)asU(tc.z); // world-space copy of a single-leaf instance
                        }
                        a.hit[rayIdx] = make_float4(hprim >= 0 ? tClosest : INFINITY, hu, hv, asF((uint32_t)hprim));
                        a.inst[rayIdx] = hinst;
                    }
                    active = false;
                } else {
                    popNext(); // nothing pending in the registers (e.g. right after a hand-out of an empty root)
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted)
                break;
            continue;
        }

        // ---- hot loop: node steps and triangle steps ---------------------------------------------------
        while (true) {
            const bool wantNode = active && (Gy & 0xFF000000u) != 0u && Ty == 0u;
            const bool wantTri = active && Ty != 0u && curInst >= 0;
            const int nNode = __popcll(__ballot(wantNode)), nTri = __popcll(__ballot(wantTri));
            const int nActive = __popcll(__ballot(active));
            const int nWork = nNode + nTri, nSpecial = nActive - nWork;
            if (nWork == 0 || nSpecial >= kParkedBreak || (!exhausted && 64 - nActive >= kRefillIdleLanes))
                break;
            PT_STAT(0, 1);
            PT_STAT(1, nWork);
            if (nNode >= nTri) {
                PT_STAT(2, 1);
                PT_STAT(5, nNode);
                PT_TIC(tInner);
                if (wantNode) {
                    // -------- take the next child of G (highest pending bit = nearest by octant order) -----------
                    const uint32_t pending = Gy;
                    const int bit = 31 - __clz((int)pending);
                    const uint32_t slot = (uint32_t)(bit - 24) ^ (octinv4 & 7u);
                    const uint32_t nodeIdx = Gx + (uint32_t)__popc(pending & 0xFFu & ((1u << slot) - 1u));
                    const uint4* np = (const uint4*)&sc.nodes8[nodeIdx];
                    const uint4 A = np[0], B = np[1], Q0 = np[2], Q1 = np[3], Q2 = np[4];
                    Gy = pending & ~(1u << bit);
                    if (Gy & 0xFF000000u)
                        push(Gx, Gy); // the siblings wait on the stack
                    // box plane = origin + 2^exp * q  =>  t = q * (2^exp / d) + (origin - o) / d : one FMA per plane
                    const float ax = asF((A.w & 0xFFu) << 23) * cid.x, ay = asF(((A.w >> 8) & 0xFFu) << 23) * cid.y,
                                az = asF(((A.w >> 16) & 0xFFu) << 23) * cid.z;
                    const float bx = fmaf(asF(A.x), cid.x, coid.x), by = fmaf(asF(A.y), cid.y, coid.y), bz = fmaf(asF(A.z), cid.z, coid.z);
                    const f2 ax2 = { ax, ax }, ay2 = { ay, ay }, az2 = { az, az }, bx2 = { bx, bx }, by2 = { by, by }, bz2 = { bz, bz };
                    const bool nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
                    // node layout: Q0 = qlox[2] qloy[2], Q1 = qloz[2] qhix[2], Q2 = qhiy[2] qhiz[2]
                    uint32_t hitmask = 0u;
                    // four children (one dword of each plane array) at a time
                    auto half = [&](uint32_t lox, uint32_t hix, uint32_t loy, uint32_t hiy, uint32_t loz, uint32_t hiz, uint32_t m4) __attribute__((always_inline)) {
                        // entry / exit plane bytes by ray-direction sign
                        const uint32_t qnx = nx ? hix : lox, qfx = nx ? lox : hix;
                        const uint32_t qny = ny ? hiy : loy, qfy = ny ? loy : hiy;
                        const uint32_t qnz = nz ? hiz : loz, qfz = nz ? loz : hiz;
                        // meta bytes of the four children at once: inner children get their bit at 24 + (slot ^ octinv),
                        // leaf children `count` bits from their item offset
                        const uint32_t inner4 = (((m4 & (m4 << 1)) & 0x10101010u) >> 4) * 0xFFu;
                        const uint32_t bitIndex4 = (m4 ^ (octinv4 & inner4)) & 0x1F1F1F1Fu;
                        const uint32_t childBits4 = (m4 >> 5) & 0x07070707u;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const f2 qx = { (float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu) };
                            const f2 qy = { (float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu) };
                            const f2 qz = { (float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu) };
                            const f2 tx = __builtin_elementwise_fma(qx, ax2, bx2), ty = __builtin_elementwise_fma(qy, ay2, by2),
                                     tz = __builtin_elementwise_fma(qz, az2, bz2);
                            const float tmin = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                            const float tmax = fminf(fminf(tx.y, ty.y), tz.y);
                            // accept test of bvh.cl:72,114 on the (slightly larger) quantised box; empty slots have no bits
                            const bool vis = tmax >= tmin && tmax >= 0.f && tmin < tClosest;
                            const uint32_t bits = ((childBits4 >> (8 * k)) & 0xFFu) << ((bitIndex4 >> (8 * k)) & 0xFFu);
                            hitmask |= vis ? bits : 0u;
                        }
                    };
                    half(Q0.x, Q1.z, Q0.z, Q2.x, Q1.x, Q2.z, B.z);
                    half(Q0.y, Q1.w, Q0.w, Q2.y, Q1.y, Q2.w, B.w);
                    Gx = B.x, Gy = (hitmask & 0xFF000000u) | (A.w >> 24);
                    Tx = B.y, Ty = hitmask & 0x00FFFFFFu;
                    if (hitmask == 0u)
                        popNext();
                }
                PT_TOC(11, tInner);
            } else {
                PT_STAT(3, 1);
                PT_STAT(6, nTri);
                PT_TIC(tLeaf);
                if (wantTri) {
                    // -------- all pending triangles of the item group (scene.cl:168-195) ----------------------
                    bool done = false;
                    while (Ty != 0u) {
                        const int bit = __ffs((int)Ty) - 1; // leaf order, as the reference walks a leaf
                        Ty &= ~(1u << bit);
                        PT_STAT(16, 1);
                        if (testTriangle(Tx + (uint32_t)bit)) {
                            done = true;
                            break;
                        }
                    }
                    if (ANY_HIT && done) { // occluded: nothing to deposit
                        if (a.occluded)
                            a.occluded[rayIdx] = 1u;
                        active = false;
                    } else if ((Gy & 0xFF000000u) == 0u) {
                        popNext();
                    }
                }
                PT_TOC(12, tLeaf);
            }
        }
    }
#ifdef PT_TRACE_STATS
    PT_TOC(10, tKernel);
    if (lane == 0)
        for (int i = 0; i < 24; i++)
            atomicAdd(&g_traceStats[i + (ANY_HIT ? 24 : 0)], statAcc[i]);
#endif
}

} // namespace ptd
