// Team traversal for launches that do not fill the machine (round 5): FOUR lanes per ray.
//
// What it computes is k_trace's result -- the reference's traceRay (assets/cl/scene.cl:61-271), closest hit or any hit, on a scene that is ONE
// world-space tree (instances copied at upload: the library's default) -- with the same box and triangle arithmetic.  Why it exists: the late
// passes of a 1-spp frame (RayTracer::rayTrace, src/main.cpp:106-119) hold 20-200 k rays for 459 k lanes, and k_trace then takes 80-100 us however
// few rays there are: one lane walks one ray, ~40 dependent node fetches of ~1.4 us each in a row, and nothing is there to hide them behind
// (measured in round 5: serving inner nodes and leaves in the same iteration, from one fetch, does not move that floor -- the chain is the ray's,
// not the wave's).  Here the four lanes of a team pop FOUR entries of the ray's stack at once -- one shared LDS stack per team -- test four nodes or
// leaves in one iteration and push what they found behind each other (counts exchanged inside the quad by DPP): the chain shrinks from "nodes a ray
// visits" towards "depth of the tree".  The price is order: the ray's subtrees are no longer visited strictly nearest first, so a closest-hit ray
// tests a few more triangles (each member culls with the team's best distance so far) and exact-t ties may resolve differently than in k_trace; any-hit
// rays do not care.  A wave = 16 teams (LDS: 64 stack entries x 16 teams x 4 B = 4 KB per wave).
//
// Stack discipline: while the team's stack is short every member pops (breadth); beyond kTeamDfsAbove entries only member 0 pops (depth first:
// from any state the stack then grows by at most the tree's worst-case depth-first need, which ptamd.hip checks against kTeamStack before it
// launches this kernel).
#pragma once
#include "pt_trace.h"

namespace ptd {

constexpr int kTeamBlock = 256;
#ifndef PT_TEAM_STACK
#define PT_TEAM_STACK 64
#define PT_TEAM_DFS_ABOVE 16
#endif
constexpr int kTeamStack = PT_TEAM_STACK; // entries per team (4 KB of LDS per wave)
constexpr int kTeamDfsAbove = PT_TEAM_DFS_ABOVE; // more entries than this: depth first (one pop per iteration); four members need no more than four entries to be busy
constexpr uint32_t kTeamStackNeedMax = (uint32_t)(kTeamStack - kTeamDfsAbove - 12); // breadth adds <= 12 per iteration (4 popped, <= 16 pushed)
#ifndef PT_TEAM_MIN_WAVES
#define PT_TEAM_MIN_WAVES 6 // 4 / 5 / 6 / 7 waves per SIMD: 720p frames 0.82-0.94 / - / 0.77-0.90 / 0.77-0.92 ms (more teams in flight: fewer rounds per launch; 7 spills)
#endif

// lane j of the caller's quad (j = 0..3)
__device__ inline uint32_t quadLane(uint32_t v, int j)
{
    switch (j) {
    case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false);
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, false);
    }
}
__device__ inline uint32_t quadXor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); } // lanes 1 0 3 2
__device__ inline uint32_t quadXor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); } // lanes 2 3 0 1

template <bool ANY_HIT>
__global__ void __launch_bounds__(kTeamBlock, PT_TEAM_MIN_WAVES) k_trace_team(TraceArgs a)
{
    __shared__ uint32_t ldsStack[kTeamBlock / 64][kTeamStack][16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t team = lane >> 2, m = lane & 3u;
    const uint32_t count = ANY_HIT ? a.ctl->shadowCount[a.pass] : a.ctl->extCount[a.pass];
    const SceneDev& sc = a.sc;
    const uint32_t totalTeams = gridDim.x * (kTeamBlock / 64) * 16u;
    uint32_t nextRay = (blockIdx.x * (kTeamBlock / 64) + wave) * 16u + team; // team-uniform: rays nextRay, nextRay + totalTeams, ...
    bool active = false; // team-uniform
    int sp = 0; // team-uniform: entries on the team's stack
    uint32_t rayIdx = 0;
    V3 co = mk(0.f), cd = mk(0.f, 0.f, 1.f), cid = mk(1.f);
    float tBest = 0.f; // the member's own closest hit so far (any hit: the ray's length)
    float hu = 0.f, hv = 0.f;
    int hprim = -1;
    uint32_t unoccluded = 0; // any hit: rays this lane deposited (member 0 only)

    while (true) {
        if (!active && nextRay < count) { // (team-uniform condition)
            rayIdx = nextRay;
            nextRay += totalTeams;
            float4 ro = a.rayO[rayIdx], rd = a.rayD[rayIdx];
            // zero components are nudged as at k_trace's hand-out (NO_PARALLEL_RAYS, scene.cl:123-137)
            if (rd.x == 0.0f) rd.x = FLT_MIN;
            if (rd.y == 0.0f) rd.y = FLT_MIN;
            if (rd.z == 0.0f) rd.z = FLT_MIN;
            if (ro.x == 0.0f) ro.x = -FLT_MIN;
            if (ro.y == 0.0f) ro.y = -FLT_MIN;
            if (ro.z == 0.0f) ro.z = -FLT_MIN;
            co = xyz(ro), cd = xyz(rd);
            cid = mk(rcpSlab(cd.x), rcpSlab(cd.y), rcpSlab(cd.z));
            tBest = ANY_HIT ? ro.w : INFINITY;
            hu = hv = 0.f, hprim = -1;
            if (m == 0u)
                ldsStack[wave][0][team] = sc.rootRef;
            sp = 1;
            active = true;
        }
        if (__ballot(active) == 0ull)
            break; // no team of the wave has a ray, and none is left for any of them
        // ---- pop: member m takes entry sp - 1 - m (breadth), or member 0 alone the top (depth first) ---------------------------
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int take = active ? min(sp, sp > kTeamDfsAbove ? 1 : 4) : 0;
        const bool mine = (int)m < take;
        const uint32_t cur = mine ? ldsStack[wave][sp - 1 - (int)m][team] : kRefFinish;
        sp -= take;
        const uint32_t kind = refCount(cur);
        const bool isInner = mine && kind == 0u, isLeaf = mine && kind != 0u && kind != kRefSpecial;
        // ---- ONE fetch per iteration: 96 bytes from where the lane stands -- a node (64 B + the 32 behind it) or the (up to) two triangles of its
        // leaf (2 x 48 B: leaves hold two at most since round 5; ptamd.hip keeps one record of slack behind both arrays)
        uint4 line[6];
        {
#if PT_OFFSET32 // base + 32-bit byte offset (pt_trace.h)
            const uint4* at = isLeaf ? (const uint4*)((const char*)sc.tris + (size_t)(uint32_t)(refIndex(cur) * (uint32_t)sizeof(TriIsect)))
                                     : (const uint4*)((const char*)sc.wide + (size_t)(uint32_t)((isInner ? refIndex(cur) : 0u) << 6));
#else
            const uint4* at = isLeaf ? (const uint4*)&sc.tris[refIndex(cur)] : (const uint4*)&sc.wide[isInner ? refIndex(cur) : 0u];
#endif
#pragma unroll
            for (int q = 0; q < 6; q++)
                line[q] = at[q];
        }
        // the distance every member culls with: the team's best so far
        float tCull = tBest;
        if (!ANY_HIT) {
            tCull = fminf(tCull, asF(quadXor1(asU(tCull))));
            tCull = fminf(tCull, asF(quadXor2(asU(tCull))));
        }
        uint32_t ref[4] = { kRefFinish, kRefFinish, kRefFinish, kRefFinish };
        uint32_t nPush = 0;
        if (isInner) {
            // -------- k_trace's inner step (pt_trace.h): four quantised child boxes, entry / exit planes by the sign of the direction ----------
            const uint4 A = line[0], B = line[1], D = line[3];
            const uint4 C = line[2];
            const float ax = asF(A.w) * cid.x, ay = asF(C.z) * cid.y, az = asF(C.w) * cid.z;
            const float bx = (asF(A.x) - co.x) * cid.x, by = (asF(A.y) - co.y) * cid.y, bz = (asF(A.z) - co.z) * cid.z;
            const bool nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
            const uint32_t qnx = nx ? B.y : B.x, qfx = nx ? B.x : B.y;
            const uint32_t qny = ny ? B.w : B.z, qfy = ny ? B.z : B.w;
            const uint32_t qnz = nz ? C.y : C.x, qfz = nz ? C.x : C.y;
            float key[4];
            ref[0] = D.x, ref[1] = D.y, ref[2] = D.z, ref[3] = D.w;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 qx = { (float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu) };
                const f2 qy = { (float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu) };
                const f2 qz = { (float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu) };
                const f2 tx = planePair(qx, ax, bx), ty = planePair(qy, ay, by), tz = planePair(qz, az, bz);
                const float tmin = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                const float tmax = fminf(fminf(tx.y, ty.y), tz.y);
                const bool vis = tmax >= tmin && tmax >= 0.f && tmin < tCull; // the accept test of bvh.cl:72,114 on the (slightly larger) quantised box
                key[k] = vis ? tmin : INFINITY;
            }
#define PT_TSWAP(i, j)                                   \
    {                                                    \
        const bool sw = key[j] < key[i];                 \
        const float tk = sw ? key[j] : key[i];           \
        key[j] = sw ? key[i] : key[j];                   \
        key[i] = tk;                                     \
        const uint32_t tr = sw ? ref[j] : ref[i];        \
        ref[j] = sw ? ref[i] : ref[j];                   \
        ref[i] = tr;                                     \
    }
            // visible children first, nearest first (any hit: any order of the visible ones would do; the network also compacts them)
            PT_TSWAP(0, 1) PT_TSWAP(2, 3) PT_TSWAP(0, 2) PT_TSWAP(1, 3) PT_TSWAP(1, 2)
#undef PT_TSWAP
            nPush = (key[0] < INFINITY ? 1u : 0u) + (key[1] < INFINITY ? 1u : 0u) + (key[2] < INFINITY ? 1u : 0u) + (key[3] < INFINITY ? 1u : 0u);
        }
        bool occluder = false;
        if (isLeaf) {
            // -------- leaf (scene.cl:168-195), Moeller-Trumbore as k_trace spells it (shapes.cl:20-72) --------------------------------------------
            const uint32_t first = refIndex(cur), n = kind;
            for (uint32_t k = 0; k < n; k++) {
                float4 ta, tb;
                float tcx;
                if (k < 2u) {
                    const uint4 la = k == 0u ? line[0] : line[3], lb = k == 0u ? line[1] : line[4];
                    ta = make_float4(asF(la.x), asF(la.y), asF(la.z), asF(la.w)), tb = make_float4(asF(lb.x), asF(lb.y), asF(lb.z), asF(lb.w));
                    tcx = asF(k == 0u ? line[2].x : line[5].x);
                } else { // (leaves of more than two triangles: PTAMD_MAX_LEAF, parity mode never gets here)
                    const TriIsect* tp = &sc.tris[first + k];
                    ta = tp->a, tb = tp->b, tcx = tp->c.x;
                }
                const V3 v0 = mk(ta.x, ta.y, ta.z), e1 = mk(ta.w, tb.x, tb.y), e2 = mk(tb.z, tb.w, tcx);
                float det, u, v, t;
                triangleTest(co, cd, v0, e1, e2, &det, &u, &v, &t);
                const bool hit = !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f && t < tBest;
                if (hit) {
                    if (ANY_HIT) {
                        occluder = true;
                        break;
                    }
                    tBest = t, hu = u, hv = v, hprim = (int)(first + k);
                }
            }
        }
        // ---- the team's pushes behind each other: member 3's children lowest, member 0's on top, each far child first ---------------------------
        const uint32_t c0 = quadLane(nPush, 0), c1 = quadLane(nPush, 1), c2 = quadLane(nPush, 2), c3 = quadLane(nPush, 3);
        const uint32_t below = (m < 3u ? c3 : 0u) + (m < 2u ? c2 : 0u) + (m < 1u ? c1 : 0u); // pushes of the members with a higher number
        const uint32_t total = c0 + c1 + c2 + c3;
        bool teamOccluded = false;
        if (ANY_HIT)
            teamOccluded = ((__ballot(occluder) >> (lane & ~3u)) & 0xFull) != 0ull;
        if (nPush && !teamOccluded) {
            const int at = sp + (int)below;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((uint32_t)k < nPush)
                    ldsStack[wave][at + (int)nPush - 1 - k][team] = ref[k]; // ref[0] = the nearest: on top of this member's run
        }
        sp += (int)total;
        // the pushes above are read by OTHER lanes of the team at the next pop: a wave-level release fence + barrier here, an acquire fence at the pop (no
        // instruction on a wave -- LDS operations of one wave complete in order -- but the compiler may no longer cache or move the accesses across the back-edge)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- done?  (any hit: an occluder ends the ray at once) ----------------------------------------------------------------------------------
        if (active && (sp == 0 || teamOccluded)) {
            if (ANY_HIT) {
                if (m == 0u) {
                    if (a.occluded)
                        a.occluded[rayIdx] = teamOccluded ? 1u : 0u;
                    if (!teamOccluded) { // the deposit of intersectShadows (kernel.cl:132-135): one live path per accumulator entry, plain RMW
                        const float4 contrib = a.rayC[rayIdx];
                        const uint32_t pixel = asU(a.rayD[rayIdx].w);
                        float4* ap = a.accum.at(asU(contrib.w) >> 16, pixel);
                        float4 px = *ap;
                        px.x += contrib.x, px.y += contrib.y, px.z += contrib.z;
                        *ap = px;
                        unoccluded++;
                    }
                }
            } else {
                // the members' own best hits folded: smaller t wins, at equal t the smaller triangle index (two exchanges inside the quad)
                float bt = tBest, bu = hu, bv = hv;
                int bp = hprim;
#pragma unroll
                for (int step = 0; step < 2; step++) {
                    const float ot = asF(step ? quadXor2(asU(bt)) : quadXor1(asU(bt)));
                    const float ou = asF(step ? quadXor2(asU(bu)) : quadXor1(asU(bu)));
                    const float ov = asF(step ? quadXor2(asU(bv)) : quadXor1(asU(bv)));
                    const int op = (int)(step ? quadXor2((uint32_t)bp) : quadXor1((uint32_t)bp));
                    const bool better = op >= 0 && (bp < 0 || ot < bt || (ot == bt && op < bp));
                    bt = better ? ot : bt, bu = better ? ou : bu, bv = better ? ov : bv, bp = better ? op : bp;
                }
                if (m == 0u) {
                    int hinst = -1;
                    if (bp >= 0) { // a hit on a world-space copy of an instance: back to (original triangle, instance)
                        const float4 tc = sc.tris[bp].c;
                        bp = (int)asU(tc.y);
                        hinst = (int)asU(tc.z);
                    }
                    a.hit[rayIdx] = make_float4(bp >= 0 ? bt : INFINITY, bu, bv, asF((uint32_t)bp));
                    a.inst[rayIdx] = hinst;
                }
            }
            active = false;
            sp = 0;
        }
    }
    if (ANY_HIT) {
        // unoccluded rays = accumulator updates of the launch (pt_stats.deposits_shadow): one atomic per wave
        uint32_t n = unoccluded;
        for (int ofs = 32; ofs > 0; ofs >>= 1)
            n += (uint32_t)__shfl_down((int)n, ofs);
        if (lane == 0u && n)
            atomicAdd(&a.ctl->depositsShadow, n);
    }
}

} // namespace ptd
