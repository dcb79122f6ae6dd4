// Shared descent: the 64 rays of a packet that leave ONE pixel's footprint walk from the root towards the surface they start on TOGETHER.
//
// The shadow rays of the primary hits and the first bounce's extension rays are half of a batch's traversal time, and
// about half of the inner steps k_trace (pt_trace.h) takes for them are at nodes that CONTAIN the ray's origin: every ray re-finds the
// surface it starts on, privately -- its own 64-byte node fetch, its own 24 byte -> float conversions, its own sort -- at 0.6-0.7 active
// lanes.  With up to 256 samples of a pixel adjacent in the queue (k_shade's compaction keeps a tile's order), the 64 consecutive
// entries of a packet start within a pixel's footprint of each other: that walk is the SAME walk for all of them.  Here a wave takes it
// once:
//   * the current node reference is wave-uniform (an SGPR); the node arrives by scalar loads (s_load_dwordx4 x 4: no vector-memory
//     instruction, no address arithmetic per lane);
//   * every lane tests the node's four child boxes against ITS OWN ray with exactly k_trace's arithmetic (the same visible set, bit for
//     bit: t = q * (2^e / d) + (origin - o) / d, accept test of bvh.cl:72,114) -- at 64 of 64 lanes, no vote, no sort;
//   * the wave then votes for the child that contains the origin of most of its rays (entry distance <= 0) and follows it; a lane
//     pushes the OTHER children its ray passes onto its own start stack (three entries, in registers: a ray leaves the way to its
//     origin with 1.8-1.9 stacked siblings on average) and stays with the wave; a lane whose origin lies outside the chosen child, or
//     whose start stack is full, drops out with the current node as its starting point (it will test that node again, privately);
//   * the walk ends at a leaf or an instance reference (the followers start THERE), or when fewer than PT_DESCENT_MIN_TOGETHER lanes
//     are left (a step here costs what it costs whatever the number of followers: below ~32 of them k_trace's private step is cheaper).
// What comes out is a START STATE per queue entry -- reference to continue with, number of stacked entries, the entries -- which the
// hand-out of k_trace<., ., true> takes instead of (root, empty stack).  The set of subtrees a ray visits is unchanged (the union of
// the start state is the tree minus the boxes its ray misses); only their order differs from k_trace's nearest-first order (stacked
// siblings pop in slot order, deepest level first), which the closest hit does not depend on except for exact-t ties, and an any-hit
// verdict not at all.
//
// Reference semantics: traceRay, scene.cl:61-271 (the descent is its loop at :141-233 for the nodes on the way to the origin).
#pragma once
#include "pt_packet.h"

#ifndef PT_DESCENT_MIN_TOGETHER
#define PT_DESCENT_MIN_TOGETHER 32
#endif
#ifndef PT_DESCENT_MIN_WAVES
#define PT_DESCENT_MIN_WAVES 8
#endif

namespace ptd {

constexpr int kDescendBlock = 256;

// start state of one queue entry: ONE 16-byte record -- x: the reference to continue with, y z w: up to three stacked entries in stack
// order (kRefNone: unused), so that the hand-out of k_trace reads it with one coalesced load per lane
struct DescendArgs {
    SceneDev sc;
    const float4* rayO;
    const float4* rayD;
    const uint32_t* count; // entries in the queue (a device word: nothing is read back)
    uint4* start;
};

#ifdef PT_TRACE_STATS
// [0] packets [1] shared steps [2] followers summed over the steps [3] rays [4] stacked entries at the end [5] rays that end on a leaf / instance
// reference [6] drop-outs because the start stack was full [7] drop-outs because the origin lay outside the chosen child [8] packets that ended for lack of followers
__device__ unsigned long long g_descendStats[16];
#endif

#ifndef PT_DESCENT_RUN
#define PT_DESCENT_RUN 8 // consecutive packets per wave: the packets of one pixel and of its neighbours walk (almost) the same path -- the nodes of the
#endif                   // previous walk are still in the scalar cache / L1
#ifndef PT_DESCENT_PREFETCH
#define PT_DESCENT_PREFETCH 0 // (measured: 16.0 instead of 12.7 ms of k_descend per batch -- the 14 v_readlane and the lane addressing cost more than the scalar loads' latency)
                              // 1: the four children of a node are fetched by the LANES (16 lanes x 4 bytes each per child) while the node is tested; the
#endif                        // chosen one is read out of that register lane by lane (v_readlane) -- no dependent scalar load per step

template <bool ANY_HIT>
__global__ void __launch_bounds__(kDescendBlock, PT_DESCENT_MIN_WAVES) k_descend(DescendArgs a)
{
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef const u4v __attribute__((address_space(4)))* ScalarU4; // uniform address + constant space = scalar loads
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = uni((blockIdx.x * kDescendBlock + threadIdx.x) >> 6);
    const uint32_t totalWaves = (gridDim.x * kDescendBlock) >> 6;
    const uint32_t count = uni(*a.count);
    const uint32_t packets = (count + 63u) >> 6;
    const SceneDev& sc = a.sc;
    const uint32_t rootRef = uni(sc.rootRef);
    const ScalarU4 wideS = (ScalarU4)(unsigned long long)sc.wide;
    const uint32_t childOfLane = lane >> 4, dwordOfLane = lane & 15u;
#ifdef PT_TRACE_STATS
    unsigned long long st[9] = {};
#endif
    for (uint32_t run = gwave * PT_DESCENT_RUN; run < packets; run += totalWaves * PT_DESCENT_RUN)
    for (uint32_t p = run; p < min(run + PT_DESCENT_RUN, packets); p++) {
        const uint32_t idx = p * 64u + lane;
        const bool valid = idx < count;
        const uint32_t e = min(idx, count - 1u);
        float4 ro = a.rayO[e], rd = a.rayD[e];
        // zero components are nudged exactly as at k_trace's hand-out (NO_PARALLEL_RAYS, scene.cl:123-137): the same ray, the same box tests
        if (rd.x == 0.0f) rd.x = FLT_MIN;
        if (rd.y == 0.0f) rd.y = FLT_MIN;
        if (rd.z == 0.0f) rd.z = FLT_MIN;
        if (ro.x == 0.0f) ro.x = -FLT_MIN;
        if (ro.y == 0.0f) ro.y = -FLT_MIN;
        if (ro.z == 0.0f) ro.z = -FLT_MIN;
        const V3 co = xyz(ro);
        const V3 cid = mk(rcpSlab(rd.x), rcpSlab(rd.y), rcpSlab(rd.z));
        const float tMax = ANY_HIT ? ro.w : INFINITY;
        const bool nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
        bool together = valid;
        uint32_t sp = 0u, s0 = kRefNone, s1 = kRefNone, s2 = kRefNone; // the lane's start stack, in stack order
        uint32_t myCur = rootRef; // together lanes: == cur
        uint32_t cur = rootRef; // wave-uniform
        u4v A = { 0u, 0u, 0u, 0u }, B = A, C = A, D = A;
        if (refCount(cur) == 0u) {
            const uint32_t ni = refIndex(cur);
            A = wideS[ni * 4u + 0u], B = wideS[ni * 4u + 1u], C = wideS[ni * 4u + 2u], D = wideS[ni * 4u + 3u];
        }
        unsigned long long tmask = __builtin_amdgcn_ballot_w64(together);
        while (refCount(cur) == 0u && __popcll(tmask) >= PT_DESCENT_MIN_TOGETHER) {
#if PT_DESCENT_PREFETCH
            // the four children, one dword per lane, on their way while this node is tested (a child that is a leaf or an instance is not a node)
            const uint32_t childRef = childOfLane == 0u ? D.x : (childOfLane == 1u ? D.y : (childOfLane == 2u ? D.z : D.w));
            uint32_t pref = 0u;
            if (refCount(childRef) == 0u)
                pref = __builtin_nontemporal_load((const uint32_t*)sc.wide + (size_t)refIndex(childRef) * 16u + dwordOfLane);
#endif
            // the box test of k_trace's inner step, operation by operation (pt_trace.h): the same planes, the same visible set
            const float ax = asF(A.w) * cid.x, ay = asF(C.z) * cid.y, az = asF(C.w) * cid.z;
            const float bx = (asF(A.x) - co.x) * cid.x, by = (asF(A.y) - co.y) * cid.y, bz = (asF(A.z) - co.z) * cid.z;
            const uint32_t qnx = nx ? B.y : B.x, qfx = nx ? B.x : B.y;
            const uint32_t qny = ny ? B.w : B.z, qfy = ny ? B.z : B.w;
            const uint32_t qnz = nz ? C.y : C.x, qfz = nz ? C.x : C.y;
            bool vis[4];
            unsigned long long inside[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 qx = { (float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu) };
                const f2 qy = { (float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu) };
                const f2 qz = { (float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu) };
                const f2 tx = planePair(qx, ax, bx), ty = planePair(qy, ay, by), tz = planePair(qz, az, bz);
                const float tmin = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                const float tmax = fminf(fminf(tx.y, ty.y), tz.y);
                vis[k] = tmax >= tmin && tmax >= 0.f && tmin < tMax; // bvh.cl:72,114
                inside[k] = __builtin_amdgcn_ballot_w64(together && vis[k] && tmin <= 0.f); // the ray starts inside this child's box
            }
            // the child that holds the origin of most rays (scalar)
            const uint32_t c0 = (uint32_t)__popcll(inside[0]), c1 = (uint32_t)__popcll(inside[1]), c2 = (uint32_t)__popcll(inside[2]), c3 = (uint32_t)__popcll(inside[3]);
            const uint32_t b01 = c1 > c0 ? 1u : 0u, b23 = c3 > c2 ? 3u : 2u;
            const uint32_t m01 = max(c0, c1), m23 = max(c2, c3);
            const uint32_t best = m23 > m01 ? b23 : b01;
            const uint32_t followers = max(m01, m23);
            if (followers < PT_DESCENT_MIN_TOGETHER) {
#ifdef PT_TRACE_STATS
                st[8]++;
#endif
                break; // the lanes still together start at this node
            }
            const unsigned long long inBest = best == 0u ? inside[0] : (best == 1u ? inside[1] : (best == 2u ? inside[2] : inside[3]));
            const uint32_t refs[4] = { D.x, D.y, D.z, D.w };
            const uint32_t next = best == 0u ? D.x : (best == 1u ? D.y : (best == 2u ? D.z : D.w));
            uint32_t others = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++)
                others += ((uint32_t)k != best && vis[k]) ? 1u : 0u;
            const bool inChild = ((inBest >> lane) & 1ull) != 0ull;
            const bool follow = inChild && sp + others <= (uint32_t)kDescentStack;
#ifdef PT_TRACE_STATS
            st[1]++;
            st[2] += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(follow));
            st[6] += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(inChild && !follow));
            st[7] += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(together && !inChild));
#endif
            together = follow; // (a lane that drops out keeps myCur = this node)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((uint32_t)k != best) { // wave-uniform
                    const bool push = follow && vis[k];
                    s2 = push && sp == 2u ? refs[k] : s2;
                    s1 = push && sp == 1u ? refs[k] : s1;
                    s0 = push && sp == 0u ? refs[k] : s0;
                    sp += push ? 1u : 0u;
                }
            cur = next;
            if (follow)
                myCur = next;
            tmask = __builtin_amdgcn_ballot_w64(together);
            if (refCount(next) == 0u) {
#if PT_DESCENT_PREFETCH
                const uint32_t l0 = best * 16u;
#define RL(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
                A = u4v { RL(pref, l0 + 0u), RL(pref, l0 + 1u), RL(pref, l0 + 2u), RL(pref, l0 + 3u) };
                B = u4v { RL(pref, l0 + 4u), RL(pref, l0 + 5u), RL(pref, l0 + 6u), RL(pref, l0 + 7u) };
                C = u4v { RL(pref, l0 + 8u), RL(pref, l0 + 9u), RL(pref, l0 + 10u), RL(pref, l0 + 11u) };
                D = u4v { RL(pref, l0 + 12u), RL(pref, l0 + 13u), RL(pref, l0 + 14u), RL(pref, l0 + 15u) };
#undef RL
#else
                const uint32_t ni = refIndex(next);
                A = wideS[ni * 4u + 0u], B = wideS[ni * 4u + 1u], C = wideS[ni * 4u + 2u], D = wideS[ni * 4u + 3u];
#endif
            }
        }
        if (valid)
            a.start[idx] = make_uint4(myCur, s0, s1, s2);
#ifdef PT_TRACE_STATS
        st[0]++;
        st[3] += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(valid));
        {
            uint32_t s = valid ? sp : 0u;
            for (int m = 32; m >= 1; m >>= 1)
                s += __shfl_xor(s, m);
            st[4] += s;
        }
        st[5] += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(valid && refCount(myCur) != 0u));
#endif
    }
#ifdef PT_TRACE_STATS
    if (lane == 0)
        for (int i = 0; i < 9; i++)
            atomicAdd(&g_descendStats[i], st[i]);
#endif
}

} // namespace ptd
