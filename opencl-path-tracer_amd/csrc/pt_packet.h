// Packet traversal: one wave walks the tree ONCE for the 64 consecutive queue entries it holds.
//
// k_gen hands consecutive queue entries the samples of one pixel (pt_shade.h), so the 64 primary rays of a wave are
// almost the same ray, and the shadow rays their hits spawn leave from one spot.  k_trace (pt_trace.h) would walk
// them in lock-step anyway but pays for 64 private traversals: a per-lane stack, a per-lane sort of the children, a
// vote per iteration.  Here the traversal state is wave-uniform -- the current node reference lives in an SGPR, the stack is
// LANES of a VGPR (entry e is lane e: v_writelane / v_readlane), the nearest child is picked by a scalar tournament -- and there
// are two ways to test a node:
//   * BEAM (closest hit; full packets whose rays all point into one octant -- the primary rays): the node against the interval
//     bundle that holds all 64 rays, the 24 planes of its four children on 24 lanes, one bound each, axes folded by two DPP steps:
//     ~21 vector instructions per node.  Conservative: a box no ray enters may pass, never the reverse; every ray tests the
//     triangles of every leaf the packet visits, with k_trace's arithmetic.
//   * PER LANE (any-hit packets, ragged or mixed-octant ones): every lane tests the four child boxes for its own ray (~85
//     instructions per node) and only takes part in a node or leaf whose box its ray passed (64-bit lane masks ride on the stack
//     entries): per ray exactly the boxes and triangles it would test in k_trace.
// In both the order of traversal can differ from k_trace's, which matters for exact-t ties only.
// Correct for any 64 rays; fast when they are coherent.  Used for the first pass of the fixed schedule when the
// scene is one world-space tree (every instance copied at upload, ptamd.hip) whose worst-case stack fits 64 entries.
//
// Reference semantics: traceRay, scene.cl:61-271 (closest hit, and hitAny :181-185); slab accept test bvh.cl:72,114;
// Moeller-Trumbore shapes.cl:20-72.
#pragma once
#include "pt_trace.h"

#ifndef PT_PACKET_DYNAMIC
#define PT_PACKET_DYNAMIC 16 // > 0: packets per claim of the queue cursor; 0: static round-robin.  4 / 8 / 16 / 32 / 64 / 128: 27.2 / 25.4 / 25.0 / 25.7 / 26.1 / 28.2 ms of closest-hit traversal per batch (27.3 static)
#endif
#ifndef PT_PACKET_MIN_WAVES
#define PT_PACKET_MIN_WAVES 8
#endif

namespace ptd {

constexpr int kPacketBlock = 256;
constexpr uint32_t kPacketStack = 64; // one lane per entry

__device__ inline uint32_t laneWrite(uint32_t v, uint32_t value, uint32_t laneIndex) // v[laneIndex] = value (both wave-uniform)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(laneIndex));
    return v;
}
__device__ inline uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// one stack entry: lane `laneIndex` of the three stack registers (all operands wave-uniform)
__device__ inline void lanePush(uint32_t& r, uint32_t& lo, uint32_t& hi, uint32_t ref, uint32_t mlo, uint32_t mhi, uint32_t laneIndex)
{
    asm volatile("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                 : "+v"(r), "+v"(lo), "+v"(hi)
                 : "s"(ref), "s"(mlo), "s"(mhi), "s"(laneIndex));
}
typedef uint32_t u4v_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kKeyNone = 0x7F800000u; // +inf: no lane sees the child

#ifndef PT_PACKET_BEAM
#define PT_PACKET_BEAM 1 // 1: packets whose rays all point into one octant test the child boxes against the BEAM (24 lanes, one plane each)
#endif
// min of a0, a1, a2 and max of b0, b1, b2 over the 64 lanes, six DPP steps each (row_shr 1 / 2 / 4 / 8 leave a row's result in its
// lane 15 -- min and max do not mind an element counted twice --, row_bcast 15 / 31 carry it on to lane 63), the six chains
// interleaved so that no step reads a register the previous instruction wrote (a DPP read needs two wait states after a vector
// write).  36 vector instructions per packet; through ds_bpermute butterflies the same cost 36 LDS round trips and 108 instructions.
#define PT_BEAM_STEP(m)                            \
    "v_min_f32_dpp %0, %0, %0 " m "\n\t"           \
    "v_min_f32_dpp %1, %1, %1 " m "\n\t"           \
    "v_min_f32_dpp %2, %2, %2 " m "\n\t"           \
    "v_max_f32_dpp %3, %3, %3 " m "\n\t"           \
    "v_max_f32_dpp %4, %4, %4 " m "\n\t"           \
    "v_max_f32_dpp %5, %5, %5 " m "\n\t"
__device__ inline void waveMin3Max3(float& a0, float& a1, float& a2, float& b0, float& b1, float& b2)
{
    asm volatile("s_nop 1\n\t" PT_BEAM_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") PT_BEAM_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
            PT_BEAM_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") PT_BEAM_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                PT_BEAM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") PT_BEAM_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2));
    a0 = asF(__builtin_amdgcn_readlane(asU(a0), 63)), a1 = asF(__builtin_amdgcn_readlane(asU(a1), 63)), a2 = asF(__builtin_amdgcn_readlane(asU(a2), 63));
    b0 = asF(__builtin_amdgcn_readlane(asU(b0), 63)), b1 = asF(__builtin_amdgcn_readlane(asU(b1), 63)), b2 = asF(__builtin_amdgcn_readlane(asU(b2), 63));
}
#undef PT_BEAM_STEP
// butterfly reduction over the 64 lanes (ds_bpermute); every lane gets the result
__device__ inline float waveMax(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
// max(v[lane], v[lane - 2]) within a row of 16 lanes (lanes 0 / 1 of a row keep v): ONE v_max_f32 with a DPP operand.  Written in
// assembly: through __builtin_amdgcn_update_dpp + fmaxf the compiler emits a move, the DPP move and a canonicalising max besides
// (the s_nop covers the two wait states a DPP read needs after a vector write of the same register)
__device__ inline float maxRowShr2(float v)
{
    float r = v;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
}
__device__ inline float rowShr1(float v) // v[lane - 1]
{
    float r = v;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
}

// TWO_LEVEL: the tree holds instance references.  The current reference is wave-uniform, so entering an instance is a uniform event:
// the twelve matrix coefficients arrive by scalar loads, every lane takes its own ray into the instance's space (scene.cl:116-139), the
// beam is rebuilt from the transformed rays (for a pinhole packet the origins stay one point: one interval reduction instead of two) and a
// sentinel goes onto the stack; popping it brings the world-space ray and beam back from LDS, where they were put when the packet started
// (14 dwords per lane).  A transform that turns the packet's directions into more than one octant (a rotation can, for the few packets that
// look along one of the instance's axes) ends the beam walk: the packet starts over on the per-lane path with the hits it has found.
#ifndef PT_PACKET_MIN_WAVES_TL
#define PT_PACKET_MIN_WAVES_TL 7 // 67 VGPRs: at 8 waves (64) the lane-role address offsets of the beam test spill to scratch -- two extra memory round trips per node
#endif
constexpr int kPacketSave = 14; // world-space state of a lane while its packet is inside an instance
// The beam of 64 rays whose directions point into one octant: per axis the interval of the origins and of |1 / direction|, and from
// them the constants of the lane's role (axis, entry | exit plane) in the node test of k_trace_packet.
__device__ inline void beamSetup(const V3 co, const V3 cid, bool nx, bool ny, bool nz, uint32_t axis, uint32_t isFar, float& S, float& negSO, float& mulPos,
    float& mulNeg, uint32_t& ofsQ)
{
    V3 oLo = co, oHi = co;
    if (__builtin_amdgcn_ballot_w64(co.x != asF(uni(asU(co.x))) || co.y != asF(uni(asU(co.y))) || co.z != asF(uni(asU(co.z)))) != 0ull) { // not a pinhole
        waveMin3Max3(oLo.x, oLo.y, oLo.z, oHi.x, oHi.y, oHi.z);
        // the reference points of a converging bundle (below) are computed, not given: a few ulps of slack around their interval
        oLo = mk(oLo.x - fabsf(oLo.x) * 0x1p-20f, oLo.y - fabsf(oLo.y) * 0x1p-20f, oLo.z - fabsf(oLo.z) * 0x1p-20f);
        oHi = mk(oHi.x + fabsf(oHi.x) * 0x1p-20f, oHi.y + fabsf(oHi.y) * 0x1p-20f, oHi.z + fabsf(oHi.z) * 0x1p-20f);
    }
    V3 mLo = mk(fabsf(cid.x), fabsf(cid.y), fabsf(cid.z)), mHi = mLo;
    waveMin3Max3(mLo.x, mLo.y, mLo.z, mHi.x, mHi.y, mHi.z);
    const bool neg = axis == 0u ? nx : (axis == 1u ? ny : nz);
    const float oLoA = axis == 0u ? oLo.x : (axis == 1u ? oLo.y : oLo.z), oHiA = axis == 0u ? oHi.x : (axis == 1u ? oHi.y : oHi.z);
    const float mLoA = (axis == 0u ? mLo.x : (axis == 1u ? mLo.y : mLo.z)) * (1.f - 1.f / 262144.f);
    const float mHiA = (axis == 0u ? mHi.x : (axis == 1u ? mHi.y : mHi.z)) * (1.f + 1.f / 262144.f);
    // g = +-(plane - origin corner) is the signed distance along the axis in the ray's sense; t = g * |1/d|.
    //   entry plane, lower bound: the corner that makes g smallest, times the small multiplier when g >= 0, the large one otherwise
    //   exit plane, upper bound: the corner that makes g largest, times the large multiplier when g >= 0, the small one otherwise
    // (exit lanes hold -upper, so that one max folds both)
    S = neg ? -1.f : 1.f;
    const float corner = (neg != (isFar != 0u)) ? oLoA : oHiA;
    negSO = -(S * corner);
    mulPos = isFar ? -mHiA : mLoA, mulNeg = isFar ? -mLoA : mHiA;
    ofsQ = 16u + 4u * (2u * axis + ((neg ? 1u : 0u) ^ isFar));
}
// the ray of every lane in the space of instance `what` (wave-uniform): the coefficients arrive by scalar loads
__device__ inline void packetIntoInstance(const Instance* instances, uint32_t what, const V3 co, const V3 cd, V3* to, V3* td, uint32_t* root)
{
    typedef const u4v_t __attribute__((address_space(4)))* ScalarU4i;
    const ScalarU4i m = (ScalarU4i)(unsigned long long)&instances[what];
    const u4v_t r0 = m[0], r1 = m[1], r2 = m[2];
    rayIntoInstance(make_float4(asF(r0.x), asF(r0.y), asF(r0.z), asF(r0.w)), make_float4(asF(r1.x), asF(r1.y), asF(r1.z), asF(r1.w)),
        make_float4(asF(r2.x), asF(r2.y), asF(r2.z), asF(r2.w)), co, cd, to, td);
    *root = m[3].x;
}
template <bool ANY_HIT, bool TWO_LEVEL>
__global__ void __launch_bounds__(kPacketBlock, TWO_LEVEL ? PT_PACKET_MIN_WAVES_TL : PT_PACKET_MIN_WAVES) k_trace_packet(TraceArgs a)
{
    __shared__ uint32_t ldsSave[TWO_LEVEL ? kPacketBlock / 64 : 1][TWO_LEVEL ? kPacketSave : 1][64];
    const uint32_t pwave = threadIdx.x >> 6;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef const u4v __attribute__((address_space(4)))* ScalarU4; // uniform address + constant space = scalar loads
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = (blockIdx.x * kPacketBlock + threadIdx.x) >> 6;
    const uint32_t totalWaves = (gridDim.x * kPacketBlock) >> 6;
    const uint32_t count = ANY_HIT ? a.ctl->shadowCount[a.pass] : a.ctl->extCount[a.pass];
    const SceneDev& sc = a.sc;
    const uint32_t packets = (count + 63u) >> 6;
    const uint32_t rootRef = uni(sc.rootRef);
    const ScalarU4 wideS = (ScalarU4)(unsigned long long)sc.wide;
    const ScalarU4 trisS = (ScalarU4)(unsigned long long)sc.tris;

#if PT_PACKET_DYNAMIC
    // The first PT_PACKET_DYNAMIC packets of a wave are static (wave w: [w * n, w * n + n)), later spans of the same
    // size come from the queue cursor: one atomic per span, well under what a device-scope word sustains (pt_trace.h).
    // Consecutive packets are neighbouring pixels, so a wave keeps finding its nodes in the scalar cache and L2, and
    // no wave is left with a long static tail while others idle.
    constexpr uint32_t kSpan = PT_PACKET_DYNAMIC;
    uint32_t spanBase = uni(gwave) * kSpan, spanLeft = kSpan;
    for (;;) {
        if (spanLeft == 0u) {
            uint32_t claim = kSpan;
#if PT_GUIDED_SPANS
            const uint32_t left = packets > spanBase ? packets - spanBase : 0u; // spanBase lags the cursor: an upper bound
            claim = min(kSpan, max(1u, left / (totalWaves * PT_GUIDED_SPANS)));
#endif
            uint32_t b = 0;
            if (lane == 0)
                b = atomicAdd(ANY_HIT ? &a.ctl->shadowCursor[a.pass] : &a.ctl->extCursor[a.pass], claim);
            spanBase = totalWaves * kSpan + uni(b);
            spanLeft = claim;
        }
        if (spanBase >= packets)
            break;
        const uint32_t p = spanBase;
        spanBase++;
        spanLeft--;
        {
#else
    // packets are dealt round-robin: every wave sees the whole queue, so the load evens out without a shared cursor
    for (uint32_t p = uni(gwave); p < packets; p += totalWaves) {
        {
#endif
        const uint32_t idx = p * 64u + lane;
        bool active = idx < count;
        float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 0, 1);
        if (active) {
            if (!ANY_HIT && a.fused) { // wave-uniform
                uint32_t px, pl;
                primaryEntry(a.fp, a.pixelList, idx, &px, &pl);
                V3 o, d;
                primaryRay(a.fp, px, pl, &o, &d);
                ro = make_float4(o.x, o.y, o.z, asF(px));
                rd = make_float4(d.x, d.y, d.z, asF(packState(FLAG_LASTSPECULAR, 0u, pl)));
                ((float4*)a.rayO)[idx] = ro; // queued for k_shade (the throughput of a primary ray is 1 and is not stored)
                ((float4*)a.rayD)[idx] = rd;
            } else {
                ro = a.rayO[idx];
                rd = a.rayD[idx];
            }
        }
        float tClosest = ANY_HIT ? ro.w : INFINITY;
        if (!ANY_HIT && active && (asU(rd.w) & FLAG_FINISHED)) {
            a.hit[idx] = make_float4(INFINITY, 0.f, 0.f, asF(0xFFFFFFFFu));
            a.inst[idx] = -1;
            active = false;
        }
        // zero components are nudged as at k_trace's hand-out (NO_PARALLEL_RAYS, scene.cl:123-137)
        if (rd.x == 0.0f) rd.x = FLT_MIN;
        if (rd.y == 0.0f) rd.y = FLT_MIN;
        if (rd.z == 0.0f) rd.z = FLT_MIN;
        if (ro.x == 0.0f) ro.x = -FLT_MIN;
        if (ro.y == 0.0f) ro.y = -FLT_MIN;
        if (ro.z == 0.0f) ro.z = -FLT_MIN;
        V3 co = xyz(ro), cd = xyz(rd);
        V3 cid = mk(rcpSlab(cd.x), rcpSlab(cd.y), rcpSlab(cd.z));
        bool nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
        float hu = 0.f, hv = 0.f;
        int hprim = -1, hinst = -1;
        int curInst = -1; // wave-uniform
        if constexpr (TWO_LEVEL) { // the world-space ray, for the way back out of an instance
            ldsSave[pwave][0][lane] = asU(co.x), ldsSave[pwave][1][lane] = asU(co.y), ldsSave[pwave][2][lane] = asU(co.z);
            ldsSave[pwave][3][lane] = asU(cd.x), ldsSave[pwave][4][lane] = asU(cd.y), ldsSave[pwave][5][lane] = asU(cd.z);
            ldsSave[pwave][6][lane] = asU(cid.x), ldsSave[pwave][7][lane] = asU(cid.y), ldsSave[pwave][8][lane] = asU(cid.z);
        }

        // ---- beam traversal (closest hit, whole packets of rays pointing into one octant) --------------------------------------
        // The 64 primary rays of a packet are the samples of one pixel: almost one ray.  Testing four child boxes for each of them is
        // 64 x the same answer at ~85 vector instructions per node.  Here the node is tested ONCE against the beam that holds all 64
        // rays -- origin interval x reciprocal-direction interval per axis -- with the 24 planes of its four children spread over 24
        // lanes: lane (child, axis, near | far) computes one bound, t_near >= lower and t_far <= upper for every ray of the packet, two
        // DPP steps fold the three axes, one compare per child decides (lower bound of the entry <= upper bound of the exit, exit >= 0,
        // entry < the farthest closest hit in the packet): ~25 vector instructions per node.  Conservative (a box no ray enters
        // may pass -- its triangles are then tested and missed), never the other way: intervals only widen, and the multipliers carry a
        // relative slack of 2^-18 against round-off.  Triangle tests are per ray, unchanged.  Packets that are not full, contain
        // finished rays or straddle a sign change of a direction component take the per-ray path below.
#ifdef PT_TRACE_STATS
        uint32_t statNodes = 0u, statLeaves = 0u; // wave-uniform: per-lane path of this packet
#endif
        bool viaBeam = false;
#if PT_PACKET_BEAM
        if (!ANY_HIT) {
            const unsigned long long all = ~0ull;
            const unsigned long long sx = __builtin_amdgcn_ballot_w64(nx), sy = __builtin_amdgcn_ballot_w64(ny), sz = __builtin_amdgcn_ballot_w64(nz);
            viaBeam = __builtin_amdgcn_ballot_w64(active) == all && (sx == 0ull || sx == all) && (sy == 0ull || sy == all) && (sz == 0ull || sz == all);
        }
        if (!ANY_HIT && viaBeam) {
            // lane roles (lanes 32-63 repeat 0-31 and are never read): group of 8 lanes per child, (axis, near | far) inside it
            const uint32_t role = lane & 7u, child = (lane >> 3) & 3u, axis = min(role >> 1, 2u), isFar = role & 1u;
            const uint32_t ofsO = 4u * axis, ofsE = axis == 0u ? 12u : 36u + 4u * axis, shift = 8u * child; // (the lane's origin component / scale: WideNode, pt_device.h)
            float S, negSO, mulPos, mulNeg;
            uint32_t ofsQ;
            // The beam of the rays as they are now (signs uniform): per axis the interval of a REFERENCE POINT of every ray and of |1 / direction|.
            // The reference point of a pinhole ray is its origin.  A thin-lens packet is a CONVERGING bundle: the samples of a pixel leave from all over the lens and meet (up to the pixel's
            // footprint) on the focal plane, which their un-normalised directions (camera.cl:71-75) put at t = 1.  Bounding such a bundle
            // by its origins -- the lens -- and its directions is as wide as the lens everywhere, also where the picture is in focus.  The
            // beam is therefore built around the rays' points at t = tShift = 1: a double cone whose waist is the pixel's footprint on the
            // focal plane, and every distance in the node test below is measured from there (t' = t - tShift; boxes in front of the waist
            // have negative t', which the sign-aware bounds were built for).  Pinhole packets: tShift = 0, the origins themselves.
            const float tShift = (a.fused && a.fp.cam.thinLens) ? 1.0f : 0.0f; // wave-uniform
            beamSetup(mk(fmaf(tShift, cd.x, co.x), fmaf(tShift, cd.y, co.y), fmaf(tShift, cd.z, co.z)), cid, nx, ny, nz, axis, isFar, S, negSO, mulPos, mulNeg, ofsQ);
            if constexpr (TWO_LEVEL) { // the world-space beam, next to the world-space ray
                ldsSave[pwave][9][lane] = asU(S), ldsSave[pwave][10][lane] = asU(negSO), ldsSave[pwave][11][lane] = asU(mulPos);
                ldsSave[pwave][12][lane] = asU(mulNeg), ldsSave[pwave][13][lane] = ofsQ;
            }
            float tcMax = INFINITY; // wave-uniform: the farthest closest hit of the packet, minus tShift (the node test's distances are t' = t - tShift)
            uint32_t stRef = 0u; // the stack: entry e is lane e
            uint32_t sp = 0u;
            uint32_t cur = rootRef;
#ifdef PT_TRACE_STATS
            uint32_t stBeamNodes = 0u, stBeamLeaves = 0u, stEnters = 0u, stEmptyVisits = 0u, stNodesAtEnter = 0u, stLeavesAtEnter = 0u;
            unsigned long long stEnterCycles = 0ull;
#endif
            while (true) {
                if (TWO_LEVEL && refCount(cur) == kRefSpecial) { // wave-uniform
                    if (cur != kRefLeaveInstance) {
#ifdef PT_TRACE_STATS
                        stEnters++;
                        stNodesAtEnter = stBeamNodes, stLeavesAtEnter = stBeamLeaves;
                        const unsigned long long tEnter = __builtin_readcyclecounter();
#endif
                        // -------- enter instance refIndex(cur): instances are only ever entered from world space ---------------
                        V3 to, td;
                        uint32_t root;
                        packetIntoInstance(sc.instances, refIndex(cur), co, cd, &to, &td, &root);
                        const V3 tid = mk(rcpSlab(td.x), rcpSlab(td.y), rcpSlab(td.z));
                        const unsigned long long all = ~0ull;
                        const unsigned long long sx = __builtin_amdgcn_ballot_w64(tid.x < 0.f), sy = __builtin_amdgcn_ballot_w64(tid.y < 0.f), sz = __builtin_amdgcn_ballot_w64(tid.z < 0.f);
                        if (!((sx == 0ull || sx == all) && (sy == 0ull || sy == all) && (sz == 0ull || sz == all))) {
                            viaBeam = false; // the packet no longer points into one octant: start over, per lane (the rays are still the world-space ones)
                            break;
                        }
                        co = to, cd = td; // (1 / direction is only needed for the beam: it does not stay in registers here)
                        beamSetup(mk(fmaf(tShift, cd.x, co.x), fmaf(tShift, cd.y, co.y), fmaf(tShift, cd.z, co.z)), tid, sx != 0ull, sy != 0ull, sz != 0ull, axis, isFar, S, negSO,
                            mulPos, mulNeg, ofsQ);
                        curInst = (int)refIndex(cur);
                        stRef = laneWrite(stRef, kRefLeaveInstance, uni(sp));
                        sp++;
                        cur = uni(root);
#ifdef PT_TRACE_STATS
                        stEnterCycles += __builtin_readcyclecounter() - tEnter;
#endif
                        continue;
                    }
#ifdef PT_TRACE_STATS
                    if (stBeamLeaves == stLeavesAtEnter && stBeamNodes <= stNodesAtEnter + 1u)
                        stEmptyVisits++; // the instance's root node was tested and nothing below it
#endif
                    // -------- the sentinel: back to the world-space ray and beam ----------------------------------------------
                    co = mk(asF(ldsSave[pwave][0][lane]), asF(ldsSave[pwave][1][lane]), asF(ldsSave[pwave][2][lane]));
                    cd = mk(asF(ldsSave[pwave][3][lane]), asF(ldsSave[pwave][4][lane]), asF(ldsSave[pwave][5][lane]));
                    S = asF(ldsSave[pwave][9][lane]), negSO = asF(ldsSave[pwave][10][lane]), mulPos = asF(ldsSave[pwave][11][lane]);
                    mulNeg = asF(ldsSave[pwave][12][lane]), ofsQ = ldsSave[pwave][13][lane];
                    curInst = -1;
                } else if (refCount(cur) == 0u) {
#ifdef PT_TRACE_STATS
                    stBeamNodes++;
#endif
                    const uint32_t ni = refIndex(cur);
                    const u4v D = wideS[ni * 4u + 3u]; // child references: scalar
                    const char* nb = (const char*)&sc.wide[ni];
                    const float originA = *(const float*)(nb + ofsO);
                    const float scaleA = *(const float*)(nb + ofsE);
                    const uint32_t qd = *(const uint32_t*)(nb + ofsQ);
                    const float q = (float)((qd >> shift) & 0xFFu);
                    const float g = fmaf(q, S * scaleA, fmaf(S, originA, negSO));
                    float v = g * (g >= 0.f ? mulPos : mulNeg);
                    v = maxRowShr2(v);
                    v = maxRowShr2(v); // lanes 4 / 5 of the group: max over the axes of the entry bounds / of the negated exit bounds
                    const float tn = rowShr1(v); // lane 5: the entry bound from lane 4
                    // entry <= exit, exit >= 0, entry < the packet's culling distance: three lane masks and-ed in scalar registers
                    // (exit >= 0 and entry < culling distance in t, i.e. exit' >= -tShift and entry' < tcMax - tShift)
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(tn + v <= 0.f) & __builtin_amdgcn_ballot_w64(v <= tShift) & __builtin_amdgcn_ballot_w64(tn < tcMax);
                    const uint32_t tloBits = (uint32_t)max((int32_t)asU(tn + tShift), 0); // bits of max(entry, 0): negative floats are negative integers
                    uint32_t key[4];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        key[k] = ((m >> (8 * k + 5)) & 1ull) ? ((__builtin_amdgcn_readlane(tloBits, 8 * k + 5) & ~3u) | (uint32_t)k) : (kKeyNone | (uint32_t)k);
                    const bool s01 = key[0] < key[1], s23 = key[2] < key[3];
                    const uint32_t k01 = s01 ? key[0] : key[1], k23 = s23 ? key[2] : key[3];
                    const uint32_t r01 = s01 ? D.x : D.y, r23 = s23 ? D.z : D.w;
                    const bool sl = k01 < k23;
                    const uint32_t best = sl ? k01 : k23;
                    if (best < kKeyNone) {
                        const uint32_t refs[4] = { D.x, D.y, D.z, D.w };
                        const uint32_t limit = kKeyNone - best - 1u;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (key[k] - best - 1u < limit) {
                                stRef = laneWrite(stRef, uni(refs[k]), uni(sp));
                                sp++;
                            }
                        cur = sl ? r01 : r23;
                        continue;
                    }
                } else {
                    const uint32_t first = refIndex(cur), n = refCount(cur);
                    bool any = false;
#ifdef PT_TRACE_STATS
                    stBeamLeaves++;
#endif
                    // (fetching triangle k + 1 while triangle k is tested was measured: +0.4 ms per batch -- scalar registers are short here)
                    for (uint32_t k = 0; k < n; k++) {
                        const u4v ta = trisS[(first + k) * 3u + 0u], tb = trisS[(first + k) * 3u + 1u];
                        const uint32_t tcx = trisS[(first + k) * 3u + 2u].x;
                        const V3 v0 = mk(asF(ta.x), asF(ta.y), asF(ta.z)), e1 = mk(asF(ta.w), asF(tb.x), asF(tb.y)), e2 = mk(asF(tb.z), asF(tb.w), asF(tcx));
                        float det, u, v, t;
                        triangleTest(co, cd, v0, e1, e2, &det, &u, &v, &t);
                        const bool hit = !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f && t < tClosest;
                        if (hit) {
                            tClosest = t;
                            hu = u;
                            hv = v;
                            hprim = (int)(first + k);
                            hinst = curInst;
                            any = true;
                        }
                    }
                    // the packet's culling distance shrinks once every ray has a hit
                    if (__builtin_amdgcn_ballot_w64(any) != 0ull && __builtin_amdgcn_ballot_w64(tClosest == INFINITY) == 0ull)
                        tcMax = asF(uni(asU(waveMax(tClosest)))) * (1.0f + 0x1p-20f) - tShift;
                }
                if (sp == 0u)
                    break;
                sp--;
                cur = __builtin_amdgcn_readlane(stRef, sp);
            }
#ifdef PT_TRACE_STATS
            if (lane == 0) { // closest-hit beam packets: [48] packets [49] nodes [50] leaves [51] instance entries [52] of which found nothing [53] cycles in entries [54] bails
                atomicAdd(&g_traceStats[48], 1ull), atomicAdd(&g_traceStats[49], (unsigned long long)stBeamNodes), atomicAdd(&g_traceStats[50], (unsigned long long)stBeamLeaves);
                atomicAdd(&g_traceStats[51], (unsigned long long)stEnters), atomicAdd(&g_traceStats[52], (unsigned long long)stEmptyVisits), atomicAdd(&g_traceStats[53], stEnterCycles);
                if (!viaBeam)
                    atomicAdd(&g_traceStats[54], 1ull);
            }
#endif
        }
#endif
        if (ANY_HIT || !viaBeam) {
        if constexpr (TWO_LEVEL && !ANY_HIT) { // (a packet that left the beam walk: its rays are the world-space ones; 1 / direction was not kept)
            cid = mk(rcpSlab(cd.x), rcpSlab(cd.y), rcpSlab(cd.z));
            nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
        }
        uint32_t stRef = 0u, stLo = 0u, stHi = 0u; // the stack: entry e is lane e
        uint32_t sp = 0u; // wave-uniform
        uint32_t cur = rootRef;
        unsigned long long curMask = __builtin_amdgcn_ballot_w64(active);
        if (curMask != 0ull)
            while (true) {
                const bool here = __builtin_amdgcn_inverse_ballot_w64(curMask) && (!ANY_HIT || active);
                if (TWO_LEVEL && refCount(cur) == kRefSpecial) { // wave-uniform
                    if (cur != kRefLeaveInstance) {
                        // -------- enter instance refIndex(cur): every lane transforms its ray, the lanes of curMask walk the subtree
                        V3 to, td;
                        uint32_t root;
                        packetIntoInstance(sc.instances, refIndex(cur), co, cd, &to, &td, &root);
                        co = to, cd = td, cid = mk(rcpSlab(td.x), rcpSlab(td.y), rcpSlab(td.z));
                        nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
                        curInst = (int)refIndex(cur);
                        lanePush(stRef, stLo, stHi, kRefLeaveInstance, 0u, 0u, uni(sp));
                        sp++;
                        cur = uni(root);
                        continue; // same lane mask
                    }
                    // -------- the sentinel: back to the world-space ray ---------------------------------------------------
                    co = mk(asF(ldsSave[pwave][0][lane]), asF(ldsSave[pwave][1][lane]), asF(ldsSave[pwave][2][lane]));
                    cd = mk(asF(ldsSave[pwave][3][lane]), asF(ldsSave[pwave][4][lane]), asF(ldsSave[pwave][5][lane]));
                    cid = mk(asF(ldsSave[pwave][6][lane]), asF(ldsSave[pwave][7][lane]), asF(ldsSave[pwave][8][lane]));
                    nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
                    curInst = -1;
                } else if (refCount(cur) == 0u) {
#ifdef PT_TRACE_STATS
                    statNodes++;
#endif
                    // -------- inner node: four quantised child boxes (scene.cl:197-231 / bvh.cl:76-115) ----------
                    const uint32_t ni = refIndex(cur);
                    const u4v A = wideS[ni * 4u + 0u], Cs = wideS[ni * 4u + 2u], D = wideS[ni * 4u + 3u]; // scalar: origin + scales, child references
                    const uint4* wp = (const uint4*)&sc.wide[ni]; // the plane bytes are selected per lane: vector registers
                    const uint4 B = wp[1];
                    const uint2 C = *(const uint2*)&wp[2];
                    const float ax = asF(A.w) * cid.x, ay = asF(Cs.z) * cid.y, az = asF(Cs.w) * cid.z;
                    const float bx = (asF(A.x) - co.x) * cid.x, by = (asF(A.y) - co.y) * cid.y, bz = (asF(A.z) - co.z) * cid.z;
                    const uint32_t qnx = nx ? B.y : B.x, qfx = nx ? B.x : B.y;
                    const uint32_t qny = ny ? B.w : B.z, qfy = ny ? B.z : B.w;
                    const uint32_t qnz = nz ? C.y : C.x, qfz = nz ? C.x : C.y;
                    const float tLimit = here ? tClosest : -INFINITY; // a lane that is not in this node sees no child
                    unsigned long long m[4];
                    uint32_t key[4]; // entry distance of the first lane that sees the child (float bits, >= 0: ordered as integers), slot in the low bits
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const f2 qx = { (float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu) };
                        const f2 qy = { (float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu) };
                        const f2 qz = { (float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu) };
                        const f2 tx = planePair(qx, ax, bx), ty = planePair(qy, ay, by), tz = planePair(qz, az, bz);
                        const float tmin = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                        const float tmax = fminf(fminf(tx.y, ty.y), tz.y);
                        // tmax >= tmin && tmax >= 0 && tmin < closest (bvh.cl:72,114), the first two folded into one compare
                        const float tlo = fmaxf(tmin, 0.f);
                        const bool vis = tmax >= tlo && tmin < tLimit;
                        m[k] = __builtin_amdgcn_ballot_w64(vis);
                        // s_ff1 of an empty mask is -1 = lane 63, which then holds +inf like every other lane
                        const uint32_t pick = (uint32_t)(__builtin_ffsll((long long)m[k]) - 1) & 63u;
                        key[k] = (__builtin_amdgcn_readlane(asU(vis ? tlo : INFINITY), pick) & ~3u) | (uint32_t)k;
                    }
                    // nearest visible child: a tournament on (key, reference, lane mask), all in scalar registers
                    const bool s01 = key[0] < key[1], s23 = key[2] < key[3];
                    const uint32_t k01 = s01 ? key[0] : key[1], k23 = s23 ? key[2] : key[3];
                    const uint32_t r01 = s01 ? D.x : D.y, r23 = s23 ? D.z : D.w;
                    const unsigned long long m01 = s01 ? m[0] : m[1], m23 = s23 ? m[2] : m[3];
                    const bool sl = k01 < k23;
                    const uint32_t best = sl ? k01 : k23;
                    if (best < kKeyNone) {
                        // the other visible children onto the stack (slot order), the nearest is next.
                        // key > best always; (key - best - 1) < (none - best - 1) <=> key != best && key < none
                        const uint32_t refs[4] = { D.x, D.y, D.z, D.w };
                        const uint32_t limit = kKeyNone - best - 1u;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (key[k] - best - 1u < limit) {
                                lanePush(stRef, stLo, stHi, uni(refs[k]), uni((uint32_t)m[k]), uni((uint32_t)(m[k] >> 32)), uni(sp));
                                sp++;
                            }
                        cur = sl ? r01 : r23;
                        curMask = sl ? m01 : m23;
                        continue;
                    }
                } else {
                    // -------- leaf (scene.cl:168-195) with Moeller-Trumbore (shapes.cl:20-72) -----------------
                    const uint32_t first = refIndex(cur), n = refCount(cur);
#ifdef PT_TRACE_STATS
                    statLeaves++;
#endif
                    for (uint32_t k = 0; k < n; k++) {
                        const u4v ta = trisS[(first + k) * 3u + 0u], tb = trisS[(first + k) * 3u + 1u];
                        const uint32_t tcx = trisS[(first + k) * 3u + 2u].x;
                        const V3 v0 = mk(asF(ta.x), asF(ta.y), asF(ta.z)), e1 = mk(asF(ta.w), asF(tb.x), asF(tb.y)), e2 = mk(asF(tb.z), asF(tb.w), asF(tcx));
                        float det, u, v, t;
                        triangleTest(co, cd, v0, e1, e2, &det, &u, &v, &t);
                        const bool hit = here && !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f
                            && t < tClosest;
                        if (hit) {
                            if (ANY_HIT) {
                                active = false; // occluded: nothing to deposit
                                if (a.occluded)
                                    a.occluded[idx] = 1u;
                            } else {
                                tClosest = t;
                                hu = u;
                                hv = v;
                                hprim = (int)(first + k);
                                hinst = curInst;
                            }
                        }
                    }
                    if (ANY_HIT && __builtin_amdgcn_ballot_w64(active) == 0ull)
                        break; // every ray of the packet is occluded
                }
                if (sp == 0u)
                    break;
                sp--;
                cur = __builtin_amdgcn_readlane(stRef, sp);
                curMask = (unsigned long long)__builtin_amdgcn_readlane(stLo, sp) | ((unsigned long long)__builtin_amdgcn_readlane(stHi, sp) << 32);
            }

        }
#ifdef PT_TRACE_STATS
        if (ANY_HIT && lane == 0) { // histogram of what a shadow packet touches: [0..23] leaves, [24..47] nodes / 4
            atomicAdd(&g_traceStats[min(statLeaves, 23u)], 1ull);
            atomicAdd(&g_traceStats[24u + min(statNodes / 4u, 23u)], 1ull);
        }
#endif
        // -------- results: consecutive lanes write consecutive records (scene.cl:257) --------------------------
        if (ANY_HIT) {
            const uint32_t nDep = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active));
            if (lane == 0 && nDep)
                atomicAdd(&a.ctl->depositsShadow, nDep);
            if (active) {
                if (a.occluded)
                    a.occluded[idx] = 0u;
                const float4 contrib = a.rayC[idx];
                const uint32_t pixel = asU(rd.w);
                float4* ap = a.accum.at(asU(contrib.w) >> 16, pixel); // one live path per entry: plain RMW
                float4 px = *ap;
                px.x += contrib.x, px.y += contrib.y, px.z += contrib.z;
                *ap = px;
            }
        } else if (active) {
            if (hprim >= 0 && hinst < 0) { // a world-space copy of an instance: back to (original triangle, instance)
                const float4 tc = sc.tris[hprim].c;
                hprim = (int)asU(tc.y);
                hinst = (int)asU(tc.z);
            }
            a.hit[idx] = make_float4(hprim >= 0 ? tClosest : INFINITY, hu, hv, asF((uint32_t)hprim));
            a.inst[idx] = hinst;
        }
        }
    }
}

} // namespace ptd
