// Super-packets: one wave walks the tree ONCE for R x 64 consecutive primary rays -- every lane carries R of them.
//
// With 256 samples of a pixel adjacent in the queue (ptamd.hip), the packets of k_trace_packet (pt_packet.h) that follow each other are
// packets of the SAME pixel: their beams are the same beam (the pixel's footprint) and they visit the same nodes -- the node tests, the
// scalar tournament and the beam set-up were done once per 64 rays for what is one bundle of 256.  Here they are done once per R x 64:
// the camera rays of a pinhole share their origin, so a lane keeps ONE origin and R directions / closest hits, the node loop is
// k_trace_packet's beam walk unchanged (it is wave-uniform: it does not care how many rays stand behind the beam), and a leaf tests its
// triangles against the lane's R rays -- with the origin-dependent half of Moeller-Trumbore (T = o - v0, Q = T x e1, e2 . Q) computed once
// per triangle.  Per 64 rays: 14 node tests -> 14 / R, one beam set-up -> 1 / R; the triangle tests, the ray generation and the
// results stay per ray.
//
// Serves the first closest-hit pass of the fixed schedule where the packet kernel generates the camera rays itself (`fused`), the camera
// is a pinhole and the scene is one world-space tree; super-packets that do not qualify as a whole -- the ragged tail of the queue, rays
// that point into more than one octant -- are walked sub-packet by sub-packet on the per-lane path (every lane its own box tests and
// lane mask, as in k_trace_packet).  Same hits as k_trace_packet and k_trace up to exact-t ties (pt_packet.h).
//
// Reference semantics: traceRay, scene.cl:61-271; slab accept test bvh.cl:72,114; Moeller-Trumbore shapes.cl:20-72.
#pragma once
#include "pt_packet.h"

#ifndef PT_MULTI_RAYS
#define PT_MULTI_RAYS 4
#endif
#ifndef PT_MULTI_MIN_WAVES // per SIMD: a lane's R rays and their hit records live in registers (R = 4: 96 VGPRs)
#define PT_MULTI_MIN_WAVES (PT_MULTI_RAYS >= 8 ? 3 : PT_MULTI_RAYS >= 4 ? 5 : 6)
#endif

namespace ptd {

// (Moeller-Trumbore in two halves, operation by operation: triOriginHalf / triRayHalf, pt_trace.h)

// Do the R x 64 directions point into one octant?  Per axis: the interval of |1 / direction| over the lane's rays (folded over the wave by
// bundleBeam) and the common sign.
template <int R>
__device__ inline bool bundleOctant(const V3 (&d)[R], V3& mLo, V3& mHi, bool& nx, bool& ny, bool& nz)
{
    mLo = mk(INFINITY), mHi = mk(0.f);
    unsigned long long sx = 0ull, sy = 0ull, sz = 0ull, ax = ~0ull, ay = ~0ull, az = ~0ull;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const V3 id = mk(rcpSlab(d[r].x), rcpSlab(d[r].y), rcpSlab(d[r].z));
        mLo = mk(fminf(mLo.x, fabsf(id.x)), fminf(mLo.y, fabsf(id.y)), fminf(mLo.z, fabsf(id.z)));
        mHi = mk(fmaxf(mHi.x, fabsf(id.x)), fmaxf(mHi.y, fabsf(id.y)), fmaxf(mHi.z, fabsf(id.z)));
        const unsigned long long bx = __builtin_amdgcn_ballot_w64(id.x < 0.f), by = __builtin_amdgcn_ballot_w64(id.y < 0.f), bz = __builtin_amdgcn_ballot_w64(id.z < 0.f);
        sx |= bx, sy |= by, sz |= bz, ax &= bx, ay &= by, az &= bz;
    }
    nx = sx != 0ull, ny = sy != 0ull, nz = sz != 0ull;
    // every direction component has one sign across the R x 64 rays: no ray negative (the OR of the ballots is empty) or all of them (the AND is full)
    return (sx == 0ull || ax == ~0ull) && (sy == 0ull || ay == ~0ull) && (sz == 0ull || az == ~0ull);
}
// the constants of the lane's role (axis, entry | exit plane) in the node test, for a bundle with ONE origin (beamSetup, pt_packet.h)
__device__ inline void bundleBeam(const V3 co, V3 mLo, V3 mHi, bool nx, bool ny, bool nz, uint32_t axis, uint32_t isFar, float& S, float& negSO, float& mulPos,
    float& mulNeg, uint32_t& ofsQ)
{
    waveMin3Max3(mLo.x, mLo.y, mLo.z, mHi.x, mHi.y, mHi.z); // the lanes' own intervals (over their R rays) folded over the wave
    const bool neg = axis == 0u ? nx : (axis == 1u ? ny : nz);
    const float oA = axis == 0u ? co.x : (axis == 1u ? co.y : co.z); // one origin: the interval of the origins is a point
    const float mLoA = (axis == 0u ? mLo.x : (axis == 1u ? mLo.y : mLo.z)) * (1.f - 1.f / 262144.f);
    const float mHiA = (axis == 0u ? mHi.x : (axis == 1u ? mHi.y : mHi.z)) * (1.f + 1.f / 262144.f);
    S = neg ? -1.f : 1.f;
    negSO = -(S * oA);
    mulPos = isFar ? -mHiA : mLoA, mulNeg = isFar ? -mLoA : mHiA;
    ofsQ = 16u + 4u * (2u * axis + ((neg ? 1u : 0u) ^ isFar));
}

// LENS (round 6): bundles of a THIN-LENS camera (camera.cl:54-77).  The R x 64 rays leave from all over the lens and meet -- up to the pixels' footprints -- on the
// focal plane, which their un-normalised directions put at t = 1: a converging bundle.  Its beam is k_trace_packet's double cone (pt_packet.h): built around the rays'
// points at t = 1 (the waist) instead of their origins, every distance of the node test measured from there (t' = t - 1).  A lane keeps R origins next to its R
// directions (3 (R - 1) registers more), a leaf runs the whole Moeller-Trumbore test per ray (no origin half to share).
// Per lane: the interval of its rays' points at t = 1, each widened by the round-off of the sum that made it
template <int R>
__device__ inline void bundleWaist(const V3 (&o)[R], const V3 (&d)[R], V3& pLo, V3& pHi)
{
    pLo = mk(INFINITY), pHi = mk(-INFINITY);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const V3 p = mk(o[r].x + d[r].x, o[r].y + d[r].y, o[r].z + d[r].z);
        const V3 e = mk((fabsf(o[r].x) + fabsf(d[r].x)) * 0x1p-22f, (fabsf(o[r].y) + fabsf(d[r].y)) * 0x1p-22f, (fabsf(o[r].z) + fabsf(d[r].z)) * 0x1p-22f);
        pLo = mk(fminf(pLo.x, p.x - e.x), fminf(pLo.y, p.y - e.y), fminf(pLo.z, p.z - e.z));
        pHi = mk(fmaxf(pHi.x, p.x + e.x), fmaxf(pHi.y, p.y + e.y), fmaxf(pHi.z, p.z + e.z));
    }
}
// the constants of the lane's role in the node test for a converging bundle (beamSetup, pt_packet.h, with the lanes' own intervals already folded over their R rays);
// `corner`: the waist corner the role measures from (kept by the caller: a translated + scaled instance moves it)
__device__ inline void bundleBeamLens(V3 pLo, V3 pHi, V3 mLo, V3 mHi, bool nx, bool ny, bool nz, uint32_t axis, uint32_t isFar, float& S, float& negSO, float& mulPos,
    float& mulNeg, uint32_t& ofsQ, float& corner)
{
    waveMin3Max3(pLo.x, pLo.y, pLo.z, pHi.x, pHi.y, pHi.z);
    waveMin3Max3(mLo.x, mLo.y, mLo.z, mHi.x, mHi.y, mHi.z);
    const bool neg = axis == 0u ? nx : (axis == 1u ? ny : nz);
    const float pLoA = axis == 0u ? pLo.x : (axis == 1u ? pLo.y : pLo.z), pHiA = axis == 0u ? pHi.x : (axis == 1u ? pHi.y : pHi.z);
    const float mLoA = (axis == 0u ? mLo.x : (axis == 1u ? mLo.y : mLo.z)) * (1.f - 1.f / 262144.f);
    const float mHiA = (axis == 0u ? mHi.x : (axis == 1u ? mHi.y : mHi.z)) * (1.f + 1.f / 262144.f);
    S = neg ? -1.f : 1.f;
    corner = (neg != (isFar != 0u)) ? pLoA : pHiA;
    negSO = -(S * corner);
    mulPos = isFar ? -mHiA : mLoA, mulNeg = isFar ? -mLoA : mHiA;
    ofsQ = 16u + 4u * (2u * axis + ((neg ? 1u : 0u) ^ isFar));
}

// TWO_LEVEL: the tree holds instance references (k_trace_packet's scheme, pt_packet.h).  Entering an instance is a wave-uniform event: every
// lane takes its origin and its R directions into the instance's space, the beam is rebuilt from them, a sentinel goes onto the stack;
// popping it brings the world-space rays and beam back from LDS (8 + 3 R dwords per lane).  Which instance a ray's closest hit lies in
// is kept in LDS too (written when a hit is accepted -- a few times per ray -- instead of R more registers).  A transform that turns the
// bundle into more than one octant ends the beam walk: the bundle starts over, sub-packet by sub-packet.
template <int R, bool TWO_LEVEL, bool LENS = false>
#ifndef PT_MULTI_MIN_WAVES_TL
#define PT_MULTI_MIN_WAVES_TL PT_MULTI_MIN_WAVES // the instantiation that enters instances.  Round 5 held it at 4 waves per SIMD (at 5 it spilled 16 registers inside the
    // node loop: 25.4 against 20.5 ms per batch).  Round 6's kernel spills 11 outside it: 5 waves are 8 % faster than 4 (camera rays of the crowds 29.2 -> 26.6 and
    // 35.3 -> 32.4 ms per batch; config 4 entered 11 593 -> 11 674 Mrays/s, 432 turned instances 9 179 -> 9 311, 208 translated ones 8 676 -> 8 753: A / B / A / B on one box)
#endif
#ifndef PT_MULTI_MIN_WAVES_LENS
#define PT_MULTI_MIN_WAVES_LENS PT_MULTI_MIN_WAVES // R origins per lane (96 VGPRs + 6 spilled at 5 waves; 113 at 4): config 5 on one GPU 11 348 -> 11 451 Mrays/s at 5 ...
#endif
#ifndef PT_MULTI_MIN_WAVES_LENS_TL
#define PT_MULTI_MIN_WAVES_LENS_TL 4 // ... but lens rays INTO instances (128 VGPRs at 4 waves) lose at 5: 10 759 -> 10 488
#endif
__global__ void __launch_bounds__(kPacketBlock, LENS ? (TWO_LEVEL ? PT_MULTI_MIN_WAVES_LENS_TL : PT_MULTI_MIN_WAVES_LENS) : (TWO_LEVEL ? PT_MULTI_MIN_WAVES_TL : PT_MULTI_MIN_WAVES))
    k_trace_multi(TraceArgs a)
{
    constexpr int RO = LENS ? R : 1; // origins a lane keeps
    constexpr float tShift = LENS ? 1.0f : 0.0f; // the node test's distances are measured from the bundle's waist (t' = t - tShift)
    constexpr int kSaveRay = 0, kSaveBeam = 3 * RO + 3 * R, kSaveInst = kSaveBeam + 5, kSave = kSaveInst + R;
    __shared__ uint32_t ldsSave[TWO_LEVEL ? kPacketBlock / 64 : 1][TWO_LEVEL ? kSave : 1][64];
    const uint32_t pwave = threadIdx.x >> 6;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef const u4v __attribute__((address_space(4)))* ScalarU4; // uniform address + constant space = scalar loads
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = (blockIdx.x * kPacketBlock + threadIdx.x) >> 6;
    const uint32_t totalWaves = (gridDim.x * kPacketBlock) >> 6;
    const uint32_t count = a.ctl->extCount[a.pass];
    const SceneDev& sc = a.sc;
    constexpr uint32_t kPer = 64u * R;
    const uint32_t supers = (count + kPer - 1u) / kPer;
    const uint32_t rootRef = uni(sc.rootRef);
    const ScalarU4 wideS = (ScalarU4)(unsigned long long)sc.wide;
    const ScalarU4 trisS = (ScalarU4)(unsigned long long)sc.tris;

    // claims as in k_trace_packet: the first span of a wave is static, later ones come from the queue cursor (one atomic per span).  Spans of 1 / 2 / 4 / 8 /
    // 16 bundles: 23.7 / 12.3 / 9.8 / 9.1 / 9.5 ms per 2 M-bundle batch -- with short spans the cursor's atomic rate shows (518 k atomics in 9.8 ms are 53 per
    // us on a word that sustains ~ 88).  Dealing half, three quarters, nine tenths or all of the spans round-robin WITHOUT the cursor was measured too: 9.1 /
    // 10.7 / 11.0 / 11.0 ms -- waves that drew cheap spans (sky) cannot take over what the others still hold.
#ifndef PT_MULTI_SPAN
#define PT_MULTI_SPAN 8
#endif
    constexpr uint32_t kSpan = PT_MULTI_SPAN;
    uint32_t spanBase = uni(gwave) * kSpan, spanLeft = kSpan;
    for (;;) {
        if (spanLeft == 0u) {
            const uint32_t left = supers > spanBase ? supers - spanBase : 0u; // spanBase lags the cursor: an upper bound
            const uint32_t claim = min(kSpan, max(1u, left / totalWaves));
            uint32_t b = 0;
            if (lane == 0)
                b = atomicAdd(&a.ctl->extCursor[a.pass], claim);
            spanBase = totalWaves * kSpan + uni(b);
            spanLeft = claim;
        }
        if (spanBase >= supers)
            break;
        const uint32_t base = spanBase * kPer;
        spanBase++;
        spanLeft--;

        // ---- the lane's R camera rays (entries base + 64 r + lane: consecutive lanes write consecutive queue entries) -------------
        V3 co[RO], cd[R];
        float tClosest[R], hu[R], hv[R];
        int hprim[R];
        bool whole = true; // wave-uniform: every ray exists, one origin, one octant
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = base + 64u * (uint32_t)r + lane;
            const bool active = idx < count;
            V3 o = mk(0.f), d = mk(0.f, 0.f, 1.f);
            if (active) {
                uint32_t px, pl;
                primaryEntry(a.fp, a.pixelList, idx, &px, &pl);
                primaryRay(a.fp, px, pl, &o, &d);
                // queued for k_shade (the throughput of a primary ray is 1 and is not stored).  Round 5: nor is the origin where the caller says so -- every
                // ray of a pinhole camera starts at the eye, and (pixel, sample) follow from the entry index: 16 B per ray less each way
                if (a.noOrigins) {
                    ((float4*)a.rayD)[idx] = make_float4(d.x, d.y, d.z, asF(px));
                } else {
                    ((float4*)a.rayO)[idx] = make_float4(o.x, o.y, o.z, asF(px));
                    ((float4*)a.rayD)[idx] = make_float4(d.x, d.y, d.z, asF(packState(FLAG_LASTSPECULAR, 0u, pl)));
                }
            }
            // zero components are nudged as at k_trace's hand-out (NO_PARALLEL_RAYS, scene.cl:123-137)
            if (d.x == 0.0f) d.x = FLT_MIN;
            if (d.y == 0.0f) d.y = FLT_MIN;
            if (d.z == 0.0f) d.z = FLT_MIN;
            if (o.x == 0.0f) o.x = -FLT_MIN;
            if (o.y == 0.0f) o.y = -FLT_MIN;
            if (o.z == 0.0f) o.z = -FLT_MIN;
            if (LENS || r == 0)
                co[LENS ? r : 0] = o;
            cd[r] = d;
            tClosest[r] = INFINITY, hu[r] = hv[r] = 0.f, hprim[r] = -1;
            if constexpr (TWO_LEVEL)
                ldsSave[pwave][kSaveInst + r][lane] = 0xFFFFFFFFu;
            whole = whole && __builtin_amdgcn_ballot_w64(active) == ~0ull;
            if constexpr (!LENS) // one origin for the whole bundle
                whole = whole && __builtin_amdgcn_ballot_w64(o.x != asF(uni(asU(co[0].x))) || o.y != asF(uni(asU(co[0].y))) || o.z != asF(uni(asU(co[0].z)))) == 0ull;
        }
        V3 mLo, mHi;
        bool nx, ny, nz;
        whole = bundleOctant<R>(cd, mLo, mHi, nx, ny, nz) && whole;

        if (whole) {
            // ---- beam walk (pt_packet.h) for the bundle of R x 64 rays ---------------------------------------------------------------
            const uint32_t role = lane & 7u, child = (lane >> 3) & 3u, axis = min(role >> 1, 2u), isFar = role & 1u;
            const uint32_t ofsO = 4u * axis, ofsE = axis == 0u ? 12u : 36u + 4u * axis, shift = 8u * child; // (the lane's origin component / scale: WideNode, pt_device.h)
            float S, negSO, mulPos, mulNeg, corner = 0.f;
            uint32_t ofsQ;
            if constexpr (LENS) {
                V3 pLo, pHi;
                bundleWaist<R>(co, cd, pLo, pHi);
                bundleBeamLens(pLo, pHi, mLo, mHi, nx, ny, nz, axis, isFar, S, negSO, mulPos, mulNeg, ofsQ, corner);
            } else {
                bundleBeam(co[0], mLo, mHi, nx, ny, nz, axis, isFar, S, negSO, mulPos, mulNeg, ofsQ);
            }
            int curInst = -1; // wave-uniform
            if constexpr (TWO_LEVEL) { // the world-space rays and beam, for the way back out of an instance
#pragma unroll
                for (int r = 0; r < RO; r++)
                    ldsSave[pwave][kSaveRay + 3 * r][lane] = asU(co[r].x), ldsSave[pwave][kSaveRay + 3 * r + 1][lane] = asU(co[r].y), ldsSave[pwave][kSaveRay + 3 * r + 2][lane] = asU(co[r].z);
#pragma unroll
                for (int r = 0; r < R; r++)
                    ldsSave[pwave][kSaveRay + 3 * RO + 3 * r][lane] = asU(cd[r].x), ldsSave[pwave][kSaveRay + 3 * RO + 1 + 3 * r][lane] = asU(cd[r].y),
                                  ldsSave[pwave][kSaveRay + 3 * RO + 2 + 3 * r][lane] = asU(cd[r].z);
                ldsSave[pwave][kSaveBeam + 0][lane] = asU(S), ldsSave[pwave][kSaveBeam + 1][lane] = asU(negSO), ldsSave[pwave][kSaveBeam + 2][lane] = asU(mulPos);
                ldsSave[pwave][kSaveBeam + 3][lane] = asU(mulNeg), ldsSave[pwave][kSaveBeam + 4][lane] = ofsQ;
            }
            float tcMax = INFINITY; // wave-uniform: the farthest closest hit of the bundle
            uint32_t stRef = 0u; // the stack: entry e is lane e
            uint32_t sp = 0u;
            uint32_t cur = rootRef;
#ifdef PT_TRACE_STATS
            uint32_t stNodes = 0u, stLeaves = 0u, stTris = 0u;
#endif
            while (true) {
#ifdef PT_TRACE_STATS
                if (refCount(cur) == 0u)
                    stNodes++;
                else if (refCount(cur) != kRefSpecial)
                    stLeaves++, stTris += refCount(cur);
#endif
                if (TWO_LEVEL && refCount(cur) == kRefSpecial) { // wave-uniform
                    if (cur != kRefLeaveInstance) {
                        // -------- enter instance refIndex(cur): instances are only ever entered from world space -------------------------
                        typedef const u4v_t __attribute__((address_space(4)))* ScalarU4i;
                        const ScalarU4i mrow = (ScalarU4i)(unsigned long long)&sc.instances[refIndex(cur)];
                        const u4v_t m0 = mrow[0], m1 = mrow[1], m2 = mrow[2];
                        const float4 r0 = make_float4(asF(m0.x), asF(m0.y), asF(m0.z), asF(m0.w)), r1 = make_float4(asF(m1.x), asF(m1.y), asF(m1.z), asF(m1.w)),
                                     r2 = make_float4(asF(m2.x), asF(m2.y), asF(m2.z), asF(m2.w));
                        if (uni(mrow[3].w) != 0u) {
                            // a translation + uniform scale (Instance::simple; the reference's own scenes, BASELINE configs 4 / 5): x' = x / s + w maps the
                            // beam onto itself -- same octant, the origin moved, every |1 / direction| times s -- so nothing is reduced over the wave
                            // again: one FMA per origin component, one product per direction component (rayIntoInstance's results for such a matrix, bit
                            // for bit, minus its zero-component nudges, as in the per-ray kernels' folded route), two products for the role's multipliers
                            // (1 ulp each against the 2^-18 of slack they carry)
#pragma unroll
                            for (int r = 0; r < RO; r++)
                                co[r] = mk(fmaf(r0.x, co[r].x, r0.w), fmaf(r1.y, co[r].y, r1.w), fmaf(r2.z, co[r].z, r2.w));
#pragma unroll
                            for (int r = 0; r < R; r++)
                                cd[r] = mk(r0.x * cd[r].x, r1.y * cd[r].y, r2.z * cd[r].z);
                            const float scl = rcpFast(r0.x);
                            if constexpr (LENS) { // the waist moves with the rays (t is shared between the spaces: it stays at t = 1); a little outwards for the round-off of the map
                                const float wA = axis == 0u ? r0.w : (axis == 1u ? r1.w : r2.w);
                                const float c = fmaf(r0.x, corner, wA);
                                const float e = (fabsf(r0.x * corner) + fabsf(wA)) * 0x1p-22f;
                                negSO = isFar ? -(S * c) + e : -(S * c) - e; // g = S * plane + negSO: entry lanes bound from below (smaller is safe), exit lanes from above
                            } else {
                                negSO = -(S * (axis == 0u ? co[0].x : (axis == 1u ? co[0].y : co[0].z)));
                            }
                            mulPos *= scl, mulNeg *= scl;
                        } else {
                            V3 to[RO], td[R];
#pragma unroll
                            for (int r = 0; r < R; r++)
                                rayIntoInstance(r0, r1, r2, co[LENS ? r : 0], cd[r], &to[LENS ? r : 0], &td[r]); // (a pinhole's origin is the same R times: folded by the compiler)
                            V3 tLo, tHi;
                            bool tnx, tny, tnz;
                            if (!bundleOctant<R>(td, tLo, tHi, tnx, tny, tnz)) {
                                whole = false; // the bundle no longer points into one octant: start over, sub-packet by sub-packet
                                break;
                            }
#pragma unroll
                            for (int r = 0; r < RO; r++)
                                co[r] = to[r];
#pragma unroll
                            for (int r = 0; r < R; r++)
                                cd[r] = td[r];
                            if constexpr (LENS) {
                                V3 pLo, pHi;
                                float cornerInst; // (`corner` stays the world-space one: the sentinel brings the world-space beam back)
                                bundleWaist<R>(co, cd, pLo, pHi);
                                bundleBeamLens(pLo, pHi, tLo, tHi, tnx, tny, tnz, axis, isFar, S, negSO, mulPos, mulNeg, ofsQ, cornerInst);
                            } else {
                                bundleBeam(co[0], tLo, tHi, tnx, tny, tnz, axis, isFar, S, negSO, mulPos, mulNeg, ofsQ);
                            }
                        }
                        curInst = (int)refIndex(cur);
                        stRef = laneWrite(stRef, kRefLeaveInstance, uni(sp));
                        sp++;
                        cur = uni(mrow[3].x);
                        continue;
                    }
                    // -------- the sentinel: back to the world-space rays and beam -------------------------------------------------------
#pragma unroll
                    for (int r = 0; r < RO; r++)
                        co[r] = mk(asF(ldsSave[pwave][kSaveRay + 3 * r][lane]), asF(ldsSave[pwave][kSaveRay + 3 * r + 1][lane]), asF(ldsSave[pwave][kSaveRay + 3 * r + 2][lane]));
#pragma unroll
                    for (int r = 0; r < R; r++)
                        cd[r] = mk(asF(ldsSave[pwave][kSaveRay + 3 * RO + 3 * r][lane]), asF(ldsSave[pwave][kSaveRay + 3 * RO + 1 + 3 * r][lane]),
                            asF(ldsSave[pwave][kSaveRay + 3 * RO + 2 + 3 * r][lane]));
                    S = asF(ldsSave[pwave][kSaveBeam + 0][lane]), negSO = asF(ldsSave[pwave][kSaveBeam + 1][lane]), mulPos = asF(ldsSave[pwave][kSaveBeam + 2][lane]);
                    mulNeg = asF(ldsSave[pwave][kSaveBeam + 3][lane]), ofsQ = ldsSave[pwave][kSaveBeam + 4][lane];
                    curInst = -1;
                } else if (refCount(cur) == 0u) {
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
                    // (LENS: exit >= 0 and entry < culling distance in t, i.e. exit' >= -tShift and entry' < tcMax - tShift: pt_packet.h)
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
                    for (uint32_t k = 0; k < n; k++) {
                        const u4v ta = trisS[(first + k) * 3u + 0u], tb = trisS[(first + k) * 3u + 1u];
                        const uint32_t tcx = trisS[(first + k) * 3u + 2u].x;
                        const V3 v0 = mk(asF(ta.x), asF(ta.y), asF(ta.z)), e1 = mk(asF(ta.w), asF(tb.x), asF(tb.y)), e2 = mk(asF(tb.z), asF(tb.w), asF(tcx));
                        // the half of Moeller-Trumbore that only knows the origin: once per triangle (LENS: per ray -- every ray has its own)
                        V3 T, Q;
                        float e2Q;
                        if constexpr (!LENS)
                            triOriginHalf(co[0], v0, e1, e2, &T, &Q, &e2Q);
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            float det, u, v, t;
                            if constexpr (LENS)
                                triOriginHalf(co[r], v0, e1, e2, &T, &Q, &e2Q);
                            triRayHalf(cd[r], e1, e2, T, Q, e2Q, &det, &u, &v, &t);
                            const bool hit = !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f && t < tClosest[r];
                            if (hit) {
                                tClosest[r] = t;
                                hu[r] = u;
                                hv[r] = v;
                                hprim[r] = (int)(first + k);
                                if constexpr (TWO_LEVEL)
                                    ldsSave[pwave][kSaveInst + r][lane] = (uint32_t)curInst;
                                any = true;
                            }
                        }
                    }
                    // the bundle's culling distance shrinks once every ray has a hit
                    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                        float far = tClosest[0];
#pragma unroll
                        for (int r = 1; r < R; r++)
                            far = fmaxf(far, tClosest[r]);
                        if (__builtin_amdgcn_ballot_w64(far == INFINITY) == 0ull)
                            tcMax = LENS ? asF(uni(asU(waveMax(far)))) * (1.0f + 0x1p-20f) - tShift : asF(uni(asU(waveMax(far))));
                    }
                }
                if (sp == 0u)
                    break;
                sp--;
                cur = __builtin_amdgcn_readlane(stRef, sp);
            }
#ifdef PT_TRACE_STATS
            if (lane == 0) { // bundles on the beam walk: [56] bundles [57] nodes [58] leaves [59] triangles [60] of which started over sub-packet by sub-packet
                atomicAdd(&g_traceStats[56], 1ull), atomicAdd(&g_traceStats[57], (unsigned long long)stNodes), atomicAdd(&g_traceStats[58], (unsigned long long)stLeaves);
                atomicAdd(&g_traceStats[59], (unsigned long long)stTris);
                if (!whole)
                    atomicAdd(&g_traceStats[60], 1ull);
            }
#endif
        }
        if (!whole) {
            // ---- not one bundle (ragged tail, several origins or octants): sub-packet by sub-packet, every lane for itself ------------
            // (k_trace_packet's per-lane path: a lane tests the four child boxes for its own ray and takes part only in nodes and
            // leaves whose box it passed -- 64-bit lane masks ride on the stack entries)
#pragma unroll 1
            for (int r = 0; r < R; r++) {
                const uint32_t idx = base + 64u * (uint32_t)r + lane;
                const bool active = idx < count;
                V3 o = mk(0.f), d = mk(0.f, 0.f, 1.f);
                if (active) { // the ray again (its origin was not kept): generated, not re-read -- the same bits either way
                    uint32_t px, pl;
                    primaryEntry(a.fp, a.pixelList, idx, &px, &pl);
                    primaryRay(a.fp, px, pl, &o, &d);
                }
                if (d.x == 0.0f) d.x = FLT_MIN;
                if (d.y == 0.0f) d.y = FLT_MIN;
                if (d.z == 0.0f) d.z = FLT_MIN;
                if (o.x == 0.0f) o.x = -FLT_MIN;
                if (o.y == 0.0f) o.y = -FLT_MIN;
                if (o.z == 0.0f) o.z = -FLT_MIN;
                V3 cid = mk(rcpSlab(d.x), rcpSlab(d.y), rcpSlab(d.z));
                bool nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
                const V3 wo = o, wd = d; // TWO_LEVEL: the world-space ray, for the way back out of an instance
                int curInst = -1; // wave-uniform
                if constexpr (TWO_LEVEL)
                    ldsSave[pwave][kSaveInst + r][lane] = 0xFFFFFFFFu; // (a bundle that left the beam walk starts from nothing)
                float tC = INFINITY, u_ = 0.f, v_ = 0.f;
                int hp = -1;
                uint32_t stRef = 0u, stLo = 0u, stHi = 0u; // the stack: entry e is lane e
                uint32_t sp = 0u;
                uint32_t cur = rootRef;
                unsigned long long curMask = __builtin_amdgcn_ballot_w64(active);
                if (curMask != 0ull)
                    while (true) {
                        const bool here = __builtin_amdgcn_inverse_ballot_w64(curMask);
                        if (TWO_LEVEL && refCount(cur) == kRefSpecial) { // wave-uniform
                            if (cur != kRefLeaveInstance) { // enter: every lane transforms its ray, the lanes of curMask walk the subtree
                                V3 to, td;
                                uint32_t root;
                                packetIntoInstance(sc.instances, refIndex(cur), o, d, &to, &td, &root);
                                o = to, d = td, cid = mk(rcpSlab(td.x), rcpSlab(td.y), rcpSlab(td.z));
                                nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
                                curInst = (int)refIndex(cur);
                                lanePush(stRef, stLo, stHi, kRefLeaveInstance, 0u, 0u, uni(sp));
                                sp++;
                                cur = uni(root);
                                continue; // same lane mask
                            }
                            o = wo, d = wd, cid = mk(rcpSlab(d.x), rcpSlab(d.y), rcpSlab(d.z));
                            nx = cid.x < 0.f, ny = cid.y < 0.f, nz = cid.z < 0.f;
                            curInst = -1;
                        } else if (refCount(cur) == 0u) {
                            const uint32_t ni = refIndex(cur);
                            const u4v A = wideS[ni * 4u + 0u], Cs = wideS[ni * 4u + 2u], D = wideS[ni * 4u + 3u];
                            const uint4* wp = (const uint4*)&sc.wide[ni];
                            const uint4 B = wp[1];
                            const uint2 C = *(const uint2*)&wp[2];
                            const float kx = asF(A.w) * cid.x, ky = asF(Cs.z) * cid.y, kz = asF(Cs.w) * cid.z;
                            const float bx = (asF(A.x) - o.x) * cid.x, by = (asF(A.y) - o.y) * cid.y, bz = (asF(A.z) - o.z) * cid.z;
                            const uint32_t qnx = nx ? B.y : B.x, qfx = nx ? B.x : B.y;
                            const uint32_t qny = ny ? B.w : B.z, qfy = ny ? B.z : B.w;
                            const uint32_t qnz = nz ? C.y : C.x, qfz = nz ? C.x : C.y;
                            const float tLimit = here ? tC : -INFINITY; // a lane that is not in this node sees no child
                            unsigned long long m[4];
                            uint32_t key[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const f2 qx = { (float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu) };
                                const f2 qy = { (float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu) };
                                const f2 qz = { (float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu) };
                                const f2 tx = planePair(qx, kx, bx), ty = planePair(qy, ky, by), tz = planePair(qz, kz, bz);
                                const float tmin = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                                const float tmax = fminf(fminf(tx.y, ty.y), tz.y);
                                const float tlo = fmaxf(tmin, 0.f);
                                const bool vis = tmax >= tlo && tmin < tLimit; // bvh.cl:72,114
                                m[k] = __builtin_amdgcn_ballot_w64(vis);
                                const uint32_t pick = (uint32_t)(__builtin_ffsll((long long)m[k]) - 1) & 63u;
                                key[k] = (__builtin_amdgcn_readlane(asU(vis ? tlo : INFINITY), pick) & ~3u) | (uint32_t)k;
                            }
                            const bool s01 = key[0] < key[1], s23 = key[2] < key[3];
                            const uint32_t k01 = s01 ? key[0] : key[1], k23 = s23 ? key[2] : key[3];
                            const uint32_t r01 = s01 ? D.x : D.y, r23 = s23 ? D.z : D.w;
                            const unsigned long long m01 = s01 ? m[0] : m[1], m23 = s23 ? m[2] : m[3];
                            const bool sl = k01 < k23;
                            const uint32_t best = sl ? k01 : k23;
                            if (best < kKeyNone) {
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
                            const uint32_t first = refIndex(cur), n = refCount(cur);
                            for (uint32_t k = 0; k < n; k++) {
                                const u4v ta = trisS[(first + k) * 3u + 0u], tb = trisS[(first + k) * 3u + 1u];
                                const uint32_t tcx = trisS[(first + k) * 3u + 2u].x;
                                const V3 v0 = mk(asF(ta.x), asF(ta.y), asF(ta.z)), e1 = mk(asF(ta.w), asF(tb.x), asF(tb.y)), e2 = mk(asF(tb.z), asF(tb.w), asF(tcx));
                                V3 T, Q;
                                float e2Q, det, u, v, t;
                                triOriginHalf(o, v0, e1, e2, &T, &Q, &e2Q);
                                triRayHalf(d, e1, e2, T, Q, e2Q, &det, &u, &v, &t);
                                const bool hit = here && !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f && t < tC;
                                if (hit) {
                                    tC = t, u_ = u, v_ = v, hp = (int)(first + k);
                                    if constexpr (TWO_LEVEL)
                                        ldsSave[pwave][kSaveInst + r][lane] = (uint32_t)curInst;
                                }
                            }
                        }
                        if (sp == 0u)
                            break;
                        sp--;
                        cur = __builtin_amdgcn_readlane(stRef, sp);
                        curMask = (unsigned long long)__builtin_amdgcn_readlane(stLo, sp) | ((unsigned long long)__builtin_amdgcn_readlane(stHi, sp) << 32);
                    }
                // into the lane's result slots (static indices: the arrays stay in registers)
#pragma unroll
                for (int q = 0; q < R; q++)
                    if (q == r)
                        tClosest[q] = tC, hu[q] = u_, hv[q] = v_, hprim[q] = hp;
            }
        }
        // ---- results: consecutive lanes write consecutive records (scene.cl:257) -------------------------------------------------
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = base + 64u * (uint32_t)r + lane;
            if (idx < count) {
                int hp = hprim[r], hinst = -1;
                if constexpr (TWO_LEVEL)
                    hinst = (int)ldsSave[pwave][kSaveInst + r][lane];
                if (hp >= 0 && hinst < 0) { // a world-space copy of an instance: back to (original triangle, instance)
                    const float4 tc = sc.tris[hp].c;
                    hp = (int)asU(tc.y);
                    hinst = (int)asU(tc.z);
                }
                a.hit[idx] = make_float4(hp >= 0 ? tClosest[r] : INFINITY, hu[r], hv[r], asF((uint32_t)hp));
                a.inst[idx] = hinst;
            }
        }
    }
}

} // namespace ptd
