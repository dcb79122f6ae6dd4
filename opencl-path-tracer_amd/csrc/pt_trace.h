// Persistent-wave two-level BVH traversal for gfx950 (closest-hit and any-hit).
//
// What it computes is the reference's traceRay (assets/cl/scene.cl:61-271): closest (or any) hit of a
// ray with the instanced scene -- instance transform without renormalisation (so t is shared between
// world and object space), the zero-component fix-up, two-sided Moeller-Trumbore with the reference's
// accept tests, `t < closestT` updates in leaf order, any-hit early out.  The ORDER in which
// subtrees are visited is ours (nearer child box first at both levels; the reference orders the top
// level by box-centre distance, scene.cl:141-157): the closest hit does not depend on it, only which
// of two hits at exactly the same t is reported.
//
// How it runs is CDNA4-specific (measurements behind every choice: DESIGN.md section 6):
//  * persistent waves at 7 waves/SIMD (72 VGPRs, 5 KB LDS per wave): the grid is sized to the machine; a wave
//    claims queue entries in spans of up to 512 with ONE atomicAdd (a device-scope word sustains only ~88
//    atomics/us) and copies the 64 entries it will hand out next into LDS ASYNCHRONOUSLY (two global_load_lds_dwordx4:
//    no staging registers, the kernel stays at 72 VGPRs): a hand-out -- consecutive entries for consecutive idle lanes
//    (ballot rank) -- reads LDS instead of stalling on HBM, so it can run as soon as 8 lanes are idle;
//  * ONE traversal stack in LDS, laid out [entry][lane] (a wave's push/pop touches 64 consecutive dwords:
//    conflict-free ds_read/ds_write_b32), 12 entries; top level and bottom level share it, separated by a
//    sentinel entry that restores the world-space ray (re-read from the queue: instances are rarely entered, see
//    below); deeper entries spill to a lane-interleaved region in global memory;
//  * 4-wide nodes with 8-bit quantised child boxes (WideNode, 64 B = four 16-byte loads for four children) at
//    both levels, so one code path -- and one slot in the vote below -- serves both; entry/exit plane bytes are
//    picked per ray-direction sign as whole dwords, planes are one packed FMA per pair
//    (q * (2^e / d) + (origin - o) / d, reciprocal clamped so axis-parallel rays stay NaN-free), the four
//    (distance, reference) pairs go through a 5-comparator network, pushes are branch-free and the stack top is
//    prefetched before the node fetch;
//  * each loop iteration executes the step kind most lanes are waiting for (ballot majority vote: inner step or
//    leaf) instead of serialising both for a fraction of the lanes; the rare kinds -- enter / leave an instance,
//    write the result -- park the lane until the hot loop breaks and are served in batches outside it.  (Most
//    scenes never enter an instance: ptamd.hip copies instances to world space at upload while a byte budget
//    lasts, which removed two parked steps per instance visit.)  Measured in round 3 and not kept: an any-hit lane on the
//    losing side of the vote working on the TOP OF ITS STACK instead when that is of the winning kind (a shadow ray may visit
//    in any order) -- in inner steps / in leaf steps / in both: 43.5 / 41.2 / 42.7 ms of any-hit traversal per batch against
//    40.4: the near-first descent finds occluders sooner than fuller steps save.
#pragma once
#include "pt_shade.h"
#include <type_traits>

namespace ptd {

typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef PT_REFILL_IDLE
#define PT_REFILL_IDLE 8 // hand out new rays once this many lanes are idle; with (PT_PARKED_BREAK, _ANY) = (32, 48): 4 / 8 / 12 / 16 -> 8 810 / 8 865 /
#endif                   // 8 867 / 8 803 Mrays/s; (16, 24, 40) with the rays read from the queue at the hand-out, round 1's setting: 8 727
#ifndef PT_TRACE_MIN_WAVES
#define PT_TRACE_MIN_WAVES 7
#endif
#ifndef PT_GUIDED_SPANS
#define PT_GUIDED_SPANS 1 // > 0: a claim takes at most (entries left) / (waves x this); 0 / 1 / 2 / 4: 8.47 / 8.52 / 8.50 / 8.43 Grays/s
#endif
#ifndef PT_PARKED_BREAK
#define PT_PARKED_BREAK 32 // (this, PT_PARKED_BREAK_ANY) at 8 idle lanes: (16, 24) / (24, 40) / (32, 48) / (40, 56) / (48, 60) -> 8 702 / 8 823 / 8 883 /
#endif                     // 8 860 / 8 733 Mrays/s: every pass over the parked lanes costs the whole wave
#ifndef PT_ANYHIT_SORT
#define PT_ANYHIT_SORT 0
#endif
#ifndef PT_VOTE_INNER
#define PT_VOTE_INNER 2 // an inner step runs when 2 * (lanes wanting one) >= 3 * (lanes wanting a leaf): leaf steps are the
#define PT_VOTE_LEAF 3 // long ones (sequential triangle fetches), so they are not left waiting for a majority
#endif
#ifndef PT_PACKED_FMA
#define PT_PACKED_FMA 0 // 1: the plane distances as 12 v_pk_fma_f32 instead of 24 v_fma_f32 (rounds 2-4; see fmaPlain: 11 040 -> 11 270 Mrays/s without them)
#endif
#ifndef PT_CLOSEST_SORT
#define PT_CLOSEST_SORT 5
#endif
#ifndef PT_OFFSET32
#define PT_OFFSET32 1
#endif
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12 // 16 and 10 measure the same on the benchmark scene; deeper entries spill to global memory
#endif
constexpr int kLdsStack = PT_LDS_STACK; // stack entries kept in LDS per lane
#ifndef PT_LDS_STACK_TL
#define PT_LDS_STACK_TL PT_LDS_STACK // ... in the instantiations that enter instances
#endif
constexpr int kLdsStackTL = PT_LDS_STACK_TL;
constexpr int kTraversalStackMax = (kLdsStackTL < kLdsStack ? kLdsStackTL : kLdsStack) + 100; // what every instantiation can hold (LDS + spill, kSpillStack below)
constexpr uint32_t kInstFoldTable = 96; // entries of the per-workgroup table of folded instance transforms (entry 0: the identity): scenes of up to 95 instances.  (20 B
                                        // each beside 20.5 KB of stacks and staged rays: 7 workgroups per CU must stay within 160 KB -- at 128 entries the any-hit
                                        // instantiation's 23 056 B round up past a seventh of it and the kernel runs at 6 waves per SIMD: 89.1 -> 94.6 ms per batch)
constexpr int kSpillStack = 100; // further entries in global memory
constexpr int kTraceBlock = 256;
#ifndef PT_PARKED_BREAK_ANY
#define PT_PARKED_BREAK_ANY 48
#endif
constexpr int kParkedBreakAny = PT_PARKED_BREAK_ANY; // same, any-hit traversal (only unoccluded rays park: they have a deposit to make)
constexpr int kParkedBreak = PT_PARKED_BREAK; // leave the hot loop once this many lanes are parked on a special step or idle
// the instantiations that enter instances park far more often (every instance entry and exit): they are served earlier.
// Benchmark scene, every instance entered, (closest, any hit) = (12, 16) / (16, 24) / (16, 32) / (24, 40) / (32, 48):
// 8 191 / 8 234 / 8 233 / 8 221 / 8 032 Mrays/s (with the one-pass special section; 7 663 / 7 682 / 7 563 for the last three before it)
#ifndef PT_PARKED_BREAK_TL
#define PT_PARKED_BREAK_TL 16
#define PT_PARKED_BREAK_ANY_TL 24
#endif
constexpr int kRefillIdleLanes = PT_REFILL_IDLE; // hand out new rays once this many lanes are idle
#ifndef PT_REFILL_IDLE_TL
#define PT_REFILL_IDLE_TL 8 // the same in the instantiations that enter instances
#endif

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
    // queue words of this launch, all in the sample's control block (one pointer + the pass index instead of three
    // pointers: the any-hit kernel sits at the scalar-register limit of 7 waves per SIMD):
    //   entries in the queue  ctl->extCount[pass] / shadowCount[pass]; fetch cursor (zero at launch)  ctl->extCursor[pass] /
    //   shadowCursor[pass]; any-hit launches add their unoccluded rays (= accumulator updates) to ctl->depositsShadow
    Control* ctl;
    uint32_t pass;
    // packet kernel, first pass of a batch: the camera rays are generated from the entry index instead of read from the queue
    uint32_t fused;
    uint32_t noOrigins; // fused bundles of a pinhole camera: only the direction (with the pixel in .w) is queued for k_shade, which knows the eye and derives the rest from the entry index
    const uint32_t* pixelList;
    FrameParams fp;
    uint32_t* spill; // kSpillStack * totalThreads dwords
    uint32_t totalThreads;
    uint32_t parityShadow; // any-hit: entries carry a FINISHED flag in rayC.w (reference semantics)
    // k_trace<., true>: the table of folded instance transforms, entry 1 + k = (1/s, w) of instance k (the identity for instances that take the general
    // route), entry 0 = the identity; instFoldCount 0: nothing is folded (more instances than the table holds, parity mode, PT_FLAG_PARKED_INSTANCES)
    // k_trace<., 2> (the general route): the ENTRY records, two float4 per instance -- (1 / s, w) and (root reference, simple flag, s, -); instFoldCount != 0: some
    // entered instance is NOT a translation + uniform scale (its 3 x 4 rows are wanted)
    const float4* instFold;
    uint32_t instFoldCount;
};

#ifdef PT_TRACE_STATS
// diagnostic build only (tools/variants.sh ... -DPT_TRACE_STATS): where do the lanes of a wave go?
// [0] iterations, [1] sum of active lanes, [2..4] iterations per kind, [5..7] lanes served per kind, [8] hand-outs, [9] rays
__device__ unsigned long long g_traceStats[64]; // [0..23] closest-hit launches, [24..47] any-hit launches, [48..63] packet kernel
#define PT_STAT(i, v) statAcc[i] += (unsigned long long)(v)
#define PT_TIC(t) const unsigned long long t = __builtin_readcyclecounter()
#define PT_TOC(i, t) statAcc[i] += __builtin_readcyclecounter() - t
#else
#define PT_STAT(i, v)
#define PT_TIC(t)
#define PT_TOC(i, t)
#endif

__device__ inline float rcpFast(float x) { return __builtin_amdgcn_rcpf(x); } // v_rcp_f32, 1 ulp
// v_fma_f32 that stays a plain v_fma_f32 (the vectoriser would pack two of them into one v_pk_fma_f32: a half-rate instruction that competes with the
// conversions, compares and selects around it, while a plain FP32 multiply-add next to one of those issues at about half its price -- measured,
// profiles/round5/r5r_valu_issue_pairs.md: v_cmp / v_cndmask / v_min3 / v_cvt_f32_ubyte + v_fma_f32 pairs take 2.55 units against 2.0 for the half-rate one alone)
__device__ inline float fmaPlain(float a, float b, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// 3 * x as one shift-add (the compiler turns "x * 48" and "(x + 2 x) << 4" alike into v_mul_lo_u32: a quarter-rate instruction)
__device__ inline uint32_t times3(uint32_t x)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
// the distances to a child's entry and exit planes of one axis: q * a + b for both
__device__ inline f2 planePair(const f2 q, const float a, const float b)
{
#if PT_PACKED_FMA
    const f2 a2 = { a, a }, b2 = { b, b };
    return __builtin_elementwise_fma(q, a2, b2);
#else
    return { fmaPlain(q.x, a, b), fmaPlain(q.y, a, b) };
#endif
}
// Reciprocal direction for the slab test, clamped to +-1e18: a zero (or FLT_MIN, scene.cl:123-137)
// component then yields plane distances of +-1e18 * (b - o) -- far beyond any scene, with the correct
// sign -- instead of the inf - inf = NaN the one-FMA form would produce from an infinite reciprocal.
__device__ inline float rcpSlab(float x) { return fminf(fmaxf(rcpFast(x), -1e18f), 1e18f); }

// Moeller-Trumbore (shapes.cl:20-72), operation by operation.  Rounds 1-4 wrote the test as cross / dot expressions and left the choice of fused
// multiply-adds to the compiler: every kernel then had to happen on the same choice ("the same hits as k_trace, to the bit" is what the tests of the
// packet, bundle and team kernels assert), and a build option that changes the choice in one of them (round 5: -fno-slp-vectorize) moves
// barycentrics in their fifth digit (cancellation in T x e1).  So the sequence those builds emitted is spelled out once, for every kernel:
//   cross(a, b).x = fma(a.y, b.z, -(a.z * b.y))              dot(a, b) = fma(a.z, b.z, fma(a.x, b.x, a.y * b.y))
//   det           = e1.z * P.z + fma(e1.x, P.x, e1.y * P.y)  (the last product rounded on its own)
// The bundle kernel computes the origin half once per triangle and the ray half per ray; the others call triangleTest.
__device__ inline V3 crossExact(const V3 a, const V3 b)
{
#pragma clang fp contract(off)
    return mk(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ inline float dotExact(const V3 a, const V3 b)
{
#pragma clang fp contract(off)
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.x, b.x, a.y * b.y));
}
// the half that only knows the origin
__device__ inline void triOriginHalf(const V3 o, const V3 v0, const V3 e1, const V3 e2, V3* T, V3* Q, float* e2Q)
{
#pragma clang fp contract(off)
    *T = mk(o.x - v0.x, o.y - v0.y, o.z - v0.z);
    *Q = crossExact(*T, e1);
    *e2Q = dotExact(e2, *Q);
}
// the half per ray: det (the caller rejects |det| < FLT_MIN), u, v, t
__device__ inline void triRayHalf(const V3 d, const V3 e1, const V3 e2, const V3 T, const V3 Q, const float e2Q, float* det, float* u, float* v, float* t)
{
#pragma clang fp contract(off)
    const V3 P = crossExact(d, e2);
    const float pz = e1.z * P.z;
    *det = pz + __builtin_fmaf(e1.x, P.x, e1.y * P.y);
    const float inv = rcpFast(*det);
    *u = dotExact(T, P) * inv;
    *v = dotExact(d, Q) * inv;
    *t = e2Q * inv;
}
__device__ inline void triangleTest(const V3 o, const V3 d, const V3 v0, const V3 e1, const V3 e2, float* det, float* u, float* v, float* t)
{
    V3 T, Q;
    float e2Q;
    triOriginHalf(o, v0, e1, e2, &T, &Q, &e2Q);
    triRayHalf(d, e1, e2, T, Q, e2Q, det, u, v, t);
}

// A ray taken into an instance's space (scene.cl:116-139): rows r0..r2 of the inverse transform; the direction is NOT
// renormalised, so t is shared between the two spaces; exactly-zero components are nudged (NO_PARALLEL_RAYS, scene.cl:123-137).
// One spelling (explicit FMAs) for every kernel that enters instances, so that they produce the same bits.
__device__ inline void rayIntoInstance(const float4 r0, const float4 r1, const float4 r2, const V3 o, const V3 d, V3* to, V3* td)
{
    *to = mk(fmaf(r0.x, o.x, fmaf(r0.y, o.y, fmaf(r0.z, o.z, r0.w))), fmaf(r1.x, o.x, fmaf(r1.y, o.y, fmaf(r1.z, o.z, r1.w))),
        fmaf(r2.x, o.x, fmaf(r2.y, o.y, fmaf(r2.z, o.z, r2.w))));
    *td = mk(fmaf(r0.x, d.x, fmaf(r0.y, d.y, r0.z * d.z)), fmaf(r1.x, d.x, fmaf(r1.y, d.y, r1.z * d.z)), fmaf(r2.x, d.x, fmaf(r2.y, d.y, r2.z * d.z)));
    if (td->x == 0.0f) td->x = FLT_MIN;
    if (td->y == 0.0f) td->y = FLT_MIN;
    if (td->z == 0.0f) td->z = FLT_MIN;
    if (to->x == 0.0f) to->x = -FLT_MIN;
    if (to->y == 0.0f) to->y = -FLT_MIN;
    if (to->z == 0.0f) to->z = -FLT_MIN;
}

// TWO_LEVEL: the tree holds instance references (instances that were not copied to world space at upload); scenes that are one
// world-space tree run the instantiation without the instance code.  Entering and leaving an instance are PARKED steps (served in
// batches outside the hot loop, see the kernel).  Measured alternatives (round 3, benchmark scene with every instance entered, ms per
// 256-sample batch of closest-hit bounce rays / shadow rays; parked: 30.4 / 53.1): both inside the hot loop at no extra iteration --
// the ray transformed at the head of the inner step on the instance's root, the world-space ray re-read from the queue at the pop
// of the sentinel -- 81 / 61 (nearly every iteration has SOME lane entering or leaving, so every iteration pays for both, and the
// kernel spills at 72 VGPRs); only the leave inside, the world-space ray of every lane parked in LDS in place of the staged
// packet: 36.1 / 54.4 at 7 waves per SIMD, 32.9 / 56.1 at 6.
#ifndef PT_TRACE_MIN_WAVES_TL
#define PT_TRACE_MIN_WAVES_TL 7
#endif
// LEVELS 2 (round 6): ANY instance transform, any number of instances, nothing parked but the end of a ray (scene.cl:116-139 in full).
//  * Entering an instance is a LEAF-KIND step: the top-level leaf of an instance is its instance reference, a lane that stands on one votes with the lanes
//    that stand on triangle leaves, and in the leaf iteration it fetches the 3 x 4 inverse transform (48 bytes: the size of a triangle record) instead of
//    triangles, takes its ray into the instance's space (rayIntoInstance: the arithmetic of the parked route, bit for bit) and continues at the mesh's
//    root.  Per visit that is one triangle test's worth of instructions shared by the lanes that enter in the same iteration -- no waiting for a quorum
//    of parked lanes, no pass outside the hot loop.
//  * Leaving an instance is NOTHING: the lane keeps the WORLD-space ray in its registers for the whole traversal; the instance-space ray (origin,
//    1 / direction, direction: nine floats) lives in a per-lane LDS slot written at the entry.  Every step reads its ray through an ADDRESS select --
//    object-space references (known from the reference's index alone, as in the folded route) read the lane's slot, world-space references a shared
//    all-zero entry -- and one multiply-add per component, ray = world * m + slot with m = 0 / 1: no vector select per component, no sentinel on
//    the stack, no re-read of the queue, no second pass.  An LDS read costs the vector ALU nothing (the kernel is vector-issue bound).
//  * LDS per wave: the slots (9 x 65 floats) take the place of the staged ray packet, which this instantiation does without (the hand-out reads
//    the queue itself: round 1's way, ~1.5 % slower on the flat scene): 12 stack entries + slots = 5.4 KB, 7 waves per SIMD.
#ifndef PT_TRACE_MIN_WAVES_GEN
#define PT_TRACE_MIN_WAVES_GEN 7
#endif
#ifndef PT_LDS_STACK_GEN
#define PT_LDS_STACK_GEN PT_LDS_STACK
#endif
constexpr int kObjPlanes = 9; // origin xyz, 1 / direction xyz, direction xyz
template <bool ANY_HIT, int LEVELS>
__global__ void __launch_bounds__(kTraceBlock, LEVELS == 2 ? PT_TRACE_MIN_WAVES_GEN : (LEVELS == 1 ? PT_TRACE_MIN_WAVES_TL : PT_TRACE_MIN_WAVES)) k_trace(TraceArgs a)
{
    // (Measured in round 6 and not kept: LEVELS 1 with the table of folded transforms read from GLOBAL memory past 95 entries -- 32 bytes per entry, two more
    // vector loads per object-space step, no entry step, the staged packet kept: 8 431 against 8 361 Mrays/s for LEVELS 2 on 208 translated + uniformly
    // scaled instances: one mechanism instead of two.)
    constexpr bool TWO_LEVEL = LEVELS == 1, GENERAL = LEVELS == 2;
    constexpr bool STAGED = !GENERAL; // the next 64 queue entries of the wave copied into LDS ahead of the hand-out
    constexpr int kLdsStack = GENERAL ? PT_LDS_STACK_GEN : (TWO_LEVEL ? kLdsStackTL : ptd::kLdsStack); // (shadows the namespace constant inside this kernel)
    __shared__ uint32_t ldsStack[kTraceBlock / 64][kLdsStack][64];
    __shared__ float4 ldsRays[STAGED ? kTraceBlock / 64 : 1][2][STAGED ? 64 : 1]; // the claimed packet: origins, directions (entry e of the packet at [.][e])
    __shared__ float ldsObj[GENERAL ? kTraceBlock / 64 : 1][GENERAL ? kObjPlanes : 1][GENERAL ? 65 : 1]; // [.][plane][lane]; [.][plane][64] = 0: what world-space steps read
    // TWO_LEVEL, round 5: instances whose transform is a translation + uniform scale are traversed WITHOUT parking and without an entry step.
    // The lane keeps the WORLD-space ray in its registers; while it walks object-space nodes / triangles -- known from the reference alone:
    // node indices below the first top-level node or in the run of per-instance root copies, triangle indices of the caller's numbering --
    // the ray is taken into the instance's space on the fly,
    //   o' = o * (1/s) + w,   1/d' = (1/d) * s,   d' = d * (1/s)        (t is shared between the spaces, scene.cl:118-121)
    // with (1/s, w) read from a per-workgroup LDS table indexed by the lane's current instance (entry 0 = the identity, read by world-space
    // steps and, through their own identity entries, by lanes inside a general instance, whose registers hold the instance-space ray): no
    // branch, no parked step.  The top-level leaf of such an instance refers to the instance's own copy of its mesh's root node (object space;
    // the copies form the last run of the node array, copy k = instance k): reaching it sets the lane's instance.  (ptamd.hip, convertDynamic.)
    __shared__ float4 ldsInstFold[TWO_LEVEL ? kInstFoldTable : 1];
    __shared__ float ldsInstScale[TWO_LEVEL ? kInstFoldTable : 1]; // s of the same entries (1 / x is a quarter-rate instruction; an LDS read is none of the vector ALU's)
    // unoccluded shadow rays of each wave so far (any-hit): kept in LDS, not in a register -- the kernel sits at the 72
    // VGPRs / ~96 SGPRs that 7 waves per SIMD allow -- and added to the device counter once, when the wave retires
    __shared__ uint32_t ldsDeposits[kTraceBlock / 64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
#define spill (a.spill + gtid) /* entry e at spill[e * totalThreads]; recomputed where used (rare) to save two registers */
    const uint32_t total = a.totalThreads;
    const uint32_t count = ANY_HIT ? a.ctl->shadowCount[a.pass] : a.ctl->extCount[a.pass];
    const SceneDev& sc = a.sc;
    if (ANY_HIT && lane == 0)
        ldsDeposits[wave] = 0u;
    if constexpr (TWO_LEVEL) {
        if (threadIdx.x < kInstFoldTable) {
            const float4 e = threadIdx.x < a.instFoldCount ? a.instFold[threadIdx.x] : make_float4(1.f, 0.f, 0.f, 0.f);
            ldsInstFold[threadIdx.x] = e;
            ldsInstScale[threadIdx.x] = e.x == 1.f ? 1.f : 1.0f / e.x;
        }
        __syncthreads();
    }
    const bool fold = TWO_LEVEL && a.instFoldCount != 0u; // wave-uniform
    if constexpr (GENERAL) {
        if (lane < (uint32_t)kObjPlanes)
            ldsObj[wave][lane][64] = 0.f; // (read by this wave alone, LDS operations of a wave complete in order: no barrier)
    }

    auto push = [&](int slot, uint32_t v) {
        if (slot < kLdsStack)
            ldsStack[wave][slot][lane] = v;
        else
            spill[(size_t)(slot - kLdsStack) * total] = v;
    };
    // always a plain ds_read_b32; the (rare) spilled entry overrides it.  Selecting between the two
    // addresses instead makes hipcc emit ONE flat_load on the critical pop -> node-fetch path.
    auto pop = [&](int slot) -> uint32_t {
        uint32_t v = ldsStack[wave][min(slot, kLdsStack - 1)][lane];
        if (slot >= kLdsStack)
            v = ((const volatile uint32_t*)spill)[(size_t)(slot - kLdsStack) * total]; // volatile: keeps the two loads apart
        return v;
    };

#ifdef PT_TRACE_STATS
    unsigned long long statAcc[24] = {};
    PT_TIC(tKernel);
    // round 5, any-hit launches: does an occluded shadow ray die in the leaf that stopped the previous occluded ray of its lane / of its wave?
    // [19] occluded rays, [20] ... in the leaf of the lane's previous occluded ray, [21] ... in the leaf that last stopped ANY ray of the wave (in an
    // earlier iteration), [22] ... on the very triangle of the lane's previous one, [23] ... whose leaf is one of the wave's last FOUR occluder leaves
    uint32_t statLaneLeaf = 0xFFFFFFFFu, statLaneTri = 0xFFFFFFFFu, statWaveLeaf[4] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
#endif
    bool active = false;
    bool exhausted = false; // wave-uniform: queue has no more rays
    uint32_t rayIdx = 0;
    // ray of the space being traversed (world or instance): origin, direction, 1/direction, -origin/direction
    V3 co = mk(0.f), cd = mk(0.f), cid = mk(0.f);
    float tClosest = 0.f, hu = 0.f, hv = 0.f;
    int hprim = -1, hinst = -1, curInst = -1;
    uint32_t cur = kRefFinish;
    int sp = 0;
    // value of pop() given the prefetched LDS entry `top` (does not move sp)
    auto popTop = [&](uint32_t top) -> uint32_t {
        uint32_t v = sp > 0 ? top : kRefFinish;
        if (sp > kLdsStack)
            v = ((const volatile uint32_t*)spill)[(size_t)(sp - 1 - kLdsStack) * total];
        return v;
    };


    auto setRay = [&](V3 o, V3 d) {
        co = o;
        cd = d;
        cid = mk(rcpSlab(d.x), rcpSlab(d.y), rcpSlab(d.z));
    };

    // ---- per-wave ray packets (see header comment) ---------------------------------------------
    uint32_t poolBase = 0, poolNext = 0, poolEnd = 0; // wave-uniform
    // Claiming queue entries: the first packet of every wave is static (wave w takes entries [64w, 64w+64)),
    // later ones come from ONE shared cursor in spans of up to 512 entries -- a single device-scope word
    // sustains only ~88 atomics/us (MI355X_MICROARCH.md, row `dequeue`), which at 64 rays per atomic would
    // cap the kernel near 5.6 Grays/s and costs ~46 us per launch for the 4096 initial requests alone.
    const uint32_t totalWaves = total >> 6, gwave = gtid >> 6;
    const uint32_t spanSize = 64u * min(8u, max(1u, count / (totalWaves * 64u * 8u)));
    // (Shrinking the static packets of small launches -- the passes of a 1-spp 1280 x 720 frame hold 0.1-0.9 M rays for 0.46 M lanes --
    // so that every wave gets a share of >= 8 rays was measured in round 3: 1.34 instead of 1.37 ms per frame alone, 1.18 instead of
    // 1.03 ms together with the overlapped passes of ptamd.hip: more resident waves slow every wave's iteration down.)
    uint32_t spanNext = gwave * 64u, spanEnd = spanNext + 64u; // wave-uniform: claimed, not yet loaded
    auto requestPacket = [&]() {
        if (spanNext >= spanEnd) {
            uint32_t base = 0xFFFFFFC0u; // "nothing left"
            uint32_t claim = spanSize;
#if PT_GUIDED_SPANS
            // guided self-scheduling: towards the end of the queue claim less, so that no wave is left with a long
            // span while the others have run dry (`spanEnd`, this wave's previous claim, lags the cursor: an upper
            // bound of what is left)
            const uint32_t left = count > spanEnd ? count - spanEnd : 0u;
            claim = min(spanSize, max(64u, (left / (totalWaves * PT_GUIDED_SPANS)) & ~63u));
#endif
            // (small launches -- the passes of a 1-spp frame -- are covered by the static packets: no dynamic part, and no wave asks the
            // cursor just to learn that: 7 168 such atomics were ~80 us of a 100-250 us launch)
            if (gwave * 64u < count && (count + 63u) / 64u > totalWaves) {
                if (lane == 0)
                    base = atomicAdd(ANY_HIT ? &a.ctl->shadowCursor[a.pass] : &a.ctl->extCursor[a.pass], claim);
                base = totalWaves * 64u + __shfl(base, 0);
            }
            spanNext = base;
            spanEnd = base + claim;
        }
        const uint32_t base = spanNext;
        spanNext += 64u;
        poolBase = base;
        poolNext = 0;
        poolEnd = base < count ? min(64u, count - base) : 0u;
        if (STAGED && poolEnd) { // wave-uniform.  Lane l copies entry l (clamped: the tail of the last packet is never handed out)
            const uint32_t e = base + min(lane, poolEnd - 1u);
            // every lane has read its ray of the previous packet (the reads were waited for before the rays were used)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.rayO + e), (__attribute__((address_space(3))) void*)&ldsRays[wave][0][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.rayD + e), (__attribute__((address_space(3))) void*)&ldsRays[wave][1][0], 16, 0, 0);
        }
    };
    requestPacket();

    while (true) {
        // ---- hand rays to idle lanes ----------------------------------------------------------
        PT_TIC(tHand);
        if (!exhausted) {
            const unsigned long long idle = __ballot(!active);
            const int nIdle = __popcll(idle);
            if (nIdle >= (TWO_LEVEL ? PT_REFILL_IDLE_TL : kRefillIdleLanes)) {
                const uint32_t avail = poolEnd - poolNext;
                if (avail == 0u) {
                    exhausted = true; // the request issued after the last hand-out came back empty
                } else {
                    PT_TIC(tShfl);
                    const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    const int e = (int)min(poolNext + rank, 63u);
                    float4 ro, rd;
                    // the packet is only claimed; the lanes that take a ray read it straight from the queue
                    // (consecutive entries for consecutive idle lanes) -- no packet registers, no shuffles
                    ro = rd = make_float4(0, 0, 0, 0);
                    if constexpr (STAGED) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the packet's copy into LDS has landed
                        if (!active && rank < avail) {
                            ro = ldsRays[wave][0][e];
                            rd = ldsRays[wave][1][e];
                        }
                    } else if (!active && rank < avail) {
                        ro = a.rayO[poolBase + (uint32_t)e];
                        rd = a.rayD[poolBase + (uint32_t)e];
                    }
                    PT_TOC(16, tShfl);
                    PT_TIC(tAssign);
                    if (!active && rank < avail) {
                        const uint32_t idx = poolBase + (uint32_t)e;
                        float tMax = ANY_HIT ? ro.w : INFINITY;
                        asm volatile("" : "+v"(tMax)); // own register: hipcc 7.2 otherwise lets the flag load below alias ro.w and drops the tClosest assignment
                        uint32_t state = asU(rd.w); // closest-hit: parity mode keeps finished rays in the queue
                        if (ANY_HIT) // contribution and pixel stay in the queue until the ray turns out unoccluded
                            state = a.parityShadow ? asU(a.rayC[idx].w) : 0u;
                        const bool live = (state & FLAG_FINISHED) == 0u;
                        if (!ANY_HIT && !live) {
                            a.hit[idx] = make_float4(INFINITY, 0.f, 0.f, asF(0xFFFFFFFFu));
                            a.inst[idx] = -1;
                        }
                        if (live) {
                            rayIdx = idx;
                            // The reference nudges exactly-zero components of the ray it takes into an instance
                            // (NO_PARALLEL_RAYS, scene.cl:123-137).  Instances copied to world space are never "entered",
                            // so the same nudge is applied to the world-space ray: identical for the identity and
                            // axis-aligned transforms, and the top-level box tests do not notice 1e-38.
                            if (rd.x == 0.0f) rd.x = FLT_MIN;
                            if (rd.y == 0.0f) rd.y = FLT_MIN;
                            if (rd.z == 0.0f) rd.z = FLT_MIN;
                            if (ro.x == 0.0f) ro.x = -FLT_MIN;
                            if (ro.y == 0.0f) ro.y = -FLT_MIN;
                            if (ro.z == 0.0f) ro.z = -FLT_MIN;
                            setRay(xyz(ro), xyz(rd));
                            tClosest = tMax;
                            hprim = -1;
                            hinst = -1;
                            curInst = -1;
                            hu = hv = 0.f;
                            cur = sc.rootRef;
                            sp = 0;
                            active = true;
                        }
                    }
                    PT_TOC(17, tAssign);
                    PT_STAT(8, 1);
                    PT_STAT(9, min((uint32_t)nIdle, avail));
                    poolNext += min((uint32_t)nIdle, avail);
                    if (poolNext == poolEnd) {
                        PT_TIC(tReq);
                        requestPacket();
                        PT_TOC(15, tReq);
                    }
                }
            }
        }
        PT_TOC(14, tHand);
        PT_TIC(tSpec);
        if constexpr (TWO_LEVEL) {
            // ---- parked lanes of a tree with instances: ONE pass serves them all -------------------------------------------
            // A parked lane has popped its leave-instance sentinel, stands before an instance, or is done.  What it needs follows from
            // what comes NEXT: after the sentinel the next stack entry -- an instance (world-space ray from the queue, then the
            // transform: exit and entry in one step), a top-level node or leaf (the world-space ray), nothing (the result).  Round 2
            // looped here, one kind of step per round: an exit followed by an entry was two rounds.
            const bool wantSpecial = active && refCount(cur) == kRefSpecial;
            const unsigned long long m = __ballot(wantSpecial);
            if (m != 0ull) {
                PT_STAT(4, 1);
                PT_STAT(7, __popcll(m));
                const bool leaving = wantSpecial && cur == kRefLeaveInstance;
                if (leaving) {
                    curInst = -1;
                    cur = sp > 0 ? pop(--sp) : kRefFinish;
                }
                const bool finishing = wantSpecial && cur == kRefFinish;
                if (ANY_HIT) { // a shadow ray only gets here unoccluded (an occluded one retires in its leaf step)
                    const uint32_t nFin = (uint32_t)__popcll(__ballot(finishing));
                    if (lane == 0)
                        ldsDeposits[wave] += nFin;
                }
                if (finishing) {
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
                        if (hprim >= 0 && hinst < 0) { // hit on a world-space copy of an instance: back to (original triangle, instance)
                            const float4 tc = sc.tris[hprim].c;
                            hprim = (int)asU(tc.y);
                            hinst = (int)asU(tc.z);
                        }
                        a.hit[rayIdx] = make_float4(hprim >= 0 ? tClosest : INFINITY, hu, hv, asF((uint32_t)hprim));
                        a.inst[rayIdx] = hinst;
                    }
                    active = false;
                } else if (wantSpecial) {
                    V3 wo = co, wd = cd; // the world-space ray: still in the registers, unless the lane comes out of an instance
                    if (leaving) {
                        const float4 qo = a.rayO[rayIdx], qd = a.rayD[rayIdx];
                        wo = xyz(qo), wd = xyz(qd);
                        if (wd.x == 0.0f) wd.x = FLT_MIN;
                        if (wd.y == 0.0f) wd.y = FLT_MIN;
                        if (wd.z == 0.0f) wd.z = FLT_MIN;
                        if (wo.x == 0.0f) wo.x = -FLT_MIN;
                        if (wo.y == 0.0f) wo.y = -FLT_MIN;
                        if (wo.z == 0.0f) wo.z = -FLT_MIN;
                    }
                    if (refCount(cur) == kRefSpecial) {
                        // -------- enter instance refIndex(cur) (scene.cl:116-139); instances are only ever entered from world space
                        const uint32_t what = refIndex(cur);
                        const Instance in = sc.instances[what];
                        curInst = (int)what; // instance index; pt_intersect reports the top-level leaf
                        if (fold && in.folded) {
                            // a folded instance reached through its instance reference (the start states of the shared descent come from the top level
                            // that holds those): the lane keeps the world-space ray, the steps below take it into the instance's space on the fly
                            setRay(wo, wd);
                        } else {
                            V3 to, td;
                            rayIntoInstance(in.r0, in.r1, in.r2, wo, wd, &to, &td);
                            setRay(to, td);
                            push(sp, kRefLeaveInstance);
                            sp++;
                        }
                        cur = in.rootRef;
                    } else {
                        setRay(wo, wd); // back in world space, at a top-level node or leaf
                    }
                }
            }
        } else {
            // ---- resolve special references (instance entry / leave, end of traversal) -------------------
            // They are kept OUT of the hot loop below: lanes that reach one park until the loop breaks, then all
            // of them are served here at once.  The hot loop thus only ever changes (cur, sp, closest hit) and
            // the ray-space registers stay loop-invariant in it.
            while (true) {
                const bool wantSpecial = active && (GENERAL ? cur == kRefFinish : refCount(cur) == kRefSpecial); // (GENERAL: an instance reference is a leaf-kind step of the hot loop)
                const unsigned long long m = __ballot(wantSpecial);
                if (m == 0ull)
                    break;
                PT_STAT(4, 1);
                PT_STAT(7, __popcll(m));
                if (ANY_HIT) { // a shadow ray only gets here unoccluded (an occluded one retires in its leaf step)
                    const uint32_t nFin = (uint32_t)__popcll(__ballot(wantSpecial && refIndex(cur) == kSpecialFinish));
                    if (lane == 0)
                        ldsDeposits[wave] += nFin;
                }
                if (wantSpecial) {
                    const uint32_t what = refIndex(cur);
                    if (what == kSpecialFinish) {
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
                            if (hprim >= 0 && hinst < 0) { // hit on a world-space copy of an instance: back to (original triangle, instance)
                                const float4 tc = sc.tris[hprim].c;
                                hprim = (int)asU(tc.y);
                                hinst = (int)asU(tc.z);
                            }
                            a.hit[rayIdx] = make_float4(hprim >= 0 ? tClosest : INFINITY, hu, hv, asF((uint32_t)hprim));
                            a.inst[rayIdx] = hinst;
                        }
                        active = false;
                    } else if (TWO_LEVEL) {
                        if (what == kSpecialLeaveInstance) {
                            // -------- back to world space ---------------------------------------------------------
                            {   // the world-space ray again, from the queue (instances are rarely entered: ptamd.hip copies them to world space)
                                float4 wo = a.rayO[rayIdx], wd = a.rayD[rayIdx];
                                if (wd.x == 0.0f) wd.x = FLT_MIN;
                                if (wd.y == 0.0f) wd.y = FLT_MIN;
                                if (wd.z == 0.0f) wd.z = FLT_MIN;
                                if (wo.x == 0.0f) wo.x = -FLT_MIN;
                                if (wo.y == 0.0f) wo.y = -FLT_MIN;
                                if (wo.z == 0.0f) wo.z = -FLT_MIN;
                                setRay(xyz(wo), xyz(wd));
                            }
                            curInst = -1;
                            cur = sp > 0 ? pop(--sp) : kRefFinish;
                        } else {
                            // -------- enter instance `what` (scene.cl:116-139); instances are only ever entered from world space
                            const Instance in = sc.instances[what];
                            V3 to, td;
                            rayIntoInstance(in.r0, in.r1, in.r2, co, cd, &to, &td);
                            setRay(to, td);
                            curInst = (int)what; // instance index; pt_intersect reports the top-level leaf
                            push(sp, kRefLeaveInstance);
                            sp++;
                            cur = in.rootRef;
                        }
                    }
                }
            }
        }
        PT_TOC(13, tSpec);
        if (__ballot(active) == 0ull) {
            if (exhausted)
                break;
            continue;
        }

        // ---- hot loop: inner steps and leaves, until enough lanes are parked (special) or idle -------
        while (true) {
            // the entry a pop would return, fetched before the node / triangle loads so that its LDS latency
            // hides under theirs (whichever step runs this iteration pops at most once, and only when it has
            // pushed nothing)
            const uint32_t stackTop = ldsStack[wave][min(max(sp - 1, 0), kLdsStack - 1)][lane];
            const uint32_t kindBits = refCount(cur);
            const bool wantInner = active && kindBits == 0u;
            const bool wantLeaf = active && kindBits != 0u && (GENERAL ? cur != kRefFinish : kindBits != kRefSpecial);
            const int nInner = __popcll(__ballot(wantInner)), nLeaf = __popcll(__ballot(wantLeaf));
            PT_STAT(0, 1);
            PT_STAT(1, nInner + nLeaf);
            // leave when nothing is left to do here, when enough lanes are parked on a special step, or when
            // enough lanes are idle for a hand-out (and the queue still has rays)
            const int nSpecial = __popcll(__ballot(active && (GENERAL ? cur == kRefFinish : kindBits == kRefSpecial)));
            const int nWork = nInner + nLeaf;
            constexpr int parkedBreak = TWO_LEVEL ? (ANY_HIT ? PT_PARKED_BREAK_ANY_TL : PT_PARKED_BREAK_TL) : (ANY_HIT ? kParkedBreakAny : kParkedBreak);
            if (nWork == 0 || nSpecial >= parkedBreak || (!exhausted && 64 - nWork - nSpecial >= (TWO_LEVEL ? PT_REFILL_IDLE_TL : kRefillIdleLanes)))
                break;
            const bool innerTurn = nInner * PT_VOTE_INNER >= nLeaf * PT_VOTE_LEAF;
            if (innerTurn) {
                PT_STAT(2, 1);
                PT_STAT(5, nInner);
                PT_TIC(tInner);
#ifdef PT_TRACE_STATS
                bool statInside = false;
#endif
                if (wantInner) {
                    // -------- inner step at either level: one 64-byte fetch, FOUR quantised child boxes ----------
                    // TWO_LEVEL: the lane's instance and its table entry, ahead of the node fetch (the LDS read hides under it)
                    float4 foldIs = make_float4(1.f, 0.f, 0.f, 0.f);
                    float foldScale = 1.f;
                    if constexpr (TWO_LEVEL) {
                        const uint32_t ni = refIndex(cur), rk = ni - sc.instRootBase;
                        if (rk < sc.numInstRoots) // an instance's copy of its mesh root: from here on (until the walk is back at world-space references) the lane is inside instance rk
                            curInst = (int)rk;
                        const bool object = ni - sc.firstWorldNode >= sc.instRootBase - sc.firstWorldNode; // (unsigned: below the first world-space node, or behind the last)
                        const uint32_t slot = fold && object ? (uint32_t)curInst + 1u : 0u;
                        foldIs = ldsInstFold[slot];
                        foldScale = ldsInstScale[slot];
                    }
                    float genM = 1.f;
                    V3 genO = mk(0.f), genI = mk(0.f);
                    if constexpr (GENERAL) { // the LDS reads ahead of the node fetch: their latency hides under it
                        const bool object = refIndex(cur) < sc.firstWorldNode; // a node of a mesh tree (the top level and the world-space copies come behind them)
                        const uint32_t slot = object ? lane : 64u;
                        genM = object ? 0.f : 1.f;
                        genO = mk(ldsObj[wave][0][slot], ldsObj[wave][1][slot], ldsObj[wave][2][slot]);
                        genI = mk(ldsObj[wave][3][slot], ldsObj[wave][4][slot], ldsObj[wave][5][slot]);
                    }
#if PT_OFFSET32
                    // base (scalar registers) + 32-bit byte offset: one shift instead of two 64-bit vector operations per step (nodes < 4 GB: checked at upload)
                    const uint4* wp = (const uint4*)((const char*)sc.wide + (size_t)(uint32_t)(refIndex(cur) << 6));
#else
                    const uint4* wp = (const uint4*)&sc.wide[refIndex(cur)];
#endif
                    const uint4 A = wp[0], B = wp[1];
                    const uint4 C = wp[2];
                    const uint4 D = wp[3];
#ifdef PT_EXTRA_LOADS // diagnostic: how sensitive is the kernel to vector-memory instruction count?
                    uint32_t extra = 0;
                    for (int q = 0; q < PT_EXTRA_LOADS; q++) {
                        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                        const u4 E = __builtin_nontemporal_load((const u4*)wp + (q & 3)); // distinct instruction, not CSE'd with A..D
                        asm volatile("" : "+v"(extra));
                        extra ^= E.x;
                    }
                    if (extra == 0x12345u) // never true for real nodes; keeps the loads alive
                        tClosest = 0.f;
#endif
                    // the ray in the node's space (TWO_LEVEL: object-space nodes are seen through the lane's instance, see the top)
                    V3 no = co, nid = cid;
                    if constexpr (TWO_LEVEL) {
                        no = mk(fmaf(co.x, foldIs.x, foldIs.y), fmaf(co.y, foldIs.x, foldIs.z), fmaf(co.z, foldIs.x, foldIs.w));
                        nid = mk(cid.x * foldScale, cid.y * foldScale, cid.z * foldScale);
                    }
                    if constexpr (GENERAL) { // world * m + slot: the lane's instance-space ray at object-space nodes (m = 0), the world-space ray elsewhere (slot = 0)
                        no = mk(fmaf(co.x, genM, genO.x), fmaf(co.y, genM, genO.y), fmaf(co.z, genM, genO.z));
                        nid = mk(fmaf(cid.x, genM, genI.x), fmaf(cid.y, genM, genI.y), fmaf(cid.z, genM, genI.z));
                    }
                    // box plane = origin + scale * q  =>  t = q * (scale / d) + (origin - o) / d : one FMA per plane
                    const float ax = asF(A.w) * nid.x, ay = asF(C.z) * nid.y, az = asF(C.w) * nid.z;
                    // (origin - o) / d from the live registers: keeping -o/d around as well would cost three VGPRs, and 72 is
                    // what 7 waves per SIMD allow
                    const float bx = (asF(A.x) - no.x) * nid.x, by = (asF(A.y) - no.y) * nid.y, bz = (asF(A.z) - no.z) * nid.z;
                    // entry / exit planes chosen by the sign of the ray direction (whole dwords: 4 children at once)
                    // instead of min/max per plane pair; an empty slot is an inverted box (q 255..0) and can never
                    // satisfy exit >= entry -- and if round-off ever made it, its reference is a degenerate triangle
                    const bool nx = nid.x < 0.f, ny = nid.y < 0.f, nz = nid.z < 0.f;
                    const uint32_t qnx = nx ? B.y : B.x, qfx = nx ? B.x : B.y;
                    const uint32_t qny = ny ? B.w : B.z, qfy = ny ? B.z : B.w;
                    const uint32_t qnz = nz ? C.y : C.x, qfz = nz ? C.x : C.y;
                    float key[4];
                    uint32_t ref[4] = { D.x, D.y, D.z, D.w };
#ifdef PT_TRACE_STATS
                    bool inside = false;
#endif
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const f2 qx = { (float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu) };
                        const f2 qy = { (float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu) };
                        const f2 qz = { (float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu) };
                        const f2 tx = planePair(qx, ax, bx), ty = planePair(qy, ay, by), tz = planePair(qz, az, bz);
                        const float tmin = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                        const float tmax = fminf(fminf(tx.y, ty.y), tz.y);
                        // accept test of bvh.cl:72,114 on the (slightly larger) quantised box
                        const bool vis = tmax >= tmin && tmax >= 0.f && tmin < tClosest;
                        key[k] = vis ? tmin : INFINITY;
#ifdef PT_TRACE_STATS
                        inside = inside || (vis && tmin <= 0.f); // the ray starts inside this child's box
#endif
                    }
#ifdef PT_TRACE_STATS
                    // [18] lane-steps whose node has a child that contains the ray's origin ("re-finding the surface the ray starts on")
                    statInside = inside;
#endif
#define PT_VISIBLE(k) (key[k] < INFINITY ? 1 : 0)
                    // sort the four (entry distance, reference) pairs: nearest first (5-comparator network)
#define PT_CSWAP(i, j)                                   \
    {                                                    \
        const bool sw = key[j] < key[i];                 \
        const float tk = sw ? key[j] : key[i];           \
        key[j] = sw ? key[i] : key[j];                   \
        key[i] = tk;                                     \
        const uint32_t tr = sw ? ref[j] : ref[i];        \
        ref[j] = sw ? ref[i] : ref[j];                   \
        ref[i] = tr;                                     \
    }
#if PT_ANYHIT_SORT == 0
                    if (ANY_HIT) { // any occluder will do: only move the nearest visible child to the front
                        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2)
                    } else
#endif
                    {
#if PT_CLOSEST_SORT == 3 // (experiments: the nearest first, the others as they come / nearest first and farthest last)
                        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2)
#elif PT_CLOSEST_SORT == 4
                        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3)
#else
                        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#endif
                    }
#undef PT_CSWAP
                    // farthest first onto the stack, continue with the nearest
                    if (sp + 3 <= kLdsStack) {
                        // common case, branch-free: every candidate is stored, the stack pointer only moves past the
                        // ones that are kept (a rejected one is overwritten by the next store)
                        ldsStack[wave][sp][lane] = ref[3];
                        sp += (int)PT_VISIBLE(3);
                        ldsStack[wave][sp][lane] = ref[2];
                        sp += (int)PT_VISIBLE(2);
                        ldsStack[wave][sp][lane] = ref[1];
                        sp += (int)PT_VISIBLE(1);
                    } else {
                        if (PT_VISIBLE(3)) {
                            push(sp, ref[3]);
                            sp++;
                        }
                        if (PT_VISIBLE(2)) {
                            push(sp, ref[2]);
                            sp++;
                        }
                        if (PT_VISIBLE(1)) {
                            push(sp, ref[1]);
                            sp++;
                        }
                    }
                    // no visible child => nothing was pushed => the prefetched stack top is still the top
                    const uint32_t next = popTop(stackTop);
                    if (PT_VISIBLE(0))
                        cur = ref[0];
                    else
                        cur = next, sp = max(sp - 1, 0);
#undef PT_VISIBLE
                }
#ifdef PT_TRACE_STATS
                PT_STAT(18, __popcll(__ballot(statInside)));
#endif
                PT_TOC(11, tInner);
            } else {
                PT_STAT(3, 1);
                PT_STAT(6, nLeaf);
                PT_TIC(tLeaf);
#ifdef PT_TRACE_STATS
                bool stOcc = false, stLane = false, stWave = false, stTri = false, stWave4 = false;
#endif
                if (GENERAL && wantLeaf && kindBits == kRefSpecial) {
                    // -------- enter instance refIndex(cur) (scene.cl:116-139): a leaf-kind step.  The lane's registers keep the world-space ray; the
                    // instance-space ray goes into the lane's LDS slot, where the steps on the instance's nodes and triangles read it
                    // Two records per instance.  The ENTRY record (32 bytes, a.instFold[2 what]: (1 / s, w), then root reference, "simple" flag, s) serves the common
                    // transform -- a translation + uniform scale -- with the folded route's arithmetic: three FMAs, six products, no reciprocal, no zero-component
                    // fix-ups (the world ray has had its own).  A crowd of such instances runs this block in four leaf iterations of five (208 instances: +19 %
                    // vector instructions per ray against the copied scene with the full transform in it, profiles/round6/r6d_*); the full 3 x 4 transform
                    // (rayIntoInstance on the 64-byte Instance record: the parked route's arithmetic, bit for bit) runs only where a lane needs it.
                    const uint32_t what = refIndex(cur);
                    const float4* ep = (const float4*)((const char*)a.instFold + (size_t)(what << 5));
                    const float4 ef = ep[0], et = ep[1];
                    // (a scene that holds instances of the other kind -- a.instFoldCount, wave-uniform -- fetches their rows in the same round trip: fetched only once
                    // the flag has arrived they cost the general scene a second, dependent one -- 0.999 against 1.008 of the copied scene; both forms behind
                    // branches of their own, the entry record's first half fetched inside: 0.879 against 0.899 on the uniform crowd)
                    const bool rows = a.instFoldCount != 0u;
                    const Instance* ip = (const Instance*)((const char*)sc.instances + (size_t)(what << 6));
                    static_assert(sizeof(Instance) == 64, "an instance record is addressed by index << 6");
                    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
                    if (rows)
                        r0 = ip->r0, r1 = ip->r1, r2 = ip->r2;
                    const uint32_t root = asU(et.x);
                    const bool simple = asU(et.y) != 0u;
                    V3 to = mk(fmaf(co.x, ef.x, ef.y), fmaf(co.y, ef.x, ef.z), fmaf(co.z, ef.x, ef.w));
                    V3 td = mk(cd.x * ef.x, cd.y * ef.x, cd.z * ef.x);
                    V3 tid = mk(cid.x * et.z, cid.y * et.z, cid.z * et.z);
                    if (rows && __ballot(!simple) != 0ull) { // (wave-uniform)
                        if (!simple) {
                            rayIntoInstance(r0, r1, r2, co, cd, &to, &td);
                            tid = mk(rcpSlab(td.x), rcpSlab(td.y), rcpSlab(td.z));
                        }
                    }
                    ldsObj[wave][0][lane] = to.x, ldsObj[wave][1][lane] = to.y, ldsObj[wave][2][lane] = to.z;
                    ldsObj[wave][3][lane] = tid.x, ldsObj[wave][4][lane] = tid.y, ldsObj[wave][5][lane] = tid.z;
                    ldsObj[wave][6][lane] = td.x, ldsObj[wave][7][lane] = td.y, ldsObj[wave][8][lane] = td.z;
                    curInst = (int)what;
                    cur = root; // nothing pushed: what lies below on the stack are world-space references, which read the world-space ray
                } else if (wantLeaf) {
                    // -------- leaf (scene.cl:168-195) with Moeller-Trumbore (shapes.cl:20-72) -----------------
                    const uint32_t first = refIndex(cur), n = kindBits;
                    V3 lo_ = co, ld_ = cd; // the ray in the triangles' space
                    bool inObject = false;
                    if constexpr (GENERAL) {
                        inObject = first < sc.numTriangles + 1u; // the caller's (object-space) triangles; world-space copies come behind them
                        const uint32_t slot = inObject ? lane : 64u;
                        const float m = inObject ? 0.f : 1.f;
                        lo_ = mk(fmaf(co.x, m, ldsObj[wave][0][slot]), fmaf(co.y, m, ldsObj[wave][1][slot]), fmaf(co.z, m, ldsObj[wave][2][slot]));
                        ld_ = mk(fmaf(cd.x, m, ldsObj[wave][6][slot]), fmaf(cd.y, m, ldsObj[wave][7][slot]), fmaf(cd.z, m, ldsObj[wave][8][slot]));
                    }
                    if constexpr (TWO_LEVEL) {
                        inObject = first < sc.numTriangles + 1u; // the caller's (object-space) triangles; world-space copies come behind them
                        const float4 is = ldsInstFold[fold && inObject ? (uint32_t)curInst + 1u : 0u];
                        lo_ = mk(fmaf(co.x, is.x, is.y), fmaf(co.y, is.x, is.z), fmaf(co.z, is.x, is.w)); // rayIntoInstance's arithmetic for such a matrix, bit for bit
                        ld_ = mk(cd.x * is.x, cd.y * is.x, cd.z * is.x);
                    }
                    bool done = false;
#ifdef PT_TRACE_STATS
                    uint32_t statTri = 0xFFFFFFFFu;
#endif
                    for (uint32_t k = 0; k < n; k++) {
#if PT_OFFSET32
                        static_assert(sizeof(TriIsect) == 48, "48 = 3 << 4: a shift-add and a shift instead of a quarter-rate 32-bit multiply");
                        const uint32_t ti = first + k;
                        const TriIsect* tp = (const TriIsect*)((const char*)sc.tris + (size_t)(uint32_t)(times3(ti) << 4));
#else
                        const TriIsect* tp = &sc.tris[first + k];
#endif
                        const float4 ta = tp->a, tb = tp->b;
                        const float tcx = tp->c.x;
                        const V3 v0 = mk(ta.x, ta.y, ta.z), e1 = mk(ta.w, tb.x, tb.y), e2 = mk(tb.z, tb.w, tcx);
                        float det, u, v, t;
                        triangleTest(lo_, ld_, v0, e1, e2, &det, &u, &v, &t);
                        const bool hit = !(det > -FLT_MIN && det < FLT_MIN) && !(u < 0.f || u > 1.f) && !(v < 0.f || u + v > 1.f) && t > 0.f
                            && t < tClosest;
                        if (hit) {
                            if (ANY_HIT) {
                                done = true;
#ifdef PT_TRACE_STATS
                                statTri = first + k;
#endif
                                break;
                            }
                            tClosest = t;
                            hu = u;
                            hv = v;
                            hprim = (int)(first + k);
                            hinst = (TWO_LEVEL || GENERAL) ? (inObject ? curInst : -1) : curInst;
                        }
                    }
#ifdef PT_TRACE_STATS
                    if (ANY_HIT && done) {
                        stOcc = true;
                        stLane = cur == statLaneLeaf;
                        stWave = cur == statWaveLeaf[0];
                        stTri = statTri == statLaneTri;
                        stWave4 = cur == statWaveLeaf[0] || cur == statWaveLeaf[1] || cur == statWaveLeaf[2] || cur == statWaveLeaf[3];
                        statLaneLeaf = cur, statLaneTri = statTri;
                    }
#endif
                    if (ANY_HIT && done) { // occluded: nothing to deposit
                        if (a.occluded)
                            a.occluded[rayIdx] = 1u;
                        active = false;
                        cur = kRefFinish;
                    } else {
                        cur = popTop(stackTop);
                        sp = max(sp - 1, 0);
                    }
                }
#ifdef PT_TRACE_STATS
                if (ANY_HIT) { // the leaves that stopped rays in this step become the wave's most recent occluder leaves (up to four, newest first)
                    PT_STAT(19, __popcll(__ballot(stOcc)));
                    PT_STAT(20, __popcll(__ballot(stLane)));
                    PT_STAT(21, __popcll(__ballot(stWave)));
                    PT_STAT(22, __popcll(__ballot(stTri)));
                    PT_STAT(23, __popcll(__ballot(stWave4)));
                    unsigned long long stopped = __ballot(stOcc);
                    for (int q = 0; q < 4 && stopped; q++) {
                        const int src = __builtin_ctzll(stopped);
                        stopped &= stopped - 1ull;
                        const uint32_t leaf = (uint32_t)__shfl((int)statLaneLeaf, src);
                        if (leaf != statWaveLeaf[0] && leaf != statWaveLeaf[1] && leaf != statWaveLeaf[2] && leaf != statWaveLeaf[3])
                            statWaveLeaf[3] = statWaveLeaf[2], statWaveLeaf[2] = statWaveLeaf[1], statWaveLeaf[1] = statWaveLeaf[0], statWaveLeaf[0] = leaf;
                    }
                }
#endif
                PT_TOC(12, tLeaf);
            }
        }
    }
    if (ANY_HIT && lane == 0 && ldsDeposits[wave])
        atomicAdd(&a.ctl->depositsShadow, ldsDeposits[wave]);
    PT_TOC(10, tKernel);
#ifdef PT_TRACE_STATS
    if (lane == 0)
        for (int i = 0; i < 24; i++)
            atomicAdd(&g_traceStats[i + (ANY_HIT ? 24 : 0)], statAcc[i]);
#endif
}

#undef spill

} // namespace ptd
