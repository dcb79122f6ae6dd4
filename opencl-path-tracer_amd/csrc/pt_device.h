// Device-side data layout of the gfx950 ray-queue path (all resident in HBM; see DESIGN.md "HBM layout").
//
// Queues are structure-of-arrays at float4 granularity: every field group is its own array of
// 16-byte elements, so a wave reads/writes 64 x 16 B = 1 KiB contiguous per instruction (the widest
// coalesced access on CDNA4) and a kernel only touches the groups it needs.  This replaces the
// reference's 80-byte AoS RayData (assets/cl/shading.cl:16-29) and the pointer-carrying 32-byte
// ShadingData (assets/cl/kernel_data.cl:26-33).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

// ---- geometry -------------------------------------------------------------------------------
// One bottom-level BVH *pair node* (64 B): the boxes of BOTH children of a reference inner node
// (SubBvhNode, assets/cl/bvh.cl:5-16) interleaved per axis + the two child references, so one
// traversal step is one 64-byte fetch instead of the reference's two dependent 48-byte fetches
// (scene.cl:199-200).
struct PairNode {
    float4 bx; // lmin.x, lmax.x, rmin.x, rmax.x
    float4 by;
    float4 bz;
    uint32_t left, right; // child references (see makeRef)
    uint32_t _pad0, _pad1;
};

// 4-wide node with 8-bit quantised child boxes (64 B), built at upload by collapsing the pair-node tree: which descendants of a
// pair node become the (up to four) children of its wide node is chosen by dynamic programming over the binary tree so that the
// summed surface area of the wide tree's inner nodes is minimal (ptamd.hip, collapseToWide).
// Child box k = origin + 2^exp * q (per axis), q in [0,255], rounded outwards: a superset of the exact
// box, so traversal visits at worst a few extra nodes and finds exactly the same triangles.  One step
// now needs four 16-byte loads for four children instead of eight for the same two levels.
struct WideNode {
    float ox, oy, oz; // quantisation origin = min corner of the union of the children
    float scaleX; // per-axis scale of the plane bytes: a power of two, as a float (rounds 1-4 packed three exponent bytes into this word: every
                  // traversal step then spent six shift / mask instructions on unpacking them -- the node had eight spare bytes)
    uint32_t qlox, qhix, qloy, qhiy; // byte k of each word belongs to child k
    uint32_t qloz, qhiz;
    float scaleY, scaleZ;
    uint32_t child[4]; // references; an unused slot has an inverted box and refers to an all-zero triangle
};

// child reference: kind/count (5 bits) | index (27 bits).
//   count 0      : inner node, index of a PairNode (top level and bottom level share the array)
//   count 1..30  : leaf, index of its first triangle (larger leaves are split at upload)
//   count 31     : special -- index = instance to enter, or one of the two markers below
constexpr uint32_t kRefIndexBits = 27;
constexpr uint32_t kRefIndexMask = (1u << kRefIndexBits) - 1u;
constexpr uint32_t kMaxLeafTris = 30;
constexpr uint32_t kRefSpecial = 31;
constexpr uint32_t kSpecialFinish = kRefIndexMask; // nothing left to traverse
constexpr uint32_t kSpecialLeaveInstance = kRefIndexMask - 1u; // stack sentinel: restore the world-space ray
constexpr uint32_t kRefFinish = 0xFFFFFFFFu;
constexpr uint32_t kRefLeaveInstance = (kRefSpecial << kRefIndexBits) | kSpecialLeaveInstance;
constexpr uint32_t kRefNone = kRefFinish;
__host__ __device__ inline uint32_t makeRef(uint32_t index, uint32_t count) { return (count << kRefIndexBits) | index; }
__host__ __device__ inline uint32_t refIndex(uint32_t r) { return r & kRefIndexMask; }
__host__ __device__ inline uint32_t refCount(uint32_t r) { return r >> kRefIndexBits; }

// Triangle for intersection (48 B): v0 and the two edges the reference recomputes per test
// (shapes.cl:37-38) -- 36 useful bytes instead of a 16 B index record + 3 x 48 B vertices.
struct TriIsect {
    float4 a; // v0.xyz, e1.x
    float4 b; // e1.yz, e2.xy
    float4 c; // e2.z, -, -, -
};

// Host-side intermediates of the shading data (caller's indexed layout: not uploaded)
struct TriShade {
    uint32_t i0, i1, i2, material;
};
struct VertexShade { // 32 B
    float4 n_u; // normal.xyz, texCoord.x
    float4 v_pad; // texCoord.y
};
// Triangle for shading: everything k_shade needs of a hit triangle -- its material included -- in ONE 128-byte line (caller's triangle numbering, object space).
// The indexed layout it replaces cost an incoherent hit five scattered lines and a dependent fetch (16 B of vertex indices + material,
// three 32-byte vertex records through them, the 48-byte intersection record for the geometric normal): from the second bounce on
// k_shade fetched more scene data than queue entries (20 GB against 14 GB for 4.4 x fewer entries, FETCH_SIZE per dispatch).
struct TriFat {
    float4 n0u, n1u, n2u; // vertex normal.xyz, texCoord.x of the three vertices
    float4 vvvm; // texCoord.y of the three vertices, bits(material)
    float4 e1e; // edge1.xyz, edge2.x          (v0 / edge1 / edge2 exactly as in TriIsect)
    float4 e2v; // edge2.yz, v0.xy
    float4 v0c; // v0.z, material colour.xyz                  } the triangle's material record (48 B in the caller's array, 28 B of it
    float4 mat; // material params.xyz, bits(material type)   } used): no second, dependent fetch through the material index
};

// instance (64 B): rows of the 3x4 inverse world transform + root reference of the mesh BVH
struct Instance {
    float4 r0, r1, r2; // row i = (m[i], m[4+i], m[8+i], m[12+i]) of the column-major matrix
    uint32_t rootRef; // makeRef
    uint32_t topNode; // index of the top-level leaf (reported as `inst` in hit records)
    uint32_t folded; // the per-ray kernels walk this instance without parking (translation + uniform scale: pt_trace.h); its table entry is not the identity
    uint32_t simple; // the transform is a translation + uniform scale: a bundle of camera rays enters by scaling its beam (pt_packet_multi.h)
};

struct Material { // the reference's 48-byte record, read as 3 x float4
    float4 colour; // diffuse / base|reflectance / absorption / emissive
    float4 params; // bits: x = textureId | smoothness | iorBasic ; y = f0NonMetal | iorRough ; z = metallic byte
    float4 typeAndPad; // bits x = type
};

struct Light { // EmissiveTriangle (light.cl:5-9) pre-digested: 80 B
    float4 v0, v1, v2; // world space; v0.w = area (Heron), v1.w.. unused
    float4 normal; // normalize(cross(v1-v0, v2-v0))
    float4 colour;
};

struct Texture {
    const void* texels; // [layer][y][x]: float4 (format 0) or 4 bytes B, G, R, A as UNORM_INT8 (format 1)
    int32_t width, height, layers;
    int32_t format; // pt_texture_format
};
// one texel as read_imagef returns it: RGBA32F as stored; CL_BGRA / CL_UNORM_INT8 (the reference's material array,
// src/opencl/texture.cpp:148) as byte / 255 with the channels back in r, g, b, a order
__device__ inline float4 fetchTexel(const Texture& tex, size_t index)
{
    if (tex.format == 0)
        return ((const float4*)tex.texels)[index];
    const uint32_t p = ((const uint32_t*)tex.texels)[index];
    return make_float4((float)((p >> 16) & 0xFFu) / 255.0f, (float)((p >> 8) & 0xFFu) / 255.0f, (float)(p & 0xFFu) / 255.0f, (float)(p >> 24) / 255.0f);
}

struct SceneDev {
    const PairNode* nodes; // the reference's binary tree, both boxes per node: host-side intermediate, not uploaded (null)
    const WideNode* wide; // 4-wide tree (k_trace)
    const TriIsect* tris; // caller's triangle numbering, then the world-space copies of baked instances
    const TriFat* triFat; // caller's triangle numbering
    const Material* materials;
    const Instance* instances;
    const Light* lights;
    Texture materialTex;
    Texture sky;
    uint32_t numLights;
    uint32_t rootRef; // reference of the top-level root: a PairNode, or an instance when there is only one
    uint32_t numTriangles;
    uint32_t firstWorldNode; // nodes below this index are object-space nodes of the mesh trees (reached through an instance); the top level and the world-space copies come behind ...
    uint32_t instRootBase; // ... and behind those, as the last run of the array, one object-space copy of its mesh's root node per instance (k_trace<., true>: pt_trace.h)
    uint32_t numInstRoots; // copies in that run (0: no instance is folded)
};

// ---- queues ---------------------------------------------------------------------------------
struct RayQueue { // extension rays + path state, capacity entries each
    float4* o; // origin.xyz, bits(pixel)
    float4* d; // direction.xyz, bits(flags | bounce << 8)
    float4* thr; // throughput rgb
};
struct HitQueue {
    float4* h; // t, u, v, bits(prim)   (t = +inf: miss)
    int32_t* inst; // instance index, -1 on miss
};
struct ShadowQueue {
    float4* o; // origin.xyz, ray length
    float4* d; // direction.xyz, bits(pixel)
    float4* c; // contribution rgb, bits(flags): parity mode keeps finished entries
};

enum : uint32_t { FLAG_FINISHED = 1, FLAG_LASTSPECULAR = 2 }; // shading.cl:11-14
// state word of a queued ray: bits 0-7 flags, 8-15 bounce count, 16-31 accumulator plane.
// Several samples of one pixel can be in flight at once; each one deposits into its own plane so
// that there is still exactly one live path per accumulator entry (no atomics, deterministic sums).
__host__ __device__ inline uint32_t packState(uint32_t flags, uint32_t bounce, uint32_t plane) { return (flags & 0xFFu) | ((bounce & 0xFFu) << 8) | (plane << 16); }
struct AccumView {
    float4* plane0; // the HDR accumulator proper (width*height float4, indexed by global pixel)
    float4* extra; // planes 1.. (scratch, folded into plane0 after every batch): [owned-pixel ordinal][plane - 1]
    const uint32_t* ordinal; // global pixel -> ordinal among the pixels this context owns; null: the identity (whole frame owned)
    uint32_t perPixel; // extra planes per pixel = samples in flight - 1
    // The extra planes exist only for the pixels a context owns (a rank of an N-GPU job holds 1/N of them) and the
    // planes of one pixel are adjacent: k_gen hands consecutive queue entries consecutive samples of the SAME
    // pixel, so the lanes of a wave deposit into one run of memory (and read one ordinal).
    __device__ inline float4* at(uint32_t plane, uint32_t pixel) const
    {
        if (plane == 0u)
            return plane0 + pixel;
        const uint32_t slot = ordinal ? ordinal[pixel] : pixel;
        return extra + (size_t)slot * perPixel + (plane - 1u);
    }
    // same, for a caller that already knows the ordinal (k_fold_planes walks the owned pixels in order)
    __device__ inline float4* atOrdinal(uint32_t plane, uint32_t slot) const { return extra + (size_t)slot * perPixel + (plane - 1u); }
};

struct CameraDev { // Camera, camera.cl:7-26
    float4 eye, screen, u, v, uN, vN;
    float focalDistance, apertureRadius, relativeAperture, shutterTime, ISO;
    uint32_t thinLens;
};

// per-sample control block in device memory: queue counts per pass and work cursors of the
// persistent kernels.  Indexed by pass so that no kernel ever resets a word another kernel of the
// same sample still reads (the reference needs a dedicated updateKernelData launch for that,
// kernel.cl:303-317).
constexpr int kMaxPasses = 16;
#ifndef PT_GEN_INTERLEAVE
#define PT_GEN_INTERLEAVE 256 // at most this many samples of one pixel are neighbours in the primary-ray queue (power of two; 1: sample-major order); 32 / 64 / 128 / 256: 8.27 / 8.41 / 8.42 / 8.47 Grays/s with 256 in flight
#endif
constexpr uint32_t kGenInterleave = PT_GEN_INTERLEAVE;
constexpr uint32_t kDepositSlots = 64;
struct Control {
    uint32_t extCount[kMaxPasses + 1]; // rays in the extension queue at pass p
    uint32_t shadowCount[kMaxPasses + 1];
    uint32_t extCursor[kMaxPasses + 1]; // persistent-kernel fetch cursors
    uint32_t shadowCursor[kMaxPasses + 1];
    uint32_t shadeHits[kMaxPasses + 1];
    uint32_t depositsShade; // accumulator updates of the sample made by k_shade (emissive hits, sky misses) ...
    uint32_t depositsShadow; // ... and by the any-hit traversal (unoccluded shadow rays)
    uint32_t generated;
    uint32_t _pad;
    // k_shade's deposit count, spread: workgroup b adds to slot b mod kDepositSlots, each slot in a cache line of its own.  The queue counters above must be
    // single words (they hand out slots); this one is a statistic, and as ONE word it cost k_shade 1 ms per 531 M-entry batch: a million workgroups -- half of
    // them hold nothing but rays that left the scene and would issue no other atomic -- queueing on one address.  k_end_sample sums the slots.
    uint32_t depositSlots[kDepositSlots][16];
};
struct Totals {
    unsigned long long raysExtension, raysShadow, raysGenerated, shadeHits, deposits, depositsShadow;
};

} // namespace ptd
