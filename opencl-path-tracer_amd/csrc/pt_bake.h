// World-space copies of instances, made ON THE DEVICE (the dynamic part of a scene state, pt_upload_dynamic_async).
//
// The reference re-uploads its top-level BVH, lights and the dynamic tail of its vertex / node buffers every frame tick
// (src/raytracer.cpp:497-595).  Here the bottom-level trees are converted ONCE, at pt_upload_static: collapsed to 4-wide nodes,
// packed breadth-first per mesh root (a mesh's nodes are one contiguous run), with the exact child boxes and the per-leaf
// offsets a copy needs kept next to them.  A tick then uploads the top level, the instance table and the lights (a few KB) and --
// where instances are copied to world space, the library's default while a byte budget lasts -- launches two kernels on the copy
// stream: one thread per (instance, node) re-fits and re-quantises the node's child boxes around their transformed corners, one
// thread per (instance, triangle reference) transforms a triangle.  Round 2 did all of this on the host, single-threaded, and
// copied the result (110 MB for the benchmark scene) through pinned staging: 240 ms per tick.
#pragma once
#include "pt_device.h"
#include <cmath>

namespace ptd {

// Quantise up to four child boxes into a WideNode (pt_device.h): origin = min corner of their union, per-axis power-of-two scale
// with (extent / scale) <= 255, planes rounded OUTWARDS and then verified with the exact expression the traversal kernels evaluate
// (origin + scale * q).  An empty slot gets an inverted box and `emptyRef`.  Shared by the host (collapse of the caller's binary
// trees) and the device (world-space copies), so that both produce the same bytes from the same boxes.
__host__ __device__ inline void quantiseWideNode(const float (*lo)[3], const float (*hi)[3], const uint32_t* refs, const bool* empty, uint32_t emptyRef, WideNode* out)
{
    float nlo[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f }, nhi[3] = { -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f };
    for (int k = 0; k < 4; k++)
        for (int a = 0; a < 3; a++)
            if (!empty[k] && lo[k][a] <= hi[k][a]) {
                nlo[a] = fminf(nlo[a], lo[k][a]);
                nhi[a] = fmaxf(nhi[a], hi[k][a]);
            }
    WideNode w {};
    float scale[3];
    for (int a = 0; a < 3; a++) {
        if (!(nlo[a] <= nhi[a]))
            nlo[a] = nhi[a] = 0.f;
        // smallest power of two s with (hi - lo) / s <= 255, evaluated in float like the kernel does
        int e = 0;
        const float extent = nhi[a] - nlo[a];
        (void)frexpf(extent / 255.0f, &e); // extent/255 = m * 2^e, m in [0.5,1)  =>  2^e >= extent/255
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        scale[a] = ldexpf(1.0f, e);
        while (extent > 0.f && nlo[a] + scale[a] * 255.0f < nhi[a] && e < 127) // guard float round-off
            scale[a] = ldexpf(1.0f, ++e);
    }
    w.ox = nlo[0], w.oy = nlo[1], w.oz = nlo[2];
    w.scaleX = scale[0], w.scaleY = scale[1], w.scaleZ = scale[2];
    uint32_t q[6] = { 0, 0, 0, 0, 0, 0 }; // qlox, qhix, qloy, qhiy, qloz, qhiz
    for (int k = 0; k < 4; k++) {
        w.child[k] = empty[k] ? emptyRef : refs[k];
        for (int a = 0; a < 3; a++) {
            uint32_t ql = 255, qh = 0;
            if (!empty[k]) {
                const float fl = floorf((lo[k][a] - nlo[a]) / scale[a]);
                const float fh = ceilf((hi[k][a] - nlo[a]) / scale[a]);
                ql = (uint32_t)fmaxf(0.f, fminf(255.f, fl));
                qh = (uint32_t)fmaxf(0.f, fminf(255.f, fh));
                while (ql > 0 && nlo[a] + scale[a] * (float)ql > lo[k][a])
                    ql--;
                while (qh < 255 && nlo[a] + scale[a] * (float)qh < hi[k][a])
                    qh++;
            }
            q[a * 2] |= ql << (8 * k);
            q[a * 2 + 1] |= qh << (8 * k);
        }
    }
    w.qlox = q[0], w.qhix = q[1], w.qloy = q[2], w.qhiy = q[3], w.qloz = q[4], w.qhiz = q[5];
    *out = w;
}

// What a world-space copy of one instance needs (built on the host per tick, a few dozen bytes per instance).
struct BakeJob {
    double m[12]; // rows 0..2 of the WORLD transform (inverse of the top-level leaf's invTransform), row-major 3 x 4
    uint32_t srcNode, numNodes; // the mesh's run of packed bottom-level nodes (0 nodes: the mesh is a single leaf)
    uint32_t dstNode; // where the copy's nodes go (same order)
    uint32_t srcRef, numRefs; // the mesh's run in the table of triangle references (leaf order; an SBVH references a triangle more than once)
    uint32_t dstTri; // where the copy's triangles go
    uint32_t instance; // instance index reported for hits on the copy
    uint32_t _pad;
};

// exact boxes of a packed node's (up to) four children, object space -- the quantised planes of the node itself are already rounded
struct WideBoxes {
    float lo[4][3], hi[4][3];
};

struct BakeArgs {
    const BakeJob* jobs;
    const WideNode* srcWide; // the mesh trees, object space
    const WideBoxes* srcBoxes;
    const uint32_t* srcLeafOfs; // [node][child]: offset of a leaf child's first triangle reference inside the mesh's run
    const uint32_t* refTri; // triangle reference -> caller's triangle index
    const TriIsect* srcTris; // caller's numbering, object space
    WideNode* dstWide;
    TriIsect* dstTris;
    uint32_t emptyRef;
};

// one thread per (job = blockIdx.y, node): child boxes re-fitted around their transformed corners (looser for rotated instances, still
// conservative), two ulps outwards for the rounding of the transformed triangles themselves, then quantised like any other node
__global__ void __launch_bounds__(128) k_bake_nodes(BakeArgs a)
{
    const BakeJob& j = a.jobs[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.numNodes)
        return;
    const WideNode src = a.srcWide[j.srcNode + i];
    const WideBoxes bx = a.srcBoxes[j.srcNode + i];
    float lo[4][3], hi[4][3];
    uint32_t refs[4];
    bool empty[4];
    for (int k = 0; k < 4; k++) {
        const uint32_t r = src.child[k];
        empty[k] = r == a.emptyRef || !(bx.lo[k][0] <= bx.hi[k][0]);
        refs[k] = a.emptyRef;
        for (int ax = 0; ax < 3; ax++)
            lo[k][ax] = 1.f, hi[k][ax] = -1.f;
        if (empty[k])
            continue;
        double wlo[3] = { 1e300, 1e300, 1e300 }, whi[3] = { -1e300, -1e300, -1e300 };
        for (int corner = 0; corner < 8; corner++) {
            const double p[3] = { (corner & 1) ? bx.hi[k][0] : bx.lo[k][0], (corner & 2) ? bx.hi[k][1] : bx.lo[k][1], (corner & 4) ? bx.hi[k][2] : bx.lo[k][2] };
            for (int ax = 0; ax < 3; ax++) {
                const double q = j.m[ax * 4 + 0] * p[0] + j.m[ax * 4 + 1] * p[1] + j.m[ax * 4 + 2] * p[2] + j.m[ax * 4 + 3];
                wlo[ax] = fmin(wlo[ax], q), whi[ax] = fmax(whi[ax], q);
            }
        }
        for (int ax = 0; ax < 3; ax++) {
            lo[k][ax] = nextafterf(nextafterf((float)wlo[ax], -INFINITY), -INFINITY);
            hi[k][ax] = nextafterf(nextafterf((float)whi[ax], INFINITY), INFINITY);
        }
        refs[k] = refCount(r) == 0u ? makeRef(j.dstNode + (refIndex(r) - j.srcNode), 0u) : makeRef(j.dstTri + a.srcLeafOfs[(size_t)(j.srcNode + i) * 4 + k], refCount(r));
    }
    WideNode out;
    quantiseWideNode(lo, hi, refs, empty, a.emptyRef, &out);
    a.dstWide[j.dstNode + i] = out;
}

// one thread per (job, triangle reference): v0 and the two edges through the world transform (double, rounded once); the copy remembers
// (caller's triangle, instance) so that a hit on it is reported like a hit inside the instance ((t, u, v) are the same in both spaces:
// the reference never renormalises the transformed direction, scene.cl:118-121)
__global__ void __launch_bounds__(256) k_bake_tris(BakeArgs a)
{
    const BakeJob& j = a.jobs[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.numRefs)
        return;
    const uint32_t orig = a.refTri[j.srcRef + i];
    const TriIsect t = a.srcTris[orig];
    const double v0[3] = { t.a.x, t.a.y, t.a.z }, e1[3] = { t.a.w, t.b.x, t.b.y }, e2[3] = { t.b.z, t.b.w, t.c.x };
    float V0[3], E1[3], E2[3];
    for (int r = 0; r < 3; r++) {
        V0[r] = (float)(j.m[r * 4 + 0] * v0[0] + j.m[r * 4 + 1] * v0[1] + j.m[r * 4 + 2] * v0[2] + j.m[r * 4 + 3]);
        E1[r] = (float)(j.m[r * 4 + 0] * e1[0] + j.m[r * 4 + 1] * e1[1] + j.m[r * 4 + 2] * e1[2]);
        E2[r] = (float)(j.m[r * 4 + 0] * e2[0] + j.m[r * 4 + 1] * e2[1] + j.m[r * 4 + 2] * e2[2]);
    }
    TriIsect b;
    b.a = make_float4(V0[0], V0[1], V0[2], E1[0]);
    b.b = make_float4(E1[1], E1[2], E2[0], E2[1]);
    b.c = make_float4(E2[2], __uint_as_float(orig), __uint_as_float(j.instance), 0.f);
    a.dstTris[j.dstTri + i] = b;
}

// The per-instance copies of mesh root nodes that folded instances are reached through (pt_trace.h, TWO_LEVEL): copy k = the packed root node of
// instance k's mesh as the DEVICE holds it (after a refit the host's mirror of the packed nodes is stale).  src[k] = ~0: the slot was filled by the host
// (a single-leaf mesh: a one-child node around the leaf) or is unused.
__global__ void __launch_bounds__(64) k_inst_roots(const WideNode* wide, const uint32_t* src, WideNode* dst, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && src[k] != 0xFFFFFFFFu)
        dst[k] = wide[src[k]];
}

// ---- refit (pt_update_geometry): a deformed frame of the SAME topology ------------------------------------------------------------------
// The caller's vertex record (pt_vertex, include/ptamd.h: 48 bytes) as the device reads it
struct VertexIn {
    float4 pos; // xyz
    float4 normal;
    float u, v, _p0, _p1;
};
struct RefitArgs {
    const VertexIn* verts;
    const TriShade* tri; // vertex indices + material of every triangle of the caller's numbering (static)
    const Material* mats;
    TriIsect* tris;
    TriFat* fat;
    uint32_t n;
};
// The caller's sub-BVH node record (pt_sub_bvh_node, include/ptamd.h: 48 bytes) as the device reads it
struct SubNodeIn {
    float mn[4], mx[4];
    uint32_t left, count, _p0, _p1;
};
struct RefitNodeArgs {
    const SubNodeIn* nodes; // the caller's refitted nodes, as handed in
    const uint32_t* kidBoxNode; // [packed node][child]: the caller's node whose box this child slot takes; 0x80000000 | i: entry i of `extra`; ~0: unused slot
    const float* extra; // boxes (lo.xyz, hi.xyz) of the pair nodes that split an oversized leaf: recomputed on the host (rare)
    WideNode* wide; // in place: the child references stay, the planes are re-made
    WideBoxes* boxes;
    uint32_t emptyRef, n;
};
// one thread per packed 4-wide node: the gather and the re-quantisation pt_upload_static does on the host, from the same boxes with the
// same routine (quantiseWideNode) -- a refitted context and a fresh one hold the same bytes
__global__ void __launch_bounds__(128) k_refit_nodes(RefitNodeArgs a)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n)
        return;
    const WideNode w = a.wide[q];
    float lo[4][3], hi[4][3];
    uint32_t refs[4];
    bool empty[4];
    WideBoxes bx;
    for (int k = 0; k < 4; k++) {
        const uint32_t src = a.kidBoxNode[(size_t)q * 4 + k];
        empty[k] = src == 0xFFFFFFFFu;
        refs[k] = w.child[k];
        for (int ax = 0; ax < 3; ax++)
            lo[k][ax] = 1.f, hi[k][ax] = -1.f;
        if (!empty[k]) {
            if (src & 0x80000000u) {
                const float* e = a.extra + (size_t)(src & 0x7FFFFFFFu) * 6;
                for (int ax = 0; ax < 3; ax++)
                    lo[k][ax] = e[ax], hi[k][ax] = e[3 + ax];
            } else {
                const SubNodeIn n = a.nodes[src];
                for (int ax = 0; ax < 3; ax++)
                    lo[k][ax] = n.mn[ax], hi[k][ax] = n.mx[ax];
            }
        }
        for (int ax = 0; ax < 3; ax++)
            bx.lo[k][ax] = lo[k][ax], bx.hi[k][ax] = hi[k][ax];
    }
    a.boxes[q] = bx;
    WideNode out;
    quantiseWideNode(lo, hi, refs, empty, a.emptyRef, &out);
    a.wide[q] = out;
}

// ---- refit on the device alone (pt_refit_vertices, round 5): what refitBVH does on the host (reference src/bvh/refit_bvh.cpp:6-34) --------------------
// The caller hands over nothing but the vertices of the deformed mesh; every box of the packed trees is recomputed here, bottom-up: the box
// of a leaf slot from the vertices of its triangles, the box of an inner slot as the union of that child's slots.  One thread per packed node
// computes its leaf slots; a node whose slots are all final (its own thread and one arrival per inner child, counted by an atomic) is
// quantised by whoever arrives last, which then carries the node's union up to the parent's slot and goes on there (Karras 2012's bottom-up
// pass on a 4-wide tree).  min / max are exact, so the boxes -- and the quantised planes -- are the bits k_refit_nodes makes from the caller's
// own refitBVH: a context refitted this way and a fresh one on the host-refitted arrays hold the same bytes.
struct RefitTreeArgs {
    const VertexIn* verts;
    const TriShade* tri; // vertex indices of every triangle of the caller's numbering
    WideNode* wide; // in place: references stay, planes are re-made
    WideBoxes* boxes;
    const uint32_t* parent; // [node]: (parent node << 2) | slot, ~0: the root of a run
    const uint32_t* need; // [node]: arrivals that complete it = inner children + 1
    uint32_t* arrived; // [node]: zero between refits (the last arrival resets it)
    uint32_t emptyRef, n;
};
__global__ void __launch_bounds__(128) k_refit_tree(RefitTreeArgs a)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n)
        return;
    {
        const WideNode w = a.wide[q];
        for (int k = 0; k < 4; k++) {
            const uint32_t r = w.child[k];
            if (r == a.emptyRef) {
                for (int ax = 0; ax < 3; ax++)
                    a.boxes[q].lo[k][ax] = 1.f, a.boxes[q].hi[k][ax] = -1.f;
                continue;
            }
            if (refCount(r) == 0u)
                continue; // an inner child: its own last arrival writes this slot
            float lo[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f }, hi[3] = { -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f };
            for (uint32_t t = refIndex(r); t < refIndex(r) + refCount(r); t++) {
                const TriShade ts = a.tri[t];
                const uint32_t vi[3] = { ts.i0, ts.i1, ts.i2 };
                for (int c = 0; c < 3; c++) {
                    const float4 p = a.verts[vi[c]].pos;
                    lo[0] = fminf(lo[0], p.x), lo[1] = fminf(lo[1], p.y), lo[2] = fminf(lo[2], p.z);
                    hi[0] = fmaxf(hi[0], p.x), hi[1] = fmaxf(hi[1], p.y), hi[2] = fmaxf(hi[2], p.z);
                }
            }
            for (int ax = 0; ax < 3; ax++)
                a.boxes[q].lo[k][ax] = lo[ax], a.boxes[q].hi[k][ax] = hi[ax];
        }
    }
    uint32_t node = q;
    while (true) {
        __threadfence(); // what this thread wrote is visible before it is counted ...
        const uint32_t arrivals = atomicAdd(&a.arrived[node], 1u) + 1u;
        if (arrivals < a.need[node])
            return; // somebody else completes this node
        __threadfence(); // ... and what the others wrote is visible to the one that counts last
        a.arrived[node] = 0u;
        const volatile WideBoxes* vb = (const volatile WideBoxes*)&a.boxes[node];
        const volatile WideNode* vw = (const volatile WideNode*)&a.wide[node];
        float lo[4][3], hi[4][3], ulo[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f }, uhi[3] = { -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f };
        uint32_t refs[4];
        bool empty[4];
        for (int k = 0; k < 4; k++) {
            refs[k] = vw->child[k];
            empty[k] = refs[k] == a.emptyRef;
            for (int ax = 0; ax < 3; ax++) {
                lo[k][ax] = vb->lo[k][ax], hi[k][ax] = vb->hi[k][ax];
                if (!empty[k])
                    ulo[ax] = fminf(ulo[ax], lo[k][ax]), uhi[ax] = fmaxf(uhi[ax], hi[k][ax]);
            }
        }
        WideNode out;
        quantiseWideNode(lo, hi, refs, empty, a.emptyRef, &out);
        a.wide[node] = out;
        const uint32_t p = a.parent[node];
        if (p == 0xFFFFFFFFu)
            return;
        for (int ax = 0; ax < 3; ax++)
            a.boxes[p >> 2].lo[p & 3u][ax] = ulo[ax], a.boxes[p >> 2].hi[p & 3u][ax] = uhi[ax];
        node = p >> 2;
    }
}

// one thread per triangle: the intersection record (v0, e1, e2: the very subtractions pt_upload_static does on the host, so that a refitted
// context and a fresh one hold the same bits) and the 128-byte shading record, from the new vertices
__global__ void __launch_bounds__(256) k_refit_tris(RefitArgs a)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n)
        return;
    const TriShade ts = a.tri[t];
    const VertexIn a0 = a.verts[ts.i0], a1 = a.verts[ts.i1], a2 = a.verts[ts.i2];
    const float e1x = a1.pos.x - a0.pos.x, e1y = a1.pos.y - a0.pos.y, e1z = a1.pos.z - a0.pos.z; // shapes.cl:37-38
    const float e2x = a2.pos.x - a0.pos.x, e2y = a2.pos.y - a0.pos.y, e2z = a2.pos.z - a0.pos.z;
    TriIsect ti;
    ti.a = make_float4(a0.pos.x, a0.pos.y, a0.pos.z, e1x);
    ti.b = make_float4(e1y, e1z, e2x, e2y);
    ti.c = make_float4(e2z, 0.f, 0.f, 0.f);
    a.tris[t] = ti;
    const Material m = a.mats[ts.material];
    TriFat f;
    f.n0u = make_float4(a0.normal.x, a0.normal.y, a0.normal.z, a0.u);
    f.n1u = make_float4(a1.normal.x, a1.normal.y, a1.normal.z, a1.u);
    f.n2u = make_float4(a2.normal.x, a2.normal.y, a2.normal.z, a2.u);
    f.vvvm = make_float4(a0.v, a1.v, a2.v, __uint_as_float(ts.material));
    f.e1e = make_float4(e1x, e1y, e1z, e2x);
    f.e2v = make_float4(e2y, e2z, a0.pos.x, a0.pos.y);
    f.v0c = make_float4(a0.pos.z, m.colour.x, m.colour.y, m.colour.z);
    f.mat = make_float4(m.params.x, m.params.y, m.params.z, m.typeAndPad.x);
    a.fat[t] = f;
}

// What a device copy reaches on the box at hand (bench.py, roofline.peak_measured_copy): a float4 grid-stride copy -- every lane 16 bytes per access, a wave 1 KiB,
// four workgroups of 256 per CU.  The best of the sixty shapes tools/micro/copy_probe.hip tries on this pool's boxes (1 / 4 / 8 accesses in flight per thread,
// plain / non-temporal, 4-32 workgroups per CU, 256-1024 threads): 5.5-5.6 TB/s of a 2 x 2 GiB copy; hipMemcpyAsync and torch's Tensor.copy_ (rounds 1-5's
// probe) reach 4.3-5.1.  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy: not reproduced here -- the line's `frac` is of the 8 TB/s spec peak either way.
__global__ void __launch_bounds__(256) k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = src[i];
}

} // namespace ptd
