// 8-wide compressed BVH for gfx950 (host-side builder + node layout).
//
// Why: k_trace is bound by the number of vector-memory instructions it issues per ray (DESIGN.md section 6:
// every extra 16-byte load per node visit costs ~8 % of the kernel).  The 4-wide node needs 4 loads for 4
// children (16 B per child); this layout needs 5 loads for 8 children (10 B per child) and ~40 % fewer node
// visits per ray.  It follows the published "compressed wide BVH" idea (Ylitie, Karras, Laine 2017): 8-bit
// child planes on a per-node power-of-two grid, children addressed implicitly -- inner children are stored
// consecutively from `childBase`, leaf items consecutively from `itemBase`, one meta byte per child -- and the
// children sit in slots chosen so that `slot ^ octant` is a usable front-to-back order, which removes the
// per-visit distance sort.  The traversal stack then holds one 8-byte GROUP (base + bit mask of pending
// children or items) per visited node instead of one entry per child.
//
// Both levels of the reference's scene (assets/cl/scene.cl:61-271) use the same node: at the bottom level an
// item is a triangle, at the top level an item is an instance to enter or a world-space triangle of a baked
// single-leaf instance (`items[]`).  Quantised planes are rounded outwards and verified with the expression the
// kernel evaluates, so the set of triangles tested is a superset of the reference's.
#pragma once
#include "pt_device.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <unordered_map>
#include <vector>

namespace ptd {

struct Node8 { // 80 B = five 16-byte loads
    float ox, oy, oz; // origin of the quantisation grid = min corner of the union of the children
    uint32_t exImask; // ex | ey << 8 | ez << 16 (biased float exponents of the grid step) | imask << 24
    uint32_t childBase; // index of the first inner child; child in slot s is childBase + popc(imask & ((1 << s) - 1))
    uint32_t itemBase; // index of the first leaf item (triangle, or top-level item)
    uint32_t meta[2]; // byte s: 0 = empty; inner: 0x20 | (24 + s); leaf: unary item count << 5 | item offset (0..23)
    uint32_t qlox[2], qloy[2]; // byte s of each array belongs to the child in slot s
    uint32_t qloz[2], qhix[2];
    uint32_t qhiy[2], qhiz[2];
};
static_assert(sizeof(Node8) == 80, "Node8 is five 16-byte chunks");

constexpr uint32_t kItemBakedTriangle = 0x80000000u; // top-level item: world-space triangle (else: instance index)
constexpr uint32_t kGroupLeaveInstance = 0xFFFFFFFFu; // stack sentinel (x of an item group with an empty mask)

struct Wide8 {
    std::vector<Node8> nodes;
    std::vector<TriIsect> tris; // re-emitted in node order; c.y = original primitive, c.z = instance of a baked copy or ~0u
    std::vector<uint32_t> items; // top-level items
    uint32_t topRoot = 0; // node index of the (wrapped) top-level root
    uint32_t stackNeed = 0; // worst-case number of pending groups
    std::unordered_map<uint32_t, uint32_t> rootOf; // bottom-level root reference -> node index
};

namespace wide8_detail {
struct Kid {
    float lo[3], hi[3];
    uint32_t ref; // reference in the pair-node tree (inner / leaf / special)
};
inline float areaOf(const Kid& k)
{
    const float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
inline Kid kidOf(const PairNode& n, int side)
{
    Kid c;
    const float* bx = &n.bx.x;
    const float* by = &n.by.x;
    const float* bz = &n.bz.x;
    c.lo[0] = bx[side * 2], c.hi[0] = bx[side * 2 + 1];
    c.lo[1] = by[side * 2], c.hi[1] = by[side * 2 + 1];
    c.lo[2] = bz[side * 2], c.hi[2] = bz[side * 2 + 1];
    c.ref = side ? n.right : n.left;
    return c;
}
inline bool validKid(const Kid& k) { return k.ref != kRefNone && k.lo[0] <= k.hi[0] && k.lo[1] <= k.hi[1] && k.lo[2] <= k.hi[2]; }
inline void triBounds(const TriIsect& t, float lo[3], float hi[3])
{
    const float v0[3] = { t.a.x, t.a.y, t.a.z }, e1[3] = { t.a.w, t.b.x, t.b.y }, e2[3] = { t.b.z, t.b.w, t.c.x };
    for (int a = 0; a < 3; a++) {
        const float p1 = v0[a] + e1[a], p2 = v0[a] + e2[a];
        // v0 + e is the vertex up to one rounding: widen by an ulp on each side
        lo[a] = std::nextafter(std::min(v0[a], std::min(p1, p2)), -INFINITY);
        hi[a] = std::nextafter(std::max(v0[a], std::max(p1, p2)), INFINITY);
    }
}
} // namespace wide8_detail

// `pair`: the unified pair-node tree (bottom levels + top level), references as in pt_device.h.
// `tris`: triangles the leaf references index (c.y / c.z already set for baked world-space copies, see ptamd.hip).
// `bottomRoots`: root references of the instanced meshes; `topRootRef`: reference of the top-level root.
// `topRootLo/Hi`: world bounds of the top-level root (used when the root is a single instance or leaf).
inline Wide8 buildWide8(std::vector<PairNode> pair, const std::vector<TriIsect>& tris, const std::vector<uint32_t>& bottomRoots,
    uint32_t topRootRef, const float topRootLo[3], const float topRootHi[3], uint32_t firstBakedTri)
{
    using namespace wide8_detail;
    // 1. a leaf child can hold at most 3 items (unary count in 3 bits): split larger leaves into synthetic pair nodes
    auto splitLeaf = [&](auto&& self, uint32_t first, uint32_t count) -> uint32_t {
        if (count <= 3u)
            return makeRef(first, count);
        const uint32_t nl = (count + 1u) / 2u;
        PairNode pn {};
        float lo[2][3], hi[2][3];
        for (int s = 0; s < 2; s++) {
            for (int a = 0; a < 3; a++)
                lo[s][a] = FLT_MAX, hi[s][a] = -FLT_MAX;
            const uint32_t b = s ? first + nl : first, e = s ? first + count : first + nl;
            for (uint32_t t = b; t < e; t++) {
                float l[3], h[3];
                triBounds(tris[t], l, h);
                for (int a = 0; a < 3; a++)
                    lo[s][a] = std::min(lo[s][a], l[a]), hi[s][a] = std::max(hi[s][a], h[a]);
            }
        }
        pn.bx = make_float4(lo[0][0], hi[0][0], lo[1][0], hi[1][0]);
        pn.by = make_float4(lo[0][1], hi[0][1], lo[1][1], hi[1][1]);
        pn.bz = make_float4(lo[0][2], hi[0][2], lo[1][2], hi[1][2]);
        pn.left = self(self, first, nl);
        pn.right = self(self, first + nl, count - nl);
        pair.push_back(pn);
        return makeRef((uint32_t)pair.size() - 1u, 0u);
    };
    const size_t originalNodes = pair.size();
    for (size_t i = 0; i < originalNodes; i++) {
        for (int side = 0; side < 2; side++) {
            const uint32_t r = side ? pair[i].right : pair[i].left;
            if (r != kRefNone && refCount(r) > 3u && refCount(r) != kRefSpecial) {
                const uint32_t nr = splitLeaf(splitLeaf, refIndex(r), refCount(r));
                (side ? pair[i].right : pair[i].left) = nr;
            }
        }
    }

    Wide8 out;
    struct Job {
        uint32_t ref; // inner pair node to collapse, or a lone leaf / special reference to wrap
        float lo[3], hi[3]; // box of a wrapped reference
        uint32_t node; // output index
        bool top;
    };
    std::vector<Job> jobs;
    auto wholeBox = [&](Job& j) {
        for (int a = 0; a < 3; a++)
            j.lo[a] = topRootLo[a], j.hi[a] = topRootHi[a];
    };

    auto emit = [&](uint32_t rootRef, bool top) -> uint32_t {
        if (rootRef != kRefNone && refCount(rootRef) > 3u && refCount(rootRef) != kRefSpecial)
            rootRef = splitLeaf(splitLeaf, refIndex(rootRef), refCount(rootRef)); // a mesh that is one large leaf
        const uint32_t rootNode = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Job j {};
        j.ref = rootRef, j.node = rootNode, j.top = top;
        wholeBox(j);
        if (refCount(rootRef) != 0u && refCount(rootRef) != kRefSpecial) { // lone leaf: its real bounds
            for (int a = 0; a < 3; a++)
                j.lo[a] = FLT_MAX, j.hi[a] = -FLT_MAX;
            for (uint32_t t = 0; t < refCount(rootRef); t++) {
                float l[3], h[3];
                triBounds(tris[refIndex(rootRef) + t], l, h);
                for (int a = 0; a < 3; a++)
                    j.lo[a] = std::min(j.lo[a], l[a]), j.hi[a] = std::max(j.hi[a], h[a]);
            }
        }
        jobs.push_back(j);
        for (size_t jq = jobs.size() - 1; jq < jobs.size(); jq++) {
            const Job job = jobs[jq];
            // ---- collect up to 8 children: surface-area greedy opening of inner children ----------------------
            Kid kids[8];
            int n = 0;
            if (refCount(job.ref) == 0u && job.ref != kRefNone) {
                const PairNode& pn = pair[refIndex(job.ref)];
                for (int side = 0; side < 2; side++) {
                    const Kid k = kidOf(pn, side);
                    if (validKid(k))
                        kids[n++] = k;
                }
                while (n < 8) {
                    int best = -1;
                    float bestArea = -1.f;
                    for (int k = 0; k < n; k++)
                        if (refCount(kids[k].ref) == 0u && areaOf(kids[k]) > bestArea)
                            best = k, bestArea = areaOf(kids[k]);
                    if (best < 0)
                        break;
                    const PairNode& g = pair[refIndex(kids[best].ref)];
                    const Kid l = kidOf(g, 0), r = kidOf(g, 1);
                    const bool lv = validKid(l), rv = validKid(r);
                    if (lv && rv) {
                        kids[best] = l;
                        kids[n++] = r;
                    } else if (lv || rv) {
                        kids[best] = lv ? l : r;
                    } else {
                        kids[best] = kids[--n];
                    }
                }
            } else if (job.ref != kRefNone) { // wrapped leaf / instance
                Kid k;
                for (int a = 0; a < 3; a++)
                    k.lo[a] = job.lo[a], k.hi[a] = job.hi[a];
                k.ref = job.ref;
                kids[n++] = k;
            }
            // ---- grid of the node ---------------------------------------------------------------------
            float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            for (int k = 0; k < n; k++)
                for (int a = 0; a < 3; a++)
                    lo[a] = std::min(lo[a], kids[k].lo[a]), hi[a] = std::max(hi[a], kids[k].hi[a]);
            Node8 w {};
            uint32_t ex[3];
            float scale[3];
            for (int a = 0; a < 3; a++) {
                if (!(lo[a] <= hi[a]))
                    lo[a] = hi[a] = 0.f;
                int e = 0;
                const float extent = hi[a] - lo[a];
                std::frexp(extent / 255.0f, &e);
                e = std::max(-126, std::min(e, 127));
                scale[a] = std::ldexp(1.0f, e);
                while (extent > 0.f && lo[a] + scale[a] * 255.0f < hi[a] && e < 127)
                    scale[a] = std::ldexp(1.0f, ++e);
                ex[a] = (uint32_t)(e + 127);
            }
            w.ox = lo[0], w.oy = lo[1], w.oz = lo[2];
            // ---- slots: child in slot s is the first one a ray with negative-direction mask s should enter ----
            int slotOf[8], kidIn[8];
            for (int s = 0; s < 8; s++)
                kidIn[s] = -1;
            for (int k = 0; k < n; k++)
                slotOf[k] = -1;
            const float centre[3] = { 0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]) };
            for (int round = 0; round < n; round++) {
                int bk = -1, bs = -1;
                float bestScore = -FLT_MAX;
                for (int k = 0; k < n; k++) {
                    if (slotOf[k] >= 0)
                        continue;
                    for (int s = 0; s < 8; s++) {
                        if (kidIn[s] >= 0)
                            continue;
                        float score = 0.f; // how far upstream the child lies for rays travelling along sign vector s
                        for (int a = 0; a < 3; a++) {
                            const float c = 0.5f * (kids[k].lo[a] + kids[k].hi[a]) - centre[a];
                            score += ((s >> a) & 1) ? c : -c;
                        }
                        if (score > bestScore)
                            bestScore = score, bk = k, bs = s;
                    }
                }
                slotOf[bk] = bs;
                kidIn[bs] = bk;
            }
            // ---- children in slot order: inner ones get consecutive nodes, leaf ones consecutive items -------
            uint32_t imask = 0, nInner = 0, nItems = 0;
            w.childBase = (uint32_t)out.nodes.size();
            w.itemBase = job.top ? (uint32_t)out.items.size() : (uint32_t)out.tris.size();
            uint32_t* q[6] = { w.qlox, w.qloy, w.qloz, w.qhix, w.qhiy, w.qhiz };
            for (int s = 0; s < 8; s++) {
                const int k = kidIn[s];
                uint32_t meta = 0;
                if (k >= 0) {
                    const uint32_t r = kids[k].ref;
                    if (refCount(r) == 0u) {
                        imask |= 1u << s;
                        meta = 0x20u | (24u + (uint32_t)s);
                        Job cj {};
                        cj.ref = r, cj.node = w.childBase + nInner, cj.top = job.top;
                        nInner++;
                        jobs.push_back(cj);
                    } else {
                        uint32_t count = 0;
                        if (refCount(r) == kRefSpecial) { // instance (top level only)
                            out.items.push_back(refIndex(r));
                            count = 1;
                        } else {
                            count = refCount(r); // <= 3 after the split above
                            for (uint32_t t = 0; t < count; t++) {
                                const uint32_t src = refIndex(r) + t;
                                TriIsect tr = tris[src];
                                if (src < firstBakedTri) {
                                    tr.c.y = __builtin_bit_cast(float, src);
                                    tr.c.z = __builtin_bit_cast(float, 0xFFFFFFFFu);
                                }
                                if (job.top)
                                    out.items.push_back(kItemBakedTriangle | (uint32_t)out.tris.size());
                                out.tris.push_back(tr);
                            }
                        }
                        meta = (((1u << count) - 1u) << 5) | nItems;
                        nItems += count;
                    }
                }
                w.meta[s >> 2] |= meta << (8 * (s & 3));
                for (int a = 0; a < 3; a++) {
                    uint32_t ql = 255, qh = 0; // empty slot: inverted box
                    if (k >= 0) {
                        const float fl = std::floor((kids[k].lo[a] - lo[a]) / scale[a]);
                        const float fh = std::ceil((kids[k].hi[a] - lo[a]) / scale[a]);
                        ql = (uint32_t)std::max(0.f, std::min(255.f, fl));
                        qh = (uint32_t)std::max(0.f, std::min(255.f, fh));
                        while (ql > 0 && lo[a] + scale[a] * (float)ql > kids[k].lo[a])
                            ql--;
                        while (qh < 255 && lo[a] + scale[a] * (float)qh < kids[k].hi[a])
                            qh++;
                    }
                    q[a][s >> 2] |= ql << (8 * (s & 3));
                    q[3 + a][s >> 2] |= qh << (8 * (s & 3));
                }
            }
            out.nodes.resize(out.nodes.size() + nInner);
            w.exImask = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
            out.nodes[job.node] = w;
        }
        jobs.clear();
        return rootNode;
    };

    for (uint32_t r : bottomRoots)
        if (!out.rootOf.count(r))
            out.rootOf[r] = emit(r, false);
    out.topRoot = emit(topRootRef, true);
    return out;
}

// Worst-case number of pending stack groups: a node visit leaves at most the group of its unvisited siblings
// behind; entering an instance leaves the sibling group, the remaining items and the leave-instance sentinel.
// `instRootNode[i]`: node index of the mesh root of instance i.
inline uint32_t wide8StackNeed(const Wide8& w, const std::vector<uint32_t>& instRootNode)
{
    std::vector<uint32_t> need(w.nodes.size(), 0u);
    // children are emitted after their parent and every bottom-level tree before the top level:
    // one reverse sweep over the bottom-level nodes, then one over the top-level nodes
    const size_t ranges[2][2] = { { 0, w.topRoot }, { w.topRoot, w.nodes.size() } };
    for (int pass = 0; pass < 2; pass++) {
        for (size_t i = ranges[pass][1]; i-- > ranges[pass][0];) {
            const Node8& n = w.nodes[i];
            uint32_t deepest = 0, inner = 0;
            for (int s = 0; s < 8; s++) {
                const uint32_t m = (n.meta[s >> 2] >> (8 * (s & 3))) & 0xFFu;
                if (m == 0u)
                    continue;
                if ((m & 0x18u) == 0x18u) {
                    deepest = std::max(deepest, 1u + need[n.childBase + inner]);
                    inner++;
                } else if (pass == 1) {
                    const uint32_t count = (uint32_t)__builtin_popcount(m >> 5);
                    for (uint32_t t = 0; t < count; t++) {
                        const size_t idx = (size_t)n.itemBase + (m & 0x1Fu) + t;
                        if (idx < w.items.size() && !(w.items[idx] & kItemBakedTriangle) && w.items[idx] < instRootNode.size())
                            deepest = std::max(deepest, 3u + need[instRootNode[w.items[idx]]]);
                    }
                }
            }
            need[i] = deepest;
        }
    }
    return 2u + need[w.topRoot];
}

} // namespace ptd
