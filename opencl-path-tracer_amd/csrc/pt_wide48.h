// 48-byte 4-wide node with implicit child addressing (host-side builder + node layout).
//
// k_trace is bound by the vector-memory instructions it issues for divergent deep-node fetches (DESIGN.md
// section 6).  The 64-byte WideNode costs four 16-byte loads, one of them only for the four explicit child
// references.  Here the children of a node are stored next to each other -- inner children consecutively in the
// node array from `nodeBase`, the triangles of leaf children consecutively in the triangle array from
// `triBase`, both in slot order -- so a 4-bit code per child (0 empty, 1..3 leaf with that many triangles,
// 8 inner) replaces the references and the node fits THREE loads.  The kernel rebuilds the references with a
// handful of integer instructions, which it has to spare.
//
// Usable when the whole scene is one world-space tree, i.e. every instance was copied to world space at upload
// (ptamd.hip); scenes that keep two-level instances need the 64-byte node and k_trace.
//
// STATUS: parity-green (all -m gpu tests pass with -DPT_NODE48=1) but NOT the default: 8.62-8.74 Grays/s in-kernel
// against 8.82-8.89 for the 64-byte node on the benchmark scene, also at a 64-byte stride (8.62-8.72), i.e. the
// fourth load was not what limited the kernel and the ~30 extra integer instructions per node visit cost more
// than it saved -- which, with the other experiments of DESIGN.md section 6, points at VALU issue as the limit.
#pragma once
#include "pt_wide8.h" // Kid, kidOf, validKid, triBounds, areaOf

namespace ptd {

struct Node48 { // three 16-byte chunks
    float ox, oy, oz; // origin of the quantisation grid = min corner of the union of the children
    uint32_t exCodes; // ex | ey << 8 | ez << 16 (biased float exponents of the grid step) | code0 << 24 | code1 << 28
    uint32_t qlox, qhix, qloy, qhiy; // byte k of each word belongs to child k
    uint32_t qloz, qhiz;
    uint32_t nodeBase; // bits 0..26: index of the first inner child, bits 28..31: code2
    uint32_t triBase; // bits 0..26: index of the first leaf triangle, bits 28..31: code3
};
static_assert(sizeof(Node48) == 48, "Node48 is three 16-byte chunks");
constexpr uint32_t kCodeInner = 8u;

struct Wide48 {
    bool usable = false; // false: the tree reaches an instance that was not copied to world space
    std::vector<Node48> nodes; // node 0 is the root
    std::vector<TriIsect> tris; // re-emitted in node order; c.y = original primitive, c.z = instance of the copy or ~0u
    uint32_t stackNeed = 0; // worst-case number of pending stack entries
};

// `pair`: the unified pair-node tree; `tris`: triangles its leaf references index (c.y / c.z set for world-space
// copies); `rootRef`: reference of the scene root; `rootLo/Hi`: its bounds (used when the root is a lone leaf).
inline Wide48 buildWide48(std::vector<PairNode> pair, const std::vector<TriIsect>& tris, uint32_t rootRef, const float rootLo[3], const float rootHi[3],
    uint32_t firstCopiedTri)
{
    using namespace wide8_detail;
    Wide48 out;
    // a leaf child holds at most 3 triangles (its code is the count): split larger leaves into synthetic pair nodes
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
    for (size_t i = 0; i < originalNodes; i++)
        for (int side = 0; side < 2; side++) {
            const uint32_t r = side ? pair[i].right : pair[i].left;
            if (r != kRefNone && refCount(r) > 3u && refCount(r) != kRefSpecial) {
                const uint32_t nr = splitLeaf(splitLeaf, refIndex(r), refCount(r));
                (side ? pair[i].right : pair[i].left) = nr;
            }
        }
    if (rootRef == kRefNone || refCount(rootRef) == kRefSpecial)
        return out;
    if (refCount(rootRef) > 3u)
        rootRef = splitLeaf(splitLeaf, refIndex(rootRef), refCount(rootRef));

    struct Job {
        uint32_t ref, node;
    };
    std::vector<Job> jobs { { rootRef, 0u } };
    out.nodes.emplace_back();
    for (size_t jq = 0; jq < jobs.size(); jq++) {
        const Job job = jobs[jq];
        // ---- up to four children: the two of the pair node, then the largest inner child is opened (surface-area greedy)
        Kid kids[4];
        int n = 0;
        if (refCount(job.ref) == 0u) {
            const PairNode& pn = pair[refIndex(job.ref)];
            for (int side = 0; side < 2; side++) {
                const Kid k = kidOf(pn, side);
                if (validKid(k))
                    kids[n++] = k;
            }
            while (n < 4) {
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
        } else { // the scene is a lone leaf: wrap it
            Kid k;
            for (int a = 0; a < 3; a++)
                k.lo[a] = rootLo[a], k.hi[a] = rootHi[a];
            k.ref = job.ref;
            kids[n++] = k;
        }
        for (int k = 0; k < n; k++)
            if (refCount(kids[k].ref) == kRefSpecial)
                return out; // an instance that is entered, not copied: this layout cannot express it
        // ---- grid
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (int k = 0; k < n; k++)
            for (int a = 0; a < 3; a++)
                lo[a] = std::min(lo[a], kids[k].lo[a]), hi[a] = std::max(hi[a], kids[k].hi[a]);
        Node48 w {};
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
        // ---- children in slot order: inner ones get consecutive nodes, leaf ones consecutive triangles
        const uint32_t nodeBase = (uint32_t)out.nodes.size(), triBase = (uint32_t)out.tris.size();
        uint32_t codes[4] = { 0, 0, 0, 0 }, nInner = 0;
        uint32_t* q[6] = { &w.qlox, &w.qloy, &w.qloz, &w.qhix, &w.qhiy, &w.qhiz };
        for (int k = 0; k < 4; k++) {
            if (k < n) {
                const uint32_t r = kids[k].ref;
                if (refCount(r) == 0u) {
                    codes[k] = kCodeInner;
                    jobs.push_back({ r, nodeBase + nInner });
                    nInner++;
                } else {
                    codes[k] = refCount(r); // 1..3
                    for (uint32_t t = 0; t < refCount(r); t++) {
                        const uint32_t src = refIndex(r) + t;
                        TriIsect tr = tris[src];
                        if (src < firstCopiedTri) {
                            tr.c.y = __builtin_bit_cast(float, src);
                            tr.c.z = __builtin_bit_cast(float, 0xFFFFFFFFu);
                        }
                        out.tris.push_back(tr);
                    }
                }
            }
            for (int a = 0; a < 3; a++) {
                uint32_t ql = 255, qh = 0; // empty slot: inverted box
                if (k < n) {
                    const float fl = std::floor((kids[k].lo[a] - lo[a]) / scale[a]);
                    const float fh = std::ceil((kids[k].hi[a] - lo[a]) / scale[a]);
                    ql = (uint32_t)std::max(0.f, std::min(255.f, fl));
                    qh = (uint32_t)std::max(0.f, std::min(255.f, fh));
                    while (ql > 0 && lo[a] + scale[a] * (float)ql > kids[k].lo[a])
                        ql--;
                    while (qh < 255 && lo[a] + scale[a] * (float)qh < kids[k].hi[a])
                        qh++;
                }
                *q[a] |= ql << (8 * k);
                *q[3 + a] |= qh << (8 * k);
            }
        }
        out.nodes.resize(out.nodes.size() + nInner);
        if (out.nodes.size() >= kRefIndexMask - 4u || out.tris.size() >= kRefIndexMask - 4u)
            return out;
        w.exCodes = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (codes[0] << 24) | (codes[1] << 28);
        w.nodeBase = nodeBase | (codes[2] << 28);
        w.triBase = triBase | (codes[3] << 28);
        out.nodes[job.node] = w;
    }
    // worst-case pending stack entries: a visit can leave all other children of the node behind
    std::vector<uint32_t> need(out.nodes.size(), 0u);
    for (size_t i = out.nodes.size(); i-- > 0;) { // children are emitted after their parent
        const Node48& nd = out.nodes[i];
        const uint32_t codes[4] = { (nd.exCodes >> 24) & 15u, nd.exCodes >> 28, nd.nodeBase >> 28, nd.triBase >> 28 };
        uint32_t kidsHere = 0, inner = 0, deepest = 0;
        for (int k = 0; k < 4; k++) {
            if (codes[k] == 0u)
                continue;
            kidsHere++;
            if (codes[k] == kCodeInner)
                deepest = std::max(deepest, need[(nd.nodeBase & kRefIndexMask) + inner++]);
        }
        need[i] = (kidsHere ? kidsHere - 1u : 0u) + deepest;
    }
    out.stackNeed = need[0];
    out.usable = true;
    return out;
}

} // namespace ptd
