// Scene conversion of the C-ABI: the caller's reference-layout arrays (include/ptamd.h) validated and turned into the HBM layouts of pt_device.h --
//   * the static part (pt_upload_static / _async, pt_update_geometry, pt_refit_vertices): pair nodes, the SAH-optimal collapse to 4-wide nodes, breadth-first
//     packing and quantisation, leaf cuts, triangle records, refit tables;
//   * the dynamic part (pt_upload_dynamic_async): the top level, the instance table and its routes (copied / folded / general / parked), lights, copy jobs.
// Host code only (no kernel launches except through the helpers it is handed).  Reference: RayTracer::initBuffersAndTransferStaticData / transferDynamicData,
// src/raytracer.cpp:201-287, 497-621.  Included by ptamd.hip (one translation unit).
#pragma once

namespace {

// Collapse the pair-node tree into 4-wide nodes (pt_device.h, WideNode): which descendants of pair node i become the (up to four)
// children of its wide node.  kids[i] describes the same subtree as pair[i], so child references keep their indices; the boxes are the
// exact ones (quantiseWideNode, pt_bake.h, makes the 8-bit planes; the world-space copies of instances are re-fitted from the exact boxes).
#ifndef PT_COLLAPSE_OPTIMAL
#define PT_COLLAPSE_OPTIMAL 1
#endif
struct WideKids {
    float lo[4][3], hi[4][3];
    uint32_t ref[4];
    uint32_t src[4]; // where the box of child k comes from: (pair node << 1) | side -- what a refit re-reads (refitStaticGeom)
    bool empty[4];
};
// Leaf formation inside the collapse (round 5).  The reference's builders stop at <= 3 triangles per leaf with Ct 1.5 / Ci 1.0 tuned for a binary tree
// (src/bvh/bvh_build.cpp:15-18); for THIS traversal a visit of a 4-wide node costs ~105 vector instructions and a triangle test ~35, and a leaf step runs
// to the longest leaf among its lanes.  So the collapse may turn a whole subtree into ONE leaf where that is cheaper:
//   asLeaf[n] = area(n) * (leaf0 + tri * (alpha * tris(n) + (1 - alpha) * cap))      (alpha 1: cost per triangle; alpha 0: every leaf visit costs the cap)
//   asRoot[n] = area(n) * inner + cheapest distribution of its (up to four) slots
// possible only where the subtree's triangle references are one contiguous run (the reference's builders emit leaves depth first: always) of <= cap.
// cap 0 = the leaves are given (rounds 1-4).  PTAMD_LEAF_FORMATION="cap[,inner,leaf0,tri,alpha]" overrides at run time (sweeps).
#ifndef PT_LEAF_CAP
#define PT_LEAF_CAP 0
#endif
struct CollapseCosts {
    uint32_t cap = PT_LEAF_CAP;
    double inner = 105.0, leaf0 = 20.0, tri = 35.0, alpha = 1.0;
    double leaf(uint32_t n) const { return leaf0 + tri * (alpha * (double)n + (1.0 - alpha) * (double)std::max(cap, 1u)); }
};
CollapseCosts collapseCostsFromEnv()
{
    CollapseCosts k;
    if (const char* e = getenv("PTAMD_LEAF_FORMATION")) {
        unsigned cap = k.cap;
        const int got = sscanf(e, "%u,%lf,%lf,%lf,%lf", &cap, &k.inner, &k.leaf0, &k.tri, &k.alpha);
        if (got >= 1)
            k.cap = std::min(cap, kMaxLeafTris);
    }
    return k;
}

std::vector<WideKids> collapseKids(const std::vector<PairNode>& pair, const CollapseCosts costs = CollapseCosts { 0u })
{
    std::vector<WideKids> out(pair.size());
    const char* seqEnv = getenv("PTAMD_BUILD_THREADS"); // (1: everything on the calling thread, as the host library's builders read it -- tests compare the two)
    const bool pooled = pair.size() >= 4096 && !(seqEnv && atoi(seqEnv) == 1);
    struct Child {
        float lo[3], hi[3];
        uint32_t ref;
        uint32_t src;
    };
    auto childOf = [&pair](const PairNode& n, int side) {
        Child c;
        c.src = ((uint32_t)(&n - pair.data()) << 1) | (uint32_t)side;
        const float* bx = &n.bx.x;
        const float* by = &n.by.x;
        const float* bz = &n.bz.x;
        c.lo[0] = bx[side * 2], c.hi[0] = bx[side * 2 + 1];
        c.lo[1] = by[side * 2], c.hi[1] = by[side * 2 + 1];
        c.lo[2] = bz[side * 2], c.hi[2] = bz[side * 2 + 1];
        c.ref = side ? n.right : n.left;
        return c;
    };
    auto area = [](const Child& c) {
        const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
        return dx >= 0.f && dy >= 0.f && dz >= 0.f ? dx * dy + dy * dz + dz * dx : -1.f;
    };
    // Which descendants become the (up to four) children of the wide node made from pair node i?  The cost of a wide tree is the
    // sum over its inner nodes of the chance a ray visits them ~ their surface area (the leaves are given).  Minimised exactly by
    // dynamic programming over the binary tree (as in Ylitie et al. 2017 for 8-wide trees):
    //   asRoot[n]   = area(n) + min over i of  atMost[left][i] + atMost[right][4 - i]          (n becomes a wide node)
    //   atMost[n][k] = cheapest way to hand subtree n to a parent that has k child slots for it:
    //                  n itself as one child (asRoot[n]), or split between its two children (i and k - i slots)
    // Round 1 opened the child of largest area until four were collected (surface-area greedy): 3 % more inner-node area on the
    // benchmark's meshes (17.67 vs 17.12 / 16.49 vs 16.02 root areas).
#if PT_COLLAPSE_OPTIMAL
    const size_t N = pair.size();
    auto isInner = [&](uint32_t r) { return r != kRefNone && refCount(r) == 0u && refIndex(r) < N; };
    struct Dp {
        double atMost[5]; // [1..4]
        uint8_t split[5]; // 0: the node itself, i: i slots to the left child
        uint8_t rootSplit, done;
        uint8_t asLeaf; // as ONE child the subtree is a leaf of [leafFirst, leafFirst + leafCount)
        uint32_t leafFirst, leafCount; // the subtree's triangle references, when they are one run of <= cap (leafCount 0: not)
    };
    const bool leafCosts = costs.cap > 0u; // the leaves are no longer given: they enter the cost
    auto childArea = [&](const PairNode& n, int side) {
        const Child c = childOf(n, side);
        const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
        return dx >= 0.0 && dy >= 0.0 && dz >= 0.0 ? dx * dy + dy * dz + dz * dx : 0.0;
    };
    std::vector<Dp> dp(N);
    for (Dp& d : dp)
        d.done = 0;
    auto nodeArea = [&](size_t n) { // box of pair node n = union of its two child boxes
        const Child a = childOf(pair[n], 0), b = childOf(pair[n], 1);
        double lo[3], hi[3];
        bool any = false;
        for (const Child* c : { &a, &b }) {
            if (!(c->lo[0] <= c->hi[0]) || c->ref == kRefNone)
                continue;
            for (int ax = 0; ax < 3; ax++) {
                lo[ax] = any ? std::min(lo[ax], (double)c->lo[ax]) : c->lo[ax];
                hi[ax] = any ? std::max(hi[ax], (double)c->hi[ax]) : c->hi[ax];
            }
            any = true;
        }
        if (!any)
            return 0.0;
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    {
        // post-order over the subtree below `root` (a stack of its own per caller: subtrees are disjoint, so several can be solved side by side)
        auto solve = [&](size_t root, std::vector<uint32_t>& stack) {
            if (dp[root].done)
                return;
            stack.push_back((uint32_t)root);
            while (!stack.empty()) {
                const uint32_t n = stack.back();
                if (dp[n].done == 2) {
                    stack.pop_back();
                    continue;
                }
                const uint32_t kids[2] = { pair[n].left, pair[n].right };
                if (dp[n].done == 0) { // first visit: children first (done = 1 marks 'on the stack': a cycle cannot loop forever)
                    dp[n].done = 1;
                    for (uint32_t r : kids)
                        if (isInner(r) && dp[refIndex(r)].done == 0)
                            stack.push_back(refIndex(r));
                    continue;
                }
                // children are final (or n sits on a cycle, which upload validation has already excluded): combine.  What either side costs with k slots,
                // looked up once: the child's own table, or one number for every k (a given leaf: what a visit of it costs; nothing where the leaves are given)
                double flat[2][5];
                const double* cost[2];
                for (int side = 0; side < 2; side++) {
                    const uint32_t r = kids[side];
                    if (isInner(r) && dp[refIndex(r)].done == 2) {
                        cost[side] = dp[refIndex(r)].atMost;
                        continue;
                    }
                    const double v = leafCosts && r != kRefNone && refCount(r) >= 1u && refCount(r) <= kMaxLeafTris ? costs.leaf(refCount(r)) * childArea(pair[n], side) : 0.0;
                    for (int k = 1; k <= 4; k++)
                        flat[side][k] = v;
                    cost[side] = flat[side];
                }
                auto costSide = [&](int side, int k) { return cost[side][k]; };
                Dp& d = dp[n];
                // the subtree's triangle references as one run?
                d.asLeaf = 0, d.leafFirst = 0, d.leafCount = 0;
                if (leafCosts) {
                    uint32_t first[2] = { 0, 0 }, cnt[2] = { 0, 0 };
                    for (int side = 0; side < 2; side++) {
                        const uint32_t r = kids[side];
                        if (isInner(r) && dp[refIndex(r)].done == 2)
                            first[side] = dp[refIndex(r)].leafFirst, cnt[side] = dp[refIndex(r)].leafCount;
                        else if (r != kRefNone && refCount(r) >= 1u && refCount(r) <= kMaxLeafTris)
                            first[side] = refIndex(r), cnt[side] = refCount(r);
                    }
                    if (cnt[0] && cnt[1] && cnt[0] + cnt[1] <= costs.cap && (first[0] + cnt[0] == first[1] || first[1] + cnt[1] == first[0]))
                        d.leafFirst = std::min(first[0], first[1]), d.leafCount = cnt[0] + cnt[1];
                }
                double best = 1e300;
                for (int i = 1; i <= 3; i++) {
                    const double v = costSide(0, i) + costSide(1, 4 - i);
                    if (v < best)
                        best = v, d.rootSplit = (uint8_t)i;
                }
                const double ownArea = nodeArea(n);
                d.atMost[1] = (leafCosts ? costs.inner : 1.0) * ownArea + best;
                if (d.leafCount) {
                    const double asLeaf = costs.leaf(d.leafCount) * ownArea;
                    if (asLeaf < d.atMost[1])
                        d.atMost[1] = asLeaf, d.asLeaf = 1;
                }
                d.split[1] = 0;
                for (int k = 2; k <= 4; k++) {
                    d.atMost[k] = d.atMost[1];
                    d.split[k] = 0;
                    for (int i = 1; i < k; i++) {
                        const double v = costSide(0, i) + costSide(1, k - i);
                        if (v < d.atMost[k])
                            d.atMost[k] = v, d.split[k] = (uint8_t)i;
                    }
                }
                d.done = 2;
                stack.pop_back();
            }
        };
        // A rebuilt tree per frame: the subtrees five levels below the roots of large trees are solved on the host library's worker pool, the tops on the
        // calling thread afterwards.  The recurrence has one solution per node whatever the order: the same tree, byte for byte.  (Built once before, on the
        // pool of four threads that parked between loops, and reverted: the NEXT Mesh build on that pool paid 0.9 ms for the 0.2 this saved -- EXPERIMENTS.md.
        // The pool of eight that polls before it parks does not show that.)
        std::vector<uint32_t> tasks;
        if (pooled) {
            std::vector<uint8_t> isChild(N, 0);
            for (size_t n = 0; n < N; n++)
                for (uint32_t r : { pair[n].left, pair[n].right })
                    if (isInner(r))
                        isChild[refIndex(r)] = 1;
            std::vector<uint32_t> level, next;
            for (size_t n = 0; n < N; n++)
                if (!isChild[n])
                    level.push_back((uint32_t)n);
            for (int depth = 0; depth < 5 && !level.empty() && level.size() < 64; depth++) {
                next.clear();
                for (uint32_t n : level)
                    for (uint32_t r : { pair[n].left, pair[n].right })
                        if (isInner(r) && refIndex(r) != n)
                            next.push_back(refIndex(r));
                level.swap(next);
            }
            std::sort(level.begin(), level.end());
            level.erase(std::unique(level.begin(), level.end()), level.end()); // (a shared subtree -- refused by the upload's validation anyway -- is solved once)
            tasks = level;
        }
        if (tasks.size() >= 2) {
            std::atomic<size_t> nextTask { 0 };
            raytracer::WorkerPool& pool = raytracer::WorkerPool::get();
            pool.parallelFor(pool.threads(), 1, [&](size_t, size_t) {
                std::vector<uint32_t> stack;
                for (size_t t; (t = nextTask.fetch_add(1)) < tasks.size();)
                    solve(tasks[t], stack);
            });
        }
        std::vector<uint32_t> stack;
        for (size_t root = 0; root < N; root++)
            solve(root, stack);
    }
#endif
    raytracer::WorkerPool::get().parallelFor(pair.size(), pooled ? 2048 : pair.size() + 1, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
        Child kids[4];
        int n = 0;
#if PT_COLLAPSE_OPTIMAL
        {
            struct Item {
                Child c;
                int slots;
            };
            Item todo[8];
            int nt = 0;
            const int ls = dp[i].rootSplit;
            todo[nt++] = { childOf(pair[i], 1), 4 - ls }; // right first: the stack pops the left one first, slot order = tree order
            todo[nt++] = { childOf(pair[i], 0), ls };
            while (nt > 0) {
                const Item it = todo[--nt];
                const uint32_t r = it.c.ref;
                const int sp = isInner(r) && refIndex(r) != i ? dp[refIndex(r)].split[it.slots] : 0;
                if (sp == 0) {
                    kids[n] = it.c;
                    if (isInner(r) && refIndex(r) != i && dp[refIndex(r)].asLeaf) // the whole subtree as ONE leaf: same box, its run of triangles
                        kids[n].ref = makeRef(dp[refIndex(r)].leafFirst, dp[refIndex(r)].leafCount);
                    n++;
                    continue;
                }
                const PairNode& g = pair[refIndex(r)];
                todo[nt++] = { childOf(g, 1), it.slots - sp };
                todo[nt++] = { childOf(g, 0), sp };
            }
        }
#else
        // the two children of the binary node, then (surface-area greedy) the largest inner child is replaced
        // by its own two children until four are collected: the expensive-to-miss boxes are the ones opened up
        kids[n++] = childOf(pair[i], 0);
        kids[n++] = childOf(pair[i], 1);
        while (n < 4) {
            int best = -1;
            float bestArea = -1.f;
            for (int k = 0; k < n; k++) {
                const uint32_t r = kids[k].ref;
                if (r != kRefNone && refCount(r) == 0u && refIndex(r) < pair.size() && refIndex(r) != i && area(kids[k]) > bestArea) {
                    best = k;
                    bestArea = area(kids[k]);
                }
            }
            if (best < 0)
                break;
            const PairNode& g = pair[refIndex(kids[best].ref)];
            kids[best] = childOf(g, 0);
            kids[n++] = childOf(g, 1);
        }
#endif
        WideKids wk {};
        for (int k = 0; k < 4; k++) {
            wk.empty[k] = k >= n || !(kids[k].lo[0] <= kids[k].hi[0]) || kids[k].ref == kRefNone;
            wk.ref[k] = wk.empty[k] ? kRefNone : kids[k].ref;
            wk.src[k] = k < n ? kids[k].src : 0u;
            for (int a = 0; a < 3; a++) {
                wk.lo[k][a] = wk.empty[k] ? 1.f : kids[k].lo[a];
                wk.hi[k][a] = wk.empty[k] ? -1.f : kids[k].hi[a];
            }
        }
        out[i] = wk;
    }
    });
    return out;
}

// world = inverse(invTransform) by Gauss-Jordan in double; m is column-major (TopBvhNode::invTransform).
// On success w[r][4 + c] holds element (r, c) of the world transform.
bool invertTransform(const float* m, double w[4][8])
{
    for (int r = 0; r < 4; r++)
        for (int col = 0; col < 4; col++) {
            w[r][col] = m[col * 4 + r];
            w[r][col + 4] = (r == col) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(w[r][col]) > std::fabs(w[piv][col]))
                piv = r;
        if (std::fabs(w[piv][col]) < 1e-300)
            return false;
        for (int k = 0; k < 8; k++)
            std::swap(w[piv][k], w[col][k]);
        const double dv = w[col][col];
        for (int k = 0; k < 8; k++)
            w[col][k] /= dv;
        for (int r = 0; r < 4; r++)
            if (r != col) {
                const double f = w[r][col];
                for (int k = 0; k < 8; k++)
                    w[r][k] -= f * w[col][k];
            }
    }
    return true;
}

// ---- the static part of a scene: bottom-level trees, converted once per pt_upload_static / pt_update_geometry -----------------------
// Collapse every mesh tree to 4-wide nodes and pack them breadth-first root by root: the four children of a node get neighbouring
// slots (half the footprint in the 4 MB-per-XCD L2, siblings share 128-byte lines) and a mesh's nodes are ONE contiguous run, which is
// what a world-space copy of an instance (pt_bake.h) is made from.  Roots are the caller's nodes no other node refers to, plus any
// node a top-level leaf has ever named (`extraRoots`).
// pair-node boxes of a refit: the caller's refitted boxes for the pairs that mirror its inner nodes (`onlyExtra`: skipped) and, for the pairs that
// split a leaf of more than kMaxLeafTris triangles (appended children first), the bounds of their triangles
void refitPairBoxes(pt_ctx* c, const pt_vertex* verts, const pt_sub_bvh_node* nodes, bool onlyExtra)
{
    std::vector<PairNode>& pair = c->st->hostBottomNodes;
    if (!onlyExtra)
        for (uint32_t i = 0; i < c->st->numRefNodes; i++) {
            const uint32_t d = c->st->denseOfNode[i];
            if (d == 0xFFFFFFFFu)
                continue;
            const uint32_t l = nodes[i].leftChildOrFirstTriangle;
            const pt_sub_bvh_node &L = nodes[l], &R = nodes[l + 1];
            pair[d].bx = make_float4(L.min[0], L.max[0], R.min[0], R.max[0]);
            pair[d].by = make_float4(L.min[1], L.max[1], R.min[1], R.max[1]);
            pair[d].bz = make_float4(L.min[2], L.max[2], R.min[2], R.max[2]);
        }
    if (pair.size() <= c->st->numDensePairs)
        return;
    auto boxOf = [&](uint32_t ref, V3& lo, V3& hi) {
        lo = mk(FLT_MAX), hi = mk(-FLT_MAX);
        if (refCount(ref) != 0u) {
            for (uint32_t t = refIndex(ref); t < refIndex(ref) + refCount(ref); t++) {
                const TriShade& ts = c->st->hostTriShade[t];
                for (uint32_t vi : { ts.i0, ts.i1, ts.i2 }) {
                    const V3 p = mk(verts[vi].vertex[0], verts[vi].vertex[1], verts[vi].vertex[2]);
                    lo = mk(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
                    hi = mk(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
                }
            }
        } else {
            const PairNode& n = pair[refIndex(ref)];
            lo = mk(fminf(n.bx.x, n.bx.z), fminf(n.by.x, n.by.z), fminf(n.bz.x, n.bz.z));
            hi = mk(fmaxf(n.bx.y, n.bx.w), fmaxf(n.by.y, n.by.w), fmaxf(n.bz.y, n.bz.w));
        }
    };
    for (size_t j = c->st->numDensePairs; j < pair.size(); j++) {
        V3 llo, lhi, rlo, rhi;
        boxOf(pair[j].left, llo, lhi);
        boxOf(pair[j].right, rlo, rhi);
        pair[j].bx = make_float4(llo.x, lhi.x, rlo.x, rhi.x);
        pair[j].by = make_float4(llo.y, lhi.y, rlo.y, rhi.y);
        pair[j].bz = make_float4(llo.z, lhi.z, rlo.z, rhi.z);
    }
}

// the packed 4-wide nodes of a refit on the host: same children in the same slots, new boxes (what k_refit_nodes does on the device)
void refitWideOnHost(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    const std::vector<PairNode>& pair = c->st->hostBottomNodes;
    for (size_t q = 0; q < g.wide.size(); q++) {
        float lo[4][3], hi[4][3];
        uint32_t refs[4];
        bool empty[4];
        for (int k = 0; k < 4; k++) {
            empty[k] = g.kidEmpty[q * 4 + k] != 0u;
            refs[k] = g.wide[q].child[k];
            if (empty[k]) {
                for (int a = 0; a < 3; a++)
                    lo[k][a] = 1.f, hi[k][a] = -1.f;
                continue;
            }
            const uint32_t src = g.kidSrc[q * 4 + k];
            const PairNode& n = pair[src >> 1];
            const int side = (int)(src & 1u);
            const float *bx = &n.bx.x, *by = &n.by.x, *bz = &n.bz.x;
            lo[k][0] = bx[side * 2], hi[k][0] = bx[side * 2 + 1];
            lo[k][1] = by[side * 2], hi[k][1] = by[side * 2 + 1];
            lo[k][2] = bz[side * 2], hi[k][2] = bz[side * 2 + 1];
        }
        for (int k = 0; k < 4; k++)
            for (int a = 0; a < 3; a++)
                g.boxes[q].lo[k][a] = lo[k][a], g.boxes[q].hi[k][a] = hi[k][a];
        quantiseWideNode(lo, hi, refs, empty, g.emptyRef, &g.wide[q]);
    }
}

// the caller's latest vertices / nodes: in the pinned staging memory after a device-side refit, in the host vectors otherwise
inline const pt_vertex* latestVerts(const pt_ctx* c) { return c->st->sg.latestInStage ? (const pt_vertex*)c->st->sg.stage : c->st->rawVerts.data(); }
inline const pt_sub_bvh_node* latestNodes(const pt_ctx* c)
{
    return c->st->sg.latestInStage ? (const pt_sub_bvh_node*)((const unsigned char*)c->st->sg.stage + (size_t)c->st->numVerts * sizeof(pt_vertex)) : c->st->hostSubNodes.data();
}

// The host's mirrors from the caller's arrays as last handed in (a refit re-makes the device's records on the device and leaves these behind): pair-node
// boxes, the packed nodes, hostTris / hostVerts.
// the caller's node boxes recomputed from the latest vertices (what refitBVH leaves, reference src/bvh/refit_bvh.cpp:6-34): after a refit on
// the device alone nobody handed refitted nodes in.  Children lie after their parent (validated at upload): one reverse sweep.
void refitHostNodeBoxes(pt_ctx* c)
{
    const pt_vertex* verts = c->st->rawVerts.data();
    std::vector<pt_sub_bvh_node>& nodes = c->st->hostSubNodes;
    for (size_t i = nodes.size(); i-- > 0;) {
        pt_sub_bvh_node& n = nodes[i];
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        if (n.triangleCount != 0) {
            for (uint32_t t = n.leftChildOrFirstTriangle; t < n.leftChildOrFirstTriangle + n.triangleCount; t++) {
                const TriShade& ts = c->st->hostTriShade[t];
                for (uint32_t vi : { ts.i0, ts.i1, ts.i2 })
                    for (int a = 0; a < 3; a++)
                        lo[a] = fminf(lo[a], verts[vi].vertex[a]), hi[a] = fmaxf(hi[a], verts[vi].vertex[a]);
            }
        } else {
            if (c->st->denseOfNode[i] == 0xFFFFFFFFu)
                continue; // an unused pad
            const pt_sub_bvh_node &L = nodes[n.leftChildOrFirstTriangle], &R = nodes[n.leftChildOrFirstTriangle + 1];
            for (int a = 0; a < 3; a++)
                lo[a] = fminf(L.min[a], R.min[a]), hi[a] = fmaxf(L.max[a], R.max[a]);
        }
        for (int a = 0; a < 3; a++)
            n.min[a] = lo[a], n.max[a] = hi[a];
    }
    c->st->hostNodeBoxesStale = false;
}

void refreshHostGeometry(pt_ctx* c)
{
    if (!c->st->hostGeomStale)
        return;
    if (c->st->hostNodeBoxesStale) { // (rawVerts is current then: pt_refit_vertices keeps it so)
        refitHostNodeBoxes(c);
        refitPairBoxes(c, c->st->rawVerts.data(), c->st->hostSubNodes.data(), false);
        if (c->st->sg.wide.size() == c->st->sg.kidEmpty.size() / 4)
            refitWideOnHost(c);
    }
    const pt_vertex* verts = latestVerts(c);
    if (c->st->sg.latestInStage) {
        refitPairBoxes(c, verts, latestNodes(c), false);
        if (c->st->sg.wide.size() == c->st->sg.kidEmpty.size() / 4)
            refitWideOnHost(c);
    }
    auto P = [&](uint32_t vi) { return mk(verts[vi].vertex[0], verts[vi].vertex[1], verts[vi].vertex[2]); };
    for (size_t t = 0; t < c->st->hostTriShade.size(); t++) {
        const TriShade& ts = c->st->hostTriShade[t];
        const V3 v0 = P(ts.i0);
        const V3 e1 = P(ts.i1) - v0, e2 = P(ts.i2) - v0; // shapes.cl:37-38
        c->st->hostTris[t].a = make_float4(v0.x, v0.y, v0.z, e1.x);
        c->st->hostTris[t].b = make_float4(e1.y, e1.z, e2.x, e2.y);
        c->st->hostTris[t].c = make_float4(e2.z, 0.f, 0.f, 0.f);
    }
    for (size_t v = 0; v < c->st->numVerts; v++) {
        c->st->hostVerts[v].n_u = make_float4(verts[v].normal[0], verts[v].normal[1], verts[v].normal[2], verts[v].texCoord[0]);
        c->st->hostVerts[v].v_pad = make_float4(verts[v].texCoord[1], 0.f, 0.f, 0.f);
    }
    if (c->st->sg.latestInStage) { // the host vectors take the latest arrays over (the staging memory is rewritten by the next refit)
        c->st->rawVerts.assign(verts, verts + c->st->numVerts);
        const pt_sub_bvh_node* nodes = latestNodes(c);
        c->st->hostSubNodes.assign(nodes, nodes + c->st->numRefNodes);
        c->st->sg.latestInStage = false;
    }
    c->st->hostGeomStale = false;
}

// shading records of the caller's triangles: one 128-byte line per triangle (TriFat, pt_device.h)
void buildFat(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    g.fat.resize(c->st->hostTriShade.size());
    for (size_t t = 0; t < c->st->hostTriShade.size(); t++) {
        const TriShade& ts = c->st->hostTriShade[t];
        const VertexShade &a0 = c->st->hostVerts[ts.i0], &a1 = c->st->hostVerts[ts.i1], &a2 = c->st->hostVerts[ts.i2];
        const TriIsect& ti = c->st->hostTris[t];
        TriFat f {};
        f.n0u = a0.n_u, f.n1u = a1.n_u, f.n2u = a2.n_u;
        float mbits;
        std::memcpy(&mbits, &ts.material, 4);
        f.vvvm = make_float4(a0.v_pad.x, a1.v_pad.x, a2.v_pad.x, mbits);
        f.e1e = make_float4(ti.a.w, ti.b.x, ti.b.y, ti.b.z); // edge1.xyz, edge2.x
        f.e2v = make_float4(ti.b.w, ti.c.x, ti.a.x, ti.a.y); // edge2.yz, v0.xy
        float m[12]; // the caller's 48-byte material record: colour (16 B), parameters (16 B), type (+ padding)
        std::memcpy(m, &c->st->hostMaterials[ts.material], sizeof m);
        f.v0c = make_float4(ti.a.z, m[0], m[1], m[2]);
        f.mat = make_float4(m[4], m[5], m[6], m[8]);
        g.fat[t] = f;
    }
}

// PTAMD_UPLOAD_TIMING=1: host time of the stages of an upload, one line per call on stderr (diagnostics; tools/r5_upload_timing.sh)
struct StageTimer {
    const char* what;
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    std::string line;
    explicit StageTimer(const char* w)
        : what(w)
        , on(getenv("PTAMD_UPLOAD_TIMING") != nullptr)
    {
        if (on)
            t0 = last = std::chrono::steady_clock::now();
    }
    void lap(const char* name)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof buf, " %s %.3f", name, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    ~StageTimer()
    {
        if (on)
            fprintf(stderr, "[ptamd] %s: total %.3f ms;%s\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), line.c_str());
    }
};

int buildStaticGeom(pt_ctx* c, bool deviceMakesRecords = false)
{
    StaticScene::StaticGeom& g = c->st->sg;
    StageTimer tm("buildStaticGeom");
    g.deviceMakesRecords = deviceMakesRecords;
    refreshHostGeometry(c);
    tm.lap("refreshHostGeometry");
    const uint32_t nN = c->st->numRefNodes, nT = c->st->numTris;
    const std::vector<WideKids> kids = collapseKids(c->st->hostBottomNodes, collapseCostsFromEnv());
    tm.lap("collapseKids");
    const uint32_t emptyRef = makeRef(nT, 1u); // the all-zero triangle stored right after the caller's triangles (det == 0: never hit)
    std::vector<uint8_t> isChild(nN, 0);
    for (uint32_t i = 0; i < nN; i++) {
        const pt_sub_bvh_node& n = c->st->hostSubNodes[i];
        const uint32_t l = n.leftChildOrFirstTriangle;
        if (n.triangleCount == 0 && c->st->nodeRef[i] != kRefNone && (uint64_t)l + 1 < nN)
            isChild[l] = isChild[l + 1] = 1;
    }
    std::vector<uint32_t> rootNodes;
    for (uint32_t i = 0; i < nN; i++)
        if (c->st->nodeRef[i] != kRefNone && (!isChild[i] || std::find(g.extraRoots.begin(), g.extraRoots.end(), i) != g.extraRoots.end()))
            rootNodes.push_back(i);
    g.wide.clear(), g.leafOfs.clear(), g.refTri.clear(), g.roots.clear(), g.kidSrc.clear(), g.kidEmpty.clear(), g.kidBoxNode.clear();
    std::vector<uint32_t> pairLeft(c->st->numDensePairs, 0u); // pair node -> the caller's node that is its left child
    for (uint32_t i = 0; i < nN; i++)
        if (c->st->denseOfNode[i] != 0xFFFFFFFFu)
            pairLeft[c->st->denseOfNode[i]] = c->st->hostSubNodes[i].leftChildOrFirstTriangle;
    g.rootOfNode.assign(nN, -1);
    tm.lap("roots");
    constexpr uint32_t kUnset = 0xFFFFFFFFu;
    std::vector<uint32_t> newIndex(kids.size(), kUnset), order;
    auto isInner = [&](uint32_t r) { return r != kRefNone && refCount(r) == 0u && refIndex(r) < kids.size(); };
    for (uint32_t rn : rootNodes) {
        StaticScene::StaticGeom::Root root {};
        const uint32_t rref = c->st->nodeRef[rn];
        root.nodeBase = (uint32_t)order.size();
        root.refBase = (uint32_t)g.refTri.size();
        root.bakeable = true;
        if (isInner(rref)) {
            if (newIndex[refIndex(rref)] != kUnset) { // reachable from an earlier root too (a top-level leaf names an interior node): shares its run
                root.ref = makeRef(newIndex[refIndex(rref)], 0u);
                root.bakeable = false;
            } else {
                const size_t first = order.size();
                newIndex[refIndex(rref)] = (uint32_t)order.size();
                order.push_back(refIndex(rref));
                for (size_t q = first; q < order.size(); q++) // breadth first
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r = kids[order[q]].ref[k];
                        if (kids[order[q]].empty[k] || !isInner(r))
                            continue;
                        if (newIndex[refIndex(r)] != kUnset) {
                            root.bakeable = false; // shares nodes with another tree: not one run
                            continue;
                        }
                        newIndex[refIndex(r)] = (uint32_t)order.size();
                        order.push_back(refIndex(r));
                    }
                root.ref = makeRef(root.nodeBase, 0u);
            }
            root.numNodes = (uint32_t)order.size() - root.nodeBase;
        } else {
            root.ref = rref; // the mesh is a single leaf
            for (uint32_t k = 0; k < refCount(rref); k++)
                g.refTri.push_back(refIndex(rref) + k);
        }
        // nodes of this run: remapped references, exact boxes, triangle-reference offsets of the leaves
        g.wide.resize(order.size());
        g.boxes.resize(order.size()); // (every slot is written below -- or, where the device makes the records, never read)
        g.leafOfs.resize(order.size() * 4, 0u);
        g.kidSrc.resize(order.size() * 4, 0u);
        g.kidEmpty.resize(order.size() * 4, 1u);
        g.kidBoxNode.resize(order.size() * 4, 0xFFFFFFFFu);
        for (size_t q = root.nodeBase; q < order.size(); q++) { // the leaves' runs in the table of triangle references: in node order, one after the other
            const WideKids& wk = kids[order[q]];
            for (int k = 0; k < 4; k++)
                if (!wk.empty[k] && !isInner(wk.ref[k])) {
                    g.leafOfs[q * 4 + k] = (uint32_t)g.refTri.size() - root.refBase;
                    for (uint32_t t = 0; t < refCount(wk.ref[k]); t++)
                        g.refTri.push_back(refIndex(wk.ref[k]) + t);
                }
        }
        // ... everything else per node on its own (the quantiser is most of a conversion's time): the host library's worker threads take ranges of them
        raytracer::WorkerPool::get().parallelFor(order.size() - root.nodeBase, 512, [&](size_t q0, size_t q1) {
            for (size_t q = root.nodeBase + q0; q < root.nodeBase + q1; q++) {
                const WideKids& wk = kids[order[q]];
                uint32_t refs[4];
                for (int k = 0; k < 4; k++) {
                    g.kidSrc[q * 4 + k] = wk.src[k], g.kidEmpty[q * 4 + k] = wk.empty[k] ? 1u : 0u;
                    if (!wk.empty[k]) { // the caller's node whose box this slot takes: the left / right child of the node its pair mirrors
                        const uint32_t pr = wk.src[k] >> 1, side = wk.src[k] & 1u;
                        g.kidBoxNode[q * 4 + k] = pr < c->st->numDensePairs ? pairLeft[pr] + side : (0x80000000u | ((pr - c->st->numDensePairs) * 2u + side));
                    }
                    refs[k] = wk.empty[k] ? emptyRef : (isInner(wk.ref[k]) ? makeRef(newIndex[refIndex(wk.ref[k])], 0u) : wk.ref[k]);
                    if (!deviceMakesRecords)
                        for (int a = 0; a < 3; a++)
                            g.boxes[q].lo[k][a] = wk.lo[k][a], g.boxes[q].hi[k][a] = wk.hi[k][a];
                }
                if (deviceMakesRecords) { // the references only: k_refit_nodes gathers the boxes and makes the planes (uploadStaticGeom)
                    WideNode w {};
                    for (int k = 0; k < 4; k++)
                        w.child[k] = refs[k];
                    g.wide[q] = w;
                } else {
                    quantiseWideNode(wk.lo, wk.hi, refs, wk.empty, emptyRef, &g.wide[q]);
                }
            }
        });
        root.numRefs = (uint32_t)g.refTri.size() - root.refBase;
        g.rootOfNode[rn] = (int32_t)g.roots.size();
        g.roots.push_back(root);
    }
    // worst-case number of pending stack entries below every packed node: visiting a node can leave all its other children on the
    // stack (children are visited nearest first, so any order can occur: the bound takes the deepest child first).  Inside a run the
    // children sit after their parent; a child that lies in ANOTHER run (a top-level leaf named an interior node, whose subtree an
    // earlier root had packed already) lies in an earlier one.  So: run by run in ascending order, each run in reverse -- every child
    // is final when its parent is reached.  (One reverse sweep over everything took 0 for the shared children: too small a bound.)
    tm.lap("pack");
    g.stackNeed.assign(g.wide.size(), 0u);
    for (const StaticScene::StaticGeom::Root& root : g.roots)
        for (size_t q = (size_t)root.nodeBase + root.numNodes; q-- > root.nodeBase;) {
            if (refCount(root.ref) != 0u || refIndex(root.ref) != root.nodeBase)
                break; // no run of its own (a single leaf, or the root sits inside an earlier run)
            uint32_t n = 0, deepest = 0;
            for (uint32_t r : g.wide[q].child) {
                if (r == emptyRef)
                    continue;
                n++;
                if (refCount(r) == 0u && refIndex(r) < g.wide.size())
                    deepest = std::max(deepest, g.stackNeed[refIndex(r)]);
            }
            g.stackNeed[q] = (n > 0 ? n - 1 : 0u) + deepest;
        }
    if (getenv("PTAMD_COLLAPSE_REPORT")) { // what the collapse made: packed nodes, leaves by size (tools / sweeps of PTAMD_LEAF_FORMATION)
        uint64_t hist[kMaxLeafTris + 1] = {}, leaves = 0, refs = 0, used = 0;
        for (const WideNode& w : g.wide)
            for (uint32_t r : w.child)
                if (r != emptyRef) {
                    used++;
                    if (refCount(r) >= 1u && refCount(r) <= kMaxLeafTris)
                        hist[refCount(r)]++, leaves++, refs += refCount(r);
                }
        const CollapseCosts k = collapseCostsFromEnv();
        fprintf(stderr, "[ptamd] collapse: cap %u costs %.0f/%.0f/%.0f alpha %.2f -> %zu wide nodes, %.2f used slots per node, %llu leaves, %.2f triangles per leaf; by size:", k.cap, k.inner,
            k.leaf0, k.tri, k.alpha, g.wide.size(), g.wide.empty() ? 0.0 : (double)used / (double)g.wide.size(), (unsigned long long)leaves, leaves ? (double)refs / (double)leaves : 0.0);
        for (uint32_t n = 1; n <= kMaxLeafTris; n++)
            if (hist[n])
                fprintf(stderr, " %u:%llu", n, (unsigned long long)hist[n]);
        fprintf(stderr, "\n");
    }
    tm.lap("stackNeed");
    if (deviceMakesRecords)
        g.fat.resize(c->st->hostTriShade.size()); // (its size is what the dynamic sets allocate by)
    else
        buildFat(c);
    tm.lap("buildFat");
    if (getenv("PTAMD_TREE_HASH") && !deviceMakesRecords) { // what a conversion made, in one line (tools/tree_hash.py: a change to the conversion's code that is meant to keep its result)
        auto fnv = [](const void* p, size_t n) {
            uint64_t h = 1469598103934665603ull;
            for (size_t i = 0; i < n; i++)
                h = (h ^ ((const uint8_t*)p)[i]) * 1099511628211ull;
            return (unsigned long long)h;
        };
        fprintf(stderr, "[ptamd] tree hash: wide %zu %016llx boxes %016llx leafOfs %016llx refTri %zu %016llx kidSrc %016llx kidBoxNode %016llx stackNeed %016llx fat %016llx pairs %zu %016llx\n",
            g.wide.size(), fnv(g.wide.data(), g.wide.size() * sizeof(WideNode)), fnv(g.boxes.data(), g.boxes.size() * sizeof(g.boxes[0])),
            fnv(g.leafOfs.data(), g.leafOfs.size() * 4), g.refTri.size(), fnv(g.refTri.data(), g.refTri.size() * 4), fnv(g.kidSrc.data(), g.kidSrc.size() * 4),
            fnv(g.kidBoxNode.data(), g.kidBoxNode.size() * 4), fnv(g.stackNeed.data(), g.stackNeed.size() * 4), fnv(g.fat.data(), g.fat.size() * sizeof(g.fat[0])),
            c->st->hostBottomNodes.size(), fnv(c->st->hostBottomNodes.data(), c->st->hostBottomNodes.size() * sizeof(PairNode)));
    }
    g.emptyRef = emptyRef;
    g.version = ++c->staticVersions;
    g.topology = g.version;
    g.onDevice = false;
    return PT_OK;
}

// the static arrays' master copy in device memory (the two dynamic sets take theirs from it, device to device)
int uploadStaticGeom(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    if (g.onDevice)
        return PT_OK;
    StageTimer tm("uploadStaticGeom");
    // the copy stream may still be reading the old master (a set being refreshed from it)
    HIPCHK(c, hipStreamSynchronize(c->copyStream));
    tm.lap("syncCopyStream");
    if (c->st->hostGeomStale) { // refitted before the master copy ever reached the device
        refreshHostGeometry(c);
        buildFat(c);
    }
    int rc;
    if (g.deviceMakesRecords) {
        // the topology and the caller's arrays as they are; the records on the device (what pt_update_geometry does after a refit: the same kernels from the same
        // boxes with the same routines -- a context that adopts this scene holds the bytes a fresh one makes on the host, tests/test_gpu_dynamic.py)
        std::vector<float> extra; // boxes of the pair nodes that cut a leaf of more than two triangles (k_refit_nodes: entry (pair - numDensePairs) * 2 + side)
        for (size_t j = c->st->numDensePairs; j < c->st->hostBottomNodes.size(); j++) {
            const PairNode& n = c->st->hostBottomNodes[j];
            const float b[12] = { n.bx.x, n.by.x, n.bz.x, n.bx.y, n.by.y, n.bz.y, n.bx.z, n.by.z, n.bz.z, n.bx.w, n.by.w, n.bz.w };
            extra.insert(extra.end(), b, b + 12);
        }
        const size_t nT = c->st->numTris;
        if ((rc = uploadVec(c, g.dWide, g.wide)) || (rc = uploadVec(c, g.dLeafOfs, g.leafOfs)) || (rc = uploadVec(c, g.dRefTri, g.refTri))
            || (rc = uploadVec(c, g.dVerts, c->st->rawVerts)) || (rc = uploadVec(c, c->st->triShade, c->st->hostTriShade))
            || (rc = uploadVec(c, g.dNodes, c->st->hostSubNodes)) || (rc = uploadVec(c, g.dKidBoxNode, g.kidBoxNode)) || (rc = uploadVec(c, g.dExtra, extra)))
            return rc;
        if (g.dBoxes.n < std::max<size_t>(g.wide.size(), 1))
            HIPCHK(c, g.dBoxes.alloc(g.wide.size() + g.wide.size() / 8 + 1));
        if (g.dTris.n < nT + 1)
            HIPCHK(c, g.dTris.alloc(nT + nT / 8 + 1));
        if (g.dFat.n < std::max<size_t>(nT, 1))
            HIPCHK(c, g.dFat.alloc(nT + nT / 8 + 1));
        tm.lap("uploads");
        static_assert(sizeof(VertexIn) == sizeof(pt_vertex) && sizeof(SubNodeIn) == sizeof(pt_sub_bvh_node), "the refit kernels read the caller's records as they are");
        HIPCHK(c, hipMemsetAsync(g.dTris.p + nT, 0, sizeof(TriIsect), c->copyStream)); // what an unused child slot refers to
        if (!g.wide.empty()) {
            RefitNodeArgs rn {};
            rn.nodes = (const SubNodeIn*)g.dNodes.p, rn.kidBoxNode = g.dKidBoxNode.p, rn.extra = g.dExtra.p, rn.wide = g.dWide.p, rn.boxes = g.dBoxes.p;
            rn.emptyRef = g.emptyRef, rn.n = (uint32_t)g.wide.size();
            hipLaunchKernelGGL(k_refit_nodes, dim3((rn.n + 127u) / 128u), dim3(128), 0, c->copyStream, rn);
        }
        RefitArgs ra {};
        ra.verts = (const VertexIn*)g.dVerts.p, ra.tri = c->st->triShade.p, ra.mats = c->st->materials.p, ra.tris = g.dTris.p, ra.fat = g.dFat.p, ra.n = (uint32_t)nT;
        hipLaunchKernelGGL(k_refit_tris, dim3(((uint32_t)nT + 255u) / 256u), dim3(256), 0, c->copyStream, ra);
        HIPCHK(c, hipGetLastError());
        tm.lap("kernels");
        g.onDevice = true;
        return PT_OK;
    }
    std::vector<TriIsect> tris = c->st->hostTris;
    tris.push_back(TriIsect { make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0) }); // what an unused child slot refers to
    if ((rc = uploadVec(c, g.dWide, g.wide)) || (rc = uploadVec(c, g.dBoxes, g.boxes)) || (rc = uploadVec(c, g.dLeafOfs, g.leafOfs))
        || (rc = uploadVec(c, g.dRefTri, g.refTri)) || (rc = uploadVec(c, g.dTris, tris)) || (rc = uploadVec(c, g.dFat, g.fat))
        || (rc = uploadVec(c, g.dVerts, c->st->rawVerts)) || (rc = uploadVec(c, c->st->triShade, c->st->hostTriShade))
        || (rc = uploadVec(c, g.dNodes, c->st->hostSubNodes)) || (rc = uploadVec(c, g.dKidBoxNode, g.kidBoxNode)))
        return rc;
    tm.lap("uploads");
    g.onDevice = true;
    return PT_OK;
}

// `async`: into c->st without touching the render stream or the dynamic state (pt_upload_static_async: c->st is the scene that is not current)
static int uploadStaticImpl(pt_ctx* c, const pt_vertex* verts, uint32_t nV, const pt_triangle* tris, uint32_t nT, const pt_material* mats,
    uint32_t nM, const pt_sub_bvh_node* nodes, uint32_t nN, bool async = false)
{
    {
    if (!c)
        return PT_ERR_INVALID;
    if (!verts || !tris || !mats || !nodes || nV == 0 || nT == 0 || nM == 0 || nN == 0)
        return fail(c, PT_ERR_INVALID, "pt_upload_static: empty or null scene array");
    if (nT > kRefIndexMask || nN > kRefIndexMask)
        return fail(c, PT_ERR_UNSUPPORTED, "pt_upload_static: more than 2^27 triangle references or nodes");
    HIPCHK(c, hipSetDevice(c->device));
    StageTimer tm("uploadStatic");
    // ---- validate: every index in range, children after their parent (rules out cycles) -----
    for (uint32_t t = 0; t < nT; t++) {
        if (tris[t].indices[0] >= nV || tris[t].indices[1] >= nV || tris[t].indices[2] >= nV)
            return fail(c, PT_ERR_INVALID, "triangle %u: vertex index out of range", t);
        if (tris[t].materialIndex >= nM)
            return fail(c, PT_ERR_INVALID, "triangle %u: material index out of range", t);
    }
    // An inner node whose children do not lie strictly after it is an unused pad (the reference's pair
    // allocator leaves one next to every root); pads may not be referenced.
    auto isPad = [&](uint32_t i) {
        const uint32_t l = nodes[i].leftChildOrFirstTriangle;
        return nodes[i].triangleCount == 0 && (l <= i || (uint64_t)l + 1 >= nN);
    };
    for (uint32_t i = 0; i < nN; i++) {
        const pt_sub_bvh_node& n = nodes[i];
        if (n.triangleCount != 0 && (uint64_t)n.leftChildOrFirstTriangle + n.triangleCount > nT)
            return fail(c, PT_ERR_INVALID, "sub-BVH leaf %u: triangle range out of bounds", i);
    }

    tm.lap("validate");
    // ---- triangles / vertices / materials ------------------------------------------------------
    std::vector<TriIsect> hTris(nT);
    std::vector<TriShade> hShade(nT);
    auto P = [&](uint32_t vi) { return mk(verts[vi].vertex[0], verts[vi].vertex[1], verts[vi].vertex[2]); };
    for (uint32_t t = 0; t < nT; t++) {
        const V3 v0 = P(tris[t].indices[0]);
        const V3 e1 = P(tris[t].indices[1]) - v0, e2 = P(tris[t].indices[2]) - v0; // shapes.cl:37-38
        hTris[t].a = make_float4(v0.x, v0.y, v0.z, e1.x);
        hTris[t].b = make_float4(e1.y, e1.z, e2.x, e2.y);
        hTris[t].c = make_float4(e2.z, 0.f, 0.f, 0.f);
        hShade[t] = { tris[t].indices[0], tris[t].indices[1], tris[t].indices[2], tris[t].materialIndex };
    }
    std::vector<VertexShade> hVerts(nV);
    for (uint32_t v = 0; v < nV; v++) {
        hVerts[v].n_u = make_float4(verts[v].normal[0], verts[v].normal[1], verts[v].normal[2], verts[v].texCoord[0]);
        hVerts[v].v_pad = make_float4(verts[v].texCoord[1], 0.f, 0.f, 0.f);
    }
    std::vector<Material> hMats(nM);
    static_assert(sizeof(Material) == sizeof(pt_material), "material record is copied verbatim");
    std::memcpy(hMats.data(), mats, (size_t)nM * sizeof(pt_material));

    tm.lap("records");
    // ---- pair nodes ----------------------------------------------------------------------------
    std::vector<uint32_t> dense(nN, 0xFFFFFFFFu);
    uint32_t numInner = 0;
    for (uint32_t i = 0; i < nN; i++)
        if (nodes[i].triangleCount == 0 && !isPad(i))
            dense[i] = numInner++;
    std::vector<PairNode> hNodes(numInner);
    auto triBox = [&](uint32_t t, V3& lo, V3& hi) {
        for (int k = 0; k < 3; k++) {
            const V3 p = P(tris[t].indices[k]);
            lo = mk(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
            hi = mk(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
        }
    };
    // Leaves larger than `maxLeaf` triangles become a small subtree over their triangle range.  Round 5: maxLeaf = 2, not the 30 a reference can
    // address -- the reference's builders stop at <= 3 triangles (src/bvh/bvh_build.cpp:15: 60 % of the benchmark meshes' leaves hold two, 39 % three),
    // and a leaf step of the traversal kernels runs to the LONGEST leaf among its lanes: with the three-triangle leaves cut into 1 + 2 at the cheaper
    // of the two places (the pieces go into free slots of the 4-wide nodes where there are any: 27 k -> 37 k nodes for 82 k triangles) every leaf step
    // is two trips at most.  Measured on the benchmark (one box, A / B / A): 10 868 -> 11 012 -> 10 832 Mrays/s (+1.5 %; leaves of ONE triangle: -0.7 %;
    // merging subtrees into leaves of up to 4 / 6 / 8 instead: -1.0 / -3.3 / -3.4 %, profiles/round5/r5_tree_shape.txt).  Parity mode keeps the caller's
    // leaves (its order of triangle tests is the reference's).  PTAMD_MAX_LEAF=n overrides (diagnostics).
#ifndef PT_MAX_LEAF
#define PT_MAX_LEAF 2
#endif
    uint32_t maxLeaf = parityMode(c) ? kMaxLeafTris : std::min<uint32_t>(PT_MAX_LEAF, kMaxLeafTris);
    if (const char* e = getenv("PTAMD_MAX_LEAF"))
        maxLeaf = std::max(1u, std::min((uint32_t)atoi(e), kMaxLeafTris));
    struct Range {
        uint32_t first, count;
    };
    auto rangeBox = [&](Range r, V3& lo, V3& hi) {
        lo = mk(FLT_MAX), hi = mk(-FLT_MAX);
        for (uint32_t t = 0; t < r.count; t++)
            triBox(r.first + t, lo, hi);
    };
    auto halfArea = [](V3 lo, V3 hi) {
        const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return dx * dy + dy * dz + dz * dx;
    };
    auto leafRefImpl = [&](auto& leafRef, Range r, V3& lo, V3& hi) -> uint32_t {
        lo = mk(FLT_MAX), hi = mk(-FLT_MAX);
        if (r.count <= maxLeaf) {
            for (uint32_t t = 0; t < r.count; t++)
                triBox(r.first + t, lo, hi);
            return makeRef(r.first, r.count);
        }
        // the range stays in the caller's order (a leaf is a run of it): cut where the two runs' surface-area cost is smallest
        uint32_t half = r.count / 2;
        if (r.count <= 8u) {
            float best = FLT_MAX;
            for (uint32_t cut = 1; cut < r.count; cut++) {
                V3 alo, ahi, blo, bhi;
                rangeBox({ r.first, cut }, alo, ahi);
                rangeBox({ r.first + cut, r.count - cut }, blo, bhi);
                const float cost = halfArea(alo, ahi) * (float)cut + halfArea(blo, bhi) * (float)(r.count - cut);
                if (cost < best)
                    best = cost, half = cut;
            }
        }
        V3 llo, lhi, rlo, rhi;
        const uint32_t l = leafRef(leafRef, { r.first, half }, llo, lhi);
        const uint32_t rr = leafRef(leafRef, { r.first + half, r.count - half }, rlo, rhi);
        PairNode pn {};
        pn.bx = make_float4(llo.x, lhi.x, rlo.x, rhi.x);
        pn.by = make_float4(llo.y, lhi.y, rlo.y, rhi.y);
        pn.bz = make_float4(llo.z, lhi.z, rlo.z, rhi.z);
        pn.left = l;
        pn.right = rr;
        lo = mk(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z));
        hi = mk(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z));
        hNodes.push_back(pn);
        return makeRef((uint32_t)hNodes.size() - 1, 0);
    };
    c->st->nodeRef.assign(nN, kRefNone);
    for (uint32_t i = 0; i < nN; i++) {
        if (nodes[i].triangleCount != 0) {
            if (nodes[i].triangleCount <= maxLeaf) {
                c->st->nodeRef[i] = makeRef(nodes[i].leftChildOrFirstTriangle, nodes[i].triangleCount);
            } else {
                V3 lo, hi;
                c->st->nodeRef[i] = leafRefImpl(leafRefImpl, { nodes[i].leftChildOrFirstTriangle, nodes[i].triangleCount }, lo, hi);
            }
        } else if (dense[i] != 0xFFFFFFFFu) {
            c->st->nodeRef[i] = makeRef(dense[i], 0);
        }
    }
    for (uint32_t i = 0; i < nN; i++) {
        if (dense[i] == 0xFFFFFFFFu)
            continue;
        const uint32_t l = nodes[i].leftChildOrFirstTriangle;
        const pt_sub_bvh_node& L = nodes[l];
        const pt_sub_bvh_node& R = nodes[l + 1];
        PairNode pn {};
        pn.bx = make_float4(L.min[0], L.max[0], R.min[0], R.max[0]);
        pn.by = make_float4(L.min[1], L.max[1], R.min[1], R.max[1]);
        pn.bz = make_float4(L.min[2], L.max[2], R.min[2], R.max[2]);
        pn.left = c->st->nodeRef[l];
        pn.right = c->st->nodeRef[l + 1];
        if (pn.left == kRefNone || pn.right == kRefNone)
            return fail(c, PT_ERR_INVALID, "sub-BVH node %u: child is an unused pad node", i);
        hNodes[dense[i]] = pn;
    }
    // depth of every subtree (children have larger indices: one reverse sweep), for the stack bound
    c->st->subtreeDepth.assign(nN, 0);
    for (uint32_t i = nN; i-- > 0;) {
        if (dense[i] == 0xFFFFFFFFu) {
            uint32_t extra = 0;
            for (uint32_t cnt = nodes[i].triangleCount; cnt > maxLeaf; cnt = cnt > 8u ? (cnt + 1) / 2 : cnt - 1) // (the cost-driven cut of a short run may peel one triangle off per level)
                extra++;
            c->st->subtreeDepth[i] = extra;
        } else {
            const uint32_t l = nodes[i].leftChildOrFirstTriangle;
            c->st->subtreeDepth[i] = 1 + std::max(c->st->subtreeDepth[l], c->st->subtreeDepth[l + 1]);
        }
    }
    if (hNodes.size() > kRefIndexMask)
        return fail(c, PT_ERR_UNSUPPORTED, "too many BVH nodes");

    tm.lap("pairNodes");
    int rc;
    c->st->hostTris = std::move(hTris);
    c->st->hostBottomNodes = std::move(hNodes);
    c->st->rawVerts.assign(verts, verts + nV);
    c->st->hostGeomStale = false;
    c->st->sg.latestInStage = false;
    c->st->denseOfNode = std::move(dense);
    c->st->numDensePairs = numInner;
    if (!async) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); // renders in flight read these buffers
        if ((rc = uploadVec(c, c->st->materials, hMats)))
            return rc;
    }
    c->st->hostTriShade = std::move(hShade);
    c->st->hostMaterials.assign(mats, mats + nM);
    c->st->hostSubNodes.assign(nodes, nodes + nN);
    c->st->hostVerts = std::move(hVerts);
    c->st->numVerts = nV;
    c->st->numRefNodes = nN;
    c->st->numTris = nT;
    c->st->have = true;
    c->st->hostNodeBoxesStale = false;
    if (!async) {
        c->haveStatic = true;
        c->haveDynamic = false; // top-level leaves reference sub-BVH roots: must be re-uploaded
        c->pending = -1;
        c->statPending = -1; // (a rebuilt scene that was waiting for its frame tick is dropped with the dynamic state)
    }
    {   // material types in use (emissive surfaces end a path in a few instructions: they do not count)
        uint32_t types = 0;
        for (uint32_t t = 0; t < nT; t++) {
            uint32_t ty;
            std::memcpy(&ty, (const char*)&mats[tris[t].materialIndex] + 32, 4); // the type word of the 48-byte record (Material::typeAndPad.x)
            types |= 1u << std::min(ty, 31u);
        }
        types &= ~(1u << MAT_EMISSIVE);
        c->st->materialBins = (types & (types - 1u)) != 0u && (c->cfg.flags & PT_FLAG_MATERIAL_BINS) != 0u; // opt-in: measured slower (pt_shade.h)
    }
    c->st->sg.extraRoots.clear();
    tm.lap("mirrors");
    if ((rc = buildStaticGeom(c, async && !getenv("PTAMD_HOST_RECORDS")))) // (PTAMD_HOST_RECORDS=1: the records of a rebuilt scene on the host too -- A / B)
        return rc;
    tm.lap("buildStaticGeom");
    // (pt_upload_static_async: the materials go last.  A blocking copy issued while a frame of the current scene holds every compute unit with its persistent
    // waves waited for most of that frame -- 0.43 ms of a 1080p frame, tools/rebuild_timing.py --1080p -- and the conversion above is host work the frame can hide)
    if (async && (rc = uploadVec(c, c->st->materials, hMats)))
        return rc;
    tm.lap("materials");
    if (!async)
        refreshSceneView(c);
    return PT_OK;
    }
}

// Who a packed node reports to in a bottom-up pass, and how many arrivals complete it (k_refit_tree).  Made once per topology.
static int ensureRefitTables(pt_ctx* c)
{
    StaticScene::StaticGeom& g = c->st->sg;
    if (g.refitTablesFor == g.topology)
        return PT_OK;
    const size_t n = g.wide.size();
    std::vector<uint32_t> parent(n, 0xFFFFFFFFu), need(n, 1u);
    bool ok = true;
    for (size_t q = 0; q < n; q++)
        for (int k = 0; k < 4; k++) {
            const uint32_t r = g.wide[q].child[k];
            if (r == g.emptyRef || refCount(r) != 0u)
                continue;
            const uint32_t ch = refIndex(r);
            if (ch >= n || ch == q || parent[ch] != 0xFFFFFFFFu) { // (two parents: roots that share a subtree -- the bottom-up pass would complete the child once and leave one parent waiting)
                ok = false;
                continue;
            }
            parent[ch] = ((uint32_t)q << 2) | (uint32_t)k;
            need[q]++;
        }
    g.refitTablesOk = ok;
    g.refitTablesFor = g.topology;
    if (!ok)
        return PT_OK;
    HIPCHK(c, hipStreamSynchronize(c->copyStream)); // (an earlier refit may still be walking the old tables)
    int rc;
    if ((rc = uploadVec(c, g.dParent, parent)) || (rc = uploadVec(c, g.dNeed, need)))
        return rc;
    if (!g.dArrived.p || g.dArrived.n < std::max<size_t>(n, 1))
        HIPCHK(c, g.dArrived.alloc(std::max<size_t>(n + n / 8, 1)));
    HIPCHK(c, hipMemset(g.dArrived.p, 0, g.dArrived.n * sizeof(uint32_t)));
    return PT_OK;
}

// What the host-side conversion of one dynamic state produces (no device call in it): the top level, the instance table, the lights,
// and the list of world-space copies the device is to make.  Everything below the top level is static (StaticScene::StaticGeom).
struct DynamicHost {
    std::vector<WideNode> topWide; // goes to wide[staticNodes ...]
    std::vector<Instance> instances;
    std::vector<Light> lights;
    std::vector<BakeJob> jobs;
    std::vector<uint32_t> instanceTopNode;
    uint32_t numLights = 0, rootRef = 0, rootRefFolded = 0;
    uint32_t foldedInstances = 0; // instances the per-ray kernels walk without parking (translation + uniform scale, pt_trace.h)
    std::vector<WideNode> instRoots; // their copies of their meshes' root nodes (one slot per instance), stored at instRootBase: the last run of the node array
    std::vector<uint32_t> instRootSrc; // per instance: the packed node its root copy is made from ON THE DEVICE (k_inst_roots), ~0: instRoots[k] holds it already
    std::vector<float4> instFold; // entry 1 + k: (1 / s, w) of instance k; entry 0 and the instances on the general route: the identity
    uint32_t instRootBase = 0;
    uint32_t topSlots = 0; // node slots reserved for the top level (the copies start behind them)
    uint32_t bakedNodes = 0, bakedTris = 0;
    bool packetOk = false, hasInstances = false, generalRoute = false;
    uint32_t stackNeed = 0, enteredInstances = 0, enteredGeneral = 0; // (... of which not a translation + uniform scale)
};

int convertDynamic(pt_ctx* c, const pt_emissive_triangle* lights, uint32_t nL, const pt_top_bvh_node* topNodes, uint32_t nTop, uint32_t topRoot, DynamicHost& out)
{
    StaticScene::StaticGeom& sg = c->st->sg;
    // a top-level leaf may name any node of the caller's sub-BVH array; the ones that are not mesh roots become roots of their own
    {
        bool grown = false;
        for (uint32_t i = 0; i < nTop; i++) {
            const pt_top_bvh_node& n = topNodes[i];
            if (!n.isLeaf)
                continue;
            if (n.a >= c->st->numRefNodes || c->st->nodeRef[n.a] == kRefNone)
                return fail(c, PT_ERR_INVALID, "top-level leaf %u: sub-BVH root %u is not a valid node", i, n.a);
            if (sg.rootOfNode[n.a] < 0 && std::find(sg.extraRoots.begin(), sg.extraRoots.end(), n.a) == sg.extraRoots.end()) {
                sg.extraRoots.push_back(n.a);
                grown = true;
            }
        }
        if (grown) {
            int rc = buildStaticGeom(c);
            if (rc)
                return rc;
        }
    }
    const uint32_t staticNodes = (uint32_t)sg.wide.size();
    const uint32_t staticTris = c->st->numTris + 1u; // the caller's triangles + the all-zero one
    // ---- instances (one per top-level leaf) and top-level pair nodes (one per top-level inner node)
    std::vector<Instance>& hInst = out.instances;
    hInst.clear();
    std::vector<uint32_t> topRef(nTop, kRefNone); // reference of top node i as a child
    std::vector<int32_t> instRoot; // instance -> roots[] slot
    out.instanceTopNode.clear();
    out.jobs.clear();
    uint32_t numTopInner = 0, maxBottomDepth = 0;
    for (uint32_t i = 0; i < nTop; i++) {
        const pt_top_bvh_node& n = topNodes[i];
        if (n.isLeaf) {
            maxBottomDepth = std::max(maxBottomDepth, c->st->subtreeDepth[n.a] + 1);
            const StaticScene::StaticGeom::Root& root = sg.roots[sg.rootOfNode[n.a]];
            Instance in {};
            const float* m = n.invTransform; // column-major
            in.r0 = make_float4(m[0], m[4], m[8], m[12]);
            in.r1 = make_float4(m[1], m[5], m[9], m[13]);
            in.r2 = make_float4(m[2], m[6], m[10], m[14]);
            in.rootRef = root.ref;
            in.topNode = i;
            {   // a translation + uniform scale?  (parity mode follows the reference's route to the letter)
                const float a = in.r0.x;
                in.simple = (!parityMode(c) && !(c->cfg.flags & PT_FLAG_PARKED_INSTANCES) && a > 0.f && std::isfinite(a) && in.r1.y == a && in.r2.z == a && in.r0.y == 0.f
                                && in.r0.z == 0.f && in.r1.x == 0.f && in.r1.z == 0.f && in.r2.x == 0.f && in.r2.y == 0.f && std::isfinite(in.r0.w) && std::isfinite(in.r1.w)
                                && std::isfinite(in.r2.w))
                    ? 1u : 0u;
            }
            if (hInst.size() >= kSpecialLeaveInstance)
                return fail(c, PT_ERR_UNSUPPORTED, "too many instances");
            topRef[i] = makeRef((uint32_t)hInst.size(), kRefSpecial);
            hInst.push_back(in);
            instRoot.push_back(sg.rootOfNode[n.a]);
            out.instanceTopNode.push_back(i);
        } else {
            if (n.a >= nTop || n.b >= nTop)
                return fail(c, PT_ERR_INVALID, "top-level node %u: child out of range", i);
            topRef[i] = makeRef(staticNodes + numTopInner, 0u);
            numTopInner++;
        }
    }
    // node slots of the top level: [the top level with instance references (<= numTopInner nodes) | the same top level for the per-ray kernels, which
    // walk translated + uniformly scaled instances without parking (<= numTopInner)]; the world-space copies start behind them, the instances' root
    // copies (one slot per instance) come last
    const uint32_t foldedBase = staticNodes + numTopInner;
    out.topSlots = 2u * numTopInner;
    // ---- instances copied to world space --------------------------------------------------------------------
    // An instance costs every ray that enters it a transform in and a restore out on top of the traversal proper.  With 288 GB of
    // HBM the instanced geometry of scenes like the benchmark's (12 x 82 k triangles: ~110 MB of nodes and triangles) simply fits
    // as world-space copies, so instances are copied while a byte budget lasts -- single-leaf meshes (a ground quad, an area light)
    // first, they cost almost nothing -- and the rest stay two-level.  (t,u,v) are the same in both spaces (the reference never
    // renormalises the transformed direction, scene.cl:118-121); the traversal kernels map a hit on a copy back to (original
    // triangle, instance).  The copies themselves are made on the device (pt_bake.h); this only lays them out.
    {
        uint64_t budgetBytes = 2ull << 30;
        if (const char* e = getenv("PTAMD_BAKE_BUDGET_GB")) // diagnostics (bench.py, two_level_general: the copied scene as the yardstick of the entered one)
            budgetBytes = (uint64_t)std::max(0.0, atof(e) * (double)(1ull << 30));
        uint64_t usedBytes = 0;
        uint32_t nextNode = staticNodes + out.topSlots, nextTri = staticTris;
        // Whole trees: ALL of them or none (round 6).  A scene that is partly copied pays for both: every ray runs the kernels that can enter instances,
        // and the copies' bytes push the shared trees out of the caches (432 instances of the 82 k-triangle meshes, 421 copied + 13 entered: 8 481 Mrays/s
        // against 9 089 with all of them entered and 9 017 with all of them copied: profiles/round6/).
        uint64_t allBytes = 0;
        for (uint32_t k = 0; k < hInst.size(); k++) {
            const StaticScene::StaticGeom::Root& root = sg.roots[instRoot[k]];
            if (refCount(root.ref) == 0u)
                allBytes += (uint64_t)root.numNodes * sizeof(WideNode) + (uint64_t)root.numRefs * sizeof(TriIsect);
        }
        const bool copiesAllowed = !(c->cfg.flags & PT_FLAG_NO_BAKED_INSTANCES), wholeTreesFit = allBytes <= budgetBytes;
        auto mayBake = [&](uint32_t instIndex) { // by the mesh alone (the byte budget and the index range are the layout's business, below)
            const StaticScene::StaticGeom::Root& root = sg.roots[instRoot[instIndex]];
            if (refCount(root.ref) != 0u)
                return true; // the mesh is one leaf
            return wholeTreesFit && root.bakeable && root.numNodes != 0u && !(c->cfg.flags & PT_FLAG_TWO_LEVEL_ONLY) && !parityMode(c); // parity mode follows the reference to the letter
        };
        // the world transforms of the instances that may be copied: a 4 x 4 inversion in double each -- ten thousand instances moved per tick are ten thousand of
        // them: on the host library's worker pool
        struct World {
            double m[12];
            bool ok;
        };
        std::vector<World> world(copiesAllowed ? hInst.size() : 0);
        raytracer::WorkerPool::get().parallelFor(world.size(), 256, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
                world[k].ok = false;
                if (!mayBake((uint32_t)k))
                    continue;
                double w[4][8]; // [r][4..7] = row r of the world transform
                if (!invertTransform(topNodes[hInst[k].topNode].invTransform, w))
                    continue; // singular: stays an instance
                for (int r = 0; r < 3; r++)
                    for (int col = 0; col < 4; col++)
                        world[k].m[r * 4 + col] = w[r][4 + col];
                world[k].ok = true;
            }
        });
        auto tryBake = [&](uint32_t instIndex, bool wholeTrees) {
            const StaticScene::StaticGeom::Root& root = sg.roots[instRoot[instIndex]];
            const bool single = refCount(root.ref) != 0u; // the mesh is one leaf
            if (single != !wholeTrees)
                return;
            if (!world[instIndex].ok)
                return; // not a mesh that is copied, or a singular transform
            const uint64_t bytes = (uint64_t)root.numNodes * sizeof(WideNode) + (uint64_t)root.numRefs * sizeof(TriIsect);
            if ((!single && usedBytes + bytes > budgetBytes) || (uint64_t)nextNode + root.numNodes >= kRefIndexMask - 4u
                || (uint64_t)nextTri + root.numRefs >= kRefIndexMask - 4u)
                return;
            usedBytes += bytes;
            BakeJob j {};
            std::memcpy(j.m, world[instIndex].m, sizeof j.m);
            j.srcNode = root.nodeBase, j.numNodes = root.numNodes, j.dstNode = nextNode;
            j.srcRef = root.refBase, j.numRefs = root.numRefs, j.dstTri = nextTri;
            j.instance = instIndex;
            out.jobs.push_back(j);
            topRef[hInst[instIndex].topNode] = single ? makeRef(nextTri, refCount(root.ref)) : makeRef(nextNode, 0u);
            nextNode += root.numNodes;
            nextTri += root.numRefs;
        };
        if (copiesAllowed) {
            for (uint32_t k = 0; k < hInst.size(); k++) // single leaves first
                tryBake(k, false);
            if (wholeTreesFit)
                for (uint32_t k = 0; k < hInst.size(); k++)
                    tryBake(k, true);
        }
        out.bakedNodes = nextNode - (staticNodes + out.topSlots);
        out.bakedTris = nextTri - staticTris;
    }
    uint32_t topDepth = 0;
    { // depth / cycle check from the root
        std::vector<std::pair<uint32_t, uint32_t>> st { { topRoot, 1u } };
        size_t visited = 0;
        while (!st.empty()) {
            auto [ni, depth] = st.back();
            st.pop_back();
            if (++visited > nTop)
                return fail(c, PT_ERR_INVALID, "top-level BVH is not a tree");
            topDepth = std::max(topDepth, depth);
            if (!topNodes[ni].isLeaf) {
                st.push_back({ topNodes[ni].a, depth + 1 });
                st.push_back({ topNodes[ni].b, depth + 1 });
            }
        }
    }
    // one pending entry per level of either tree + the leave-instance sentinel
    if (topDepth + 1 + maxBottomDepth > (uint32_t)kTraversalStackMax)
        return fail(c, PT_ERR_UNSUPPORTED, "BVH depth %u (top) + %u (bottom) exceeds the traversal stack (%d)", topDepth, maxBottomDepth, kTraversalStackMax);
    if ((uint64_t)staticNodes + out.topSlots + out.bakedNodes + hInst.size() > kRefIndexMask)
        return fail(c, PT_ERR_UNSUPPORTED, "too many BVH nodes");
    // ---- the top level: pair nodes -> 4-wide, packed breadth-first into the slots behind the static nodes -----------------
    std::vector<PairNode> topPairs(numTopInner);
    auto local = [&](uint32_t ref) { return refIndex(ref) - staticNodes; }; // top-level inner reference -> index into topPairs
    auto isTopInner = [&](uint32_t ref) { return ref != kRefNone && refCount(ref) == 0u && refIndex(ref) >= staticNodes && refIndex(ref) < staticNodes + numTopInner; };
    for (uint32_t i = 0; i < nTop; i++) {
        const pt_top_bvh_node& n = topNodes[i];
        if (n.isLeaf)
            continue;
        const pt_top_bvh_node& L = topNodes[n.a];
        const pt_top_bvh_node& R = topNodes[n.b];
        PairNode pn {};
        pn.bx = make_float4(L.min[0], L.max[0], R.min[0], R.max[0]);
        pn.by = make_float4(L.min[1], L.max[1], R.min[1], R.max[1]);
        pn.bz = make_float4(L.min[2], L.max[2], R.min[2], R.max[2]);
        // inside the collapse the top-level children are indices into topPairs; every other reference is opaque to it (instance
        // references and leaves by their count, the roots of world-space copies by an index beyond the array: they start behind the
        // top level's slots)
        pn.left = isTopInner(topRef[n.a]) ? makeRef(local(topRef[n.a]), 0u) : topRef[n.a];
        pn.right = isTopInner(topRef[n.b]) ? makeRef(local(topRef[n.b]), 0u) : topRef[n.b];
        topPairs[local(topRef[i])] = pn;
    }
    const std::vector<WideKids> kids = collapseKids(topPairs);
    // breadth-first packing of the top-level nodes the collapse kept
    uint32_t rootRef = topRef[topRoot];
    constexpr uint32_t kUnset = 0xFFFFFFFFu;
    std::vector<uint32_t> newIndex(numTopInner, kUnset), order;
    auto isKept = [&](uint32_t r) { return r != kRefNone && refCount(r) == 0u && refIndex(r) < numTopInner; };
    if (isTopInner(rootRef)) {
        newIndex[local(rootRef)] = 0;
        order.push_back(local(rootRef));
        for (size_t q = 0; q < order.size(); q++)
            for (int k = 0; k < 4; k++) {
                const uint32_t r = kids[order[q]].ref[k];
                if (!kids[order[q]].empty[k] && isKept(r) && newIndex[refIndex(r)] == kUnset) {
                    newIndex[refIndex(r)] = (uint32_t)order.size();
                    order.push_back(refIndex(r));
                }
            }
        rootRef = makeRef(staticNodes, 0u);
    }
    out.topWide.resize(order.size());
    raytracer::WorkerPool::get().parallelFor(order.size(), 512, [&](size_t q0, size_t q1) {
        for (size_t q = q0; q < q1; q++) {
            const WideKids& wk = kids[order[q]];
            uint32_t refs[4];
            for (int k = 0; k < 4; k++)
                refs[k] = wk.empty[k] ? sg.emptyRef : (isKept(wk.ref[k]) ? makeRef(staticNodes + newIndex[refIndex(wk.ref[k])], 0u) : wk.ref[k]);
            quantiseWideNode(wk.lo, wk.hi, refs, wk.empty, sg.emptyRef, &out.topWide[q]);
        }
    });
    out.hasInstances = refCount(rootRef) == kRefSpecial;
    for (size_t q = 0; q < order.size() && !out.hasInstances; q++)
        for (uint32_t r : out.topWide[q].child)
            if (r != sg.emptyRef && refCount(r) == kRefSpecial)
                out.hasInstances = true;
    // ---- worst-case traversal stack: the top level on top of the deepest thing below it (an entered instance adds its sentinel)
    std::vector<uint32_t> topNeed(order.size(), 0u);
    // the copies' roots are looked up by node index: a map for scenes with many of them
    std::vector<std::pair<uint32_t, uint32_t>> copyRoots;
    for (const BakeJob& j : out.jobs)
        if (j.numNodes)
            copyRoots.push_back({ j.dstNode, sg.stackNeed[j.srcNode] });
    std::sort(copyRoots.begin(), copyRoots.end());
    auto needOf = [&](uint32_t ref) -> uint32_t {
        if (refCount(ref) == 0u && refIndex(ref) >= staticNodes && refIndex(ref) < staticNodes + order.size())
            return topNeed[refIndex(ref) - staticNodes];
        if (refCount(ref) == 0u) {
            auto it = std::lower_bound(copyRoots.begin(), copyRoots.end(), std::make_pair(refIndex(ref), 0u));
            return it != copyRoots.end() && it->first == refIndex(ref) ? it->second : 0u;
        }
        if (refCount(ref) == kRefSpecial) { // an entered instance: its sentinel + its mesh tree
            const uint32_t rr = hInst[refIndex(ref)].rootRef;
            return 1u + (refCount(rr) == 0u ? sg.stackNeed[refIndex(rr)] : 0u);
        }
        return 0u;
    };
    for (size_t q = order.size(); q-- > 0;) {
        uint32_t n = 0, deepest = 0;
        for (uint32_t r : out.topWide[q].child)
            if (r != sg.emptyRef)
                n++, deepest = std::max(deepest, needOf(r));
        topNeed[q] = (n > 0 ? n - 1 : 0u) + deepest;
    }
    const uint32_t stackNeed = needOf(rootRef);
    if (stackNeed > (uint32_t)kTraversalStackMax)
        return fail(c, PT_ERR_UNSUPPORTED, "BVH needs %u traversal stack entries, %d are available", stackNeed, kTraversalStackMax);
    // k_trace_packet keeps its stack in the 64 lanes of a register (instance references are entered there too, pt_packet.h)
    out.packetOk = stackNeed <= kPacketStack;
    out.stackNeed = stackNeed;
    // ---- the top level once more, for the per-ray kernels: instances whose transform is a translation + uniform scale (the reference's own scenes,
    // BASELINE configs 4 / 5) are walked WITHOUT parking (pt_trace.h).  In this copy of the top level such an instance is an ordinary inner reference
    // -- to the instance's own copy of its mesh's ROOT node (object space, 64 bytes; the copies are the LAST run of the node array, copy k = instance
    // k) -- and (1 / s, w = -t / s) of its inverse transform sits in a table the kernel keeps in LDS.  Same pairs, same boxes, hence the same
    // collapse and a worst-case stack no larger than the one computed above (no sentinel).
    out.rootRefFolded = rootRef;
    out.foldedInstances = 0;
    out.instRoots.clear();
    out.instRootSrc.clear();
    out.instFold.clear();
    out.instRootBase = staticNodes + out.topSlots + out.bakedNodes;
    {
        static const bool envNoFold = getenv("PTAMD_NO_FOLDED_INSTANCES") != nullptr; // diagnostics: every entered instance takes the parked route (rounds 2-4)
        const bool parked = envNoFold || (c->cfg.flags & PT_FLAG_PARKED_INSTANCES) != 0u || parityMode(c); // (parity mode follows the reference to the letter)
        auto simple = [](const Instance& in) { return in.simple != 0u; };
        // Which route for the instances that are entered?  Every one a translation + uniform scale and few enough for the LDS table: folded (no entry step at
        // all).  Otherwise -- a rotation, a non-uniform scale, a shear, or instance number 96 -- the general route (round 6): every instance is entered as a
        // leaf-kind step, nothing is parked (pt_trace.h, LEVELS 2).  PTAMD_GENERAL_ROUTE=1 / 0 (diagnostics): the general route for every scene with entered
        // instances / never (rounds 2-5: such scenes park).
        uint32_t entered = 0, enteredGeneral = 0;
        for (size_t k = 0; k < hInst.size(); k++)
            if (refCount(topRef[hInst[k].topNode]) == kRefSpecial)
                entered++, enteredGeneral += simple(hInst[k]) ? 0u : 1u;
        out.enteredInstances = entered, out.enteredGeneral = enteredGeneral;
        // A scene with FEW such instances among many translated + uniformly scaled ones (at most a quarter) that fits the table keeps the folded route
        // for those -- no entry step at all -- and parks the few.
        static const char* envGeneral = getenv("PTAMD_GENERAL_ROUTE");
        const bool tableHolds = hInst.size() + 1 <= kInstFoldTable;
        const bool mostlySimple = enteredGeneral * 4u <= entered;
        out.generalRoute = !parked && entered > 0u && (envGeneral ? atoi(envGeneral) != 0 : (!mostlySimple || !tableHolds));
        const bool noFold = parked || out.generalRoute || !tableHolds;
        if (out.generalRoute) { // the entry records of the general route (pt_trace.h): 32 bytes per instance
            out.instFold.assign(hInst.size() * 2, make_float4(0.f, 0.f, 0.f, 0.f));
            for (size_t k = 0; k < hInst.size(); k++) {
                const Instance& in = hInst[k];
                const bool sim = simple(in);
                out.instFold[2 * k] = sim ? make_float4(in.r0.x, in.r0.w, in.r1.w, in.r2.w) : make_float4(1.f, 0.f, 0.f, 0.f);
                float4 tail = make_float4(0.f, 0.f, sim ? 1.0f / in.r0.x : 1.f, 0.f);
                std::memcpy(&tail.x, &in.rootRef, 4);
                const uint32_t flag = sim ? 1u : 0u;
                std::memcpy(&tail.y, &flag, 4);
                out.instFold[2 * k + 1] = tail;
            }
        }
        std::vector<uint8_t> folded(hInst.size(), 0);
        for (size_t k = 0; k < hInst.size() && !noFold; k++)
            if (refCount(topRef[hInst[k].topNode]) == kRefSpecial && simple(hInst[k]))
                folded[k] = 1, out.foldedInstances++;
        if (out.foldedInstances) {
            const uint32_t instRootBase = out.instRootBase;
            auto foldRef = [&](uint32_t r) { return refCount(r) == kRefSpecial && refIndex(r) < hInst.size() && folded[refIndex(r)] ? makeRef(instRootBase + refIndex(r), 0u) : r; };
            std::vector<PairNode> pairsB = topPairs;
            for (PairNode& pn : pairsB)
                pn.left = foldRef(pn.left), pn.right = foldRef(pn.right);
            const std::vector<WideKids> kidsB = collapseKids(pairsB);
            std::vector<uint32_t> newB(numTopInner, kUnset), orderB;
            uint32_t rootB = foldRef(topRef[topRoot]);
            if (isTopInner(topRef[topRoot])) {
                newB[local(topRef[topRoot])] = 0;
                orderB.push_back(local(topRef[topRoot]));
                for (size_t q = 0; q < orderB.size(); q++)
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r = kidsB[orderB[q]].ref[k];
                        if (!kidsB[orderB[q]].empty[k] && isKept(r) && newB[refIndex(r)] == kUnset) {
                            newB[refIndex(r)] = (uint32_t)orderB.size();
                            orderB.push_back(refIndex(r));
                        }
                    }
                rootB = makeRef(foldedBase, 0u);
            }
            out.topWide.resize((size_t)numTopInner + orderB.size()); // (the gap behind the first top level stays zero: never referenced)
            raytracer::WorkerPool::get().parallelFor(orderB.size(), 512, [&](size_t q0, size_t q1) {
                for (size_t q = q0; q < q1; q++) {
                    const WideKids& wk = kidsB[orderB[q]];
                    uint32_t refs[4];
                    for (int k = 0; k < 4; k++)
                        refs[k] = wk.empty[k] ? sg.emptyRef : (isKept(wk.ref[k]) ? makeRef(foldedBase + newB[refIndex(wk.ref[k])], 0u) : wk.ref[k]);
                    quantiseWideNode(wk.lo, wk.hi, refs, wk.empty, sg.emptyRef, &out.topWide[(size_t)numTopInner + q]);
                }
            });
            out.rootRefFolded = rootB;
            // the instances' root copies and the table of their transforms (entry 0: the identity; instances on the general route: the identity too --
            // their lanes hold the instance-space ray in registers)
            out.instRoots.assign(hInst.size(), WideNode {});
            out.instRootSrc.assign(hInst.size(), 0xFFFFFFFFu);
            out.instFold.assign(hInst.size() + 1, make_float4(1.f, 0.f, 0.f, 0.f));
            for (size_t k = 0; k < hInst.size(); k++) {
                if (!folded[k])
                    continue;
                Instance& in = hInst[k];
                out.instFold[k + 1] = make_float4(in.r0.x, in.r0.w, in.r1.w, in.r2.w);
                in.folded = 1u;
                if (refCount(in.rootRef) == 0u) {
                    out.instRootSrc[k] = refIndex(in.rootRef); // the mesh's packed root node as the device holds it (the host's mirror goes stale with a refit): object space, children in the shared tree
                } else { // the mesh is a single leaf: a one-child node around it -- the top-level leaf's box taken into object space, a few ulps outwards
                    const pt_top_bvh_node& leaf = topNodes[in.topNode];
                    float lo[4][3], hi[4][3];
                    const uint32_t refs[4] = { in.rootRef, sg.emptyRef, sg.emptyRef, sg.emptyRef };
                    const bool empty[4] = { false, true, true, true };
                    const float w[3] = { in.r0.w, in.r1.w, in.r2.w };
                    for (int a = 0; a < 3; a++) {
                        const double l = (double)leaf.min[a] * in.r0.x + w[a], h = (double)leaf.max[a] * in.r0.x + w[a];
                        lo[0][a] = nextafterf(nextafterf((float)l, -INFINITY), -INFINITY), hi[0][a] = nextafterf(nextafterf((float)h, INFINITY), INFINITY);
                        for (int q = 1; q < 4; q++)
                            lo[q][a] = 1.f, hi[q][a] = -1.f;
                    }
                    quantiseWideNode(lo, hi, refs, empty, sg.emptyRef, &out.instRoots[k]);
                }
            }
        }
    }
    std::vector<Light>& hLights = out.lights;
    hLights.resize(nL);
    for (uint32_t i = 0; i < nL; i++) {
        const pt_emissive_triangle& e = lights[i];
        const V3 v0 = mk(e.vertices[0][0], e.vertices[0][1], e.vertices[0][2]);
        const V3 v1 = mk(e.vertices[1][0], e.vertices[1][1], e.vertices[1][2]);
        const V3 v2 = mk(e.vertices[2][0], e.vertices[2][1], e.vertices[2][2]);
        // Heron's formula (shading_helper.cl:204-214)
        const V3 A = v1 - v0, B = v2 - v1, C = v0 - v2;
        const float la = sqrtf(dot(A, A)), lb = sqrtf(dot(B, B)), lc = sqrtf(dot(C, C));
        const float s = (la + lb + lc) / 2.0f;
        const float area = sqrtf(s * (s - la) * (s - lb) * (s - lc));
        const V3 nrm = normalize(cross(v1 - v0, v2 - v0));
        hLights[i].v0 = make_float4(v0.x, v0.y, v0.z, area);
        hLights[i].v1 = make_float4(v1.x, v1.y, v1.z, 0.f);
        hLights[i].v2 = make_float4(v2.x, v2.y, v2.z, 0.f);
        hLights[i].normal = make_float4(nrm.x, nrm.y, nrm.z, 0.f);
        hLights[i].colour = make_float4(e.material.u.emissive.emissiveColour[0], e.material.u.emissive.emissiveColour[1], e.material.u.emissive.emissiveColour[2], 0.f);
    }
    out.numLights = nL;
    out.rootRef = rootRef;
    return PT_OK;
}

template <typename T>
int growTo(pt_ctx* c, DevBuf<T>& buf, size_t count)
{
    if (buf.n >= std::max<size_t>(count, 1))
        return PT_OK;
    HIPCHK(c, buf.alloc(std::max<size_t>(count + count / 8, 1))); // some headroom: the number of world-space copies varies from state to state
    return PT_OK;
}

} // namespace
