// Bottom-level BVH construction: binned SAH (3 axes or longest axis) and SBVH with spatial
// splits + reference unsplitting.  See bvh_build.h for the contract and the reference citations.
#include "bvh_build.h"
#include "parallel.h"
#include <algorithm>
#include <array>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace raytracer {
namespace {

    constexpr uint32_t kLeafSize = 3; // MIN_PRIMS_PER_LEAF (bvh_build.cpp:15)
    constexpr float kTraversalCost = 1.5f; // SAH_TRAVERSAL_COST (:17)
    constexpr float kIntersectCost = 1.0f; // SAH_INTERSECTION_COST (:18)
    constexpr float kSpatialAlpha = 1e-5f; // SPATIAL_SPLIT_ALPHA (:16)
    constexpr int kObjectBins = 32; // bvh_object_split.cpp:10
    constexpr int kSpatialBins = 8; // bvh_spatial_split.cpp:10
    constexpr int kMedianFallbackDepth = 40;

    struct PrimRef {
        uint32_t prim;
        AABB box;
    };

    struct Candidate {
        bool valid = false;
        int axis = 0;
        int plane = 0; // split between bin plane-1 and bin plane
        float cost = 0.0f; // SA_L*N_L + SA_R*N_R
        AABB leftBox, rightBox;
        size_t leftCount = 0, rightCount = 0;
        // a node whose bins the workers filled side by side: how many references of each worker's range go left (the partition scatters the ranges side by side too)
        std::vector<size_t> leftOfPart;
        size_t partChunk = 0;
    };

    inline vec3 position(const pt_vertex& v) { return { v.vertex[0], v.vertex[1], v.vertex[2] }; }

    inline void storeBox(SubBVHNode& n, const AABB& b)
    {
        n.min[0] = b.min.x, n.min[1] = b.min.y, n.min[2] = b.min.z, n.min[3] = 0.0f;
        n.max[0] = b.max.x, n.max[1] = b.max.y, n.max[2] = b.max.z, n.max[3] = 0.0f;
    }
    inline AABB loadBox(const SubBVHNode& n) { return { { n.min[0], n.min[1], n.min[2] }, { n.max[0], n.max[1], n.max[2] } }; }

    inline int longestAxis(vec3 e)
    {
        int a = 0;
        if (e.y > e[a]) a = 1;
        if (e.z > e[a]) a = 2;
        return a;
    }

    constexpr size_t kParallelBinning = 4096; // references of a node from which the object-split bins are filled by all workers
    inline bool sequentialBuild()
    {
        const char* env = std::getenv("PTAMD_BUILD_THREADS");
        return env && std::atoi(env) == 1;
    }

    inline float centerOn(const AABB& b, int axis) { return (b.min[axis] + b.max[axis]) / 2.0f; } // AABB::center()[axis], one component computed

    inline int binOf(float x, float lo, float invWidth, int numBins)
    {
        int b = (int)((x - lo) * invWidth);
        return std::clamp(b, 0, numBins - 1);
    }

    // ---- object split: bin reference-box centroids, sweep 31 planes -----------------------
    // (the sweeps visit the occupied bins only: a plane behind an empty bin repeats the candidate before it, which the strict comparison never
    // prefers -- the same choice as a sweep over all 31 planes, at the cost of the bins a small node really fills)
    Candidate findObjectSplit(const AABB& nodeBox, const PrimRef* refs, size_t numRefs, const int* axes, int numAxes)
    {
        Candidate best;
        vec3 ext = nodeBox.extent();
        for (int ai = 0; ai < numAxes; ai++) {
            const int axis = axes[ai];
            if (!(ext[axis] > std::numeric_limits<float>::min()))
                continue;
            const float invWidth = (float)kObjectBins / ext[axis];
            std::array<AABB, kObjectBins> box;
            std::array<size_t, kObjectBins> count {};
            std::vector<std::array<size_t, kObjectBins>> pcount; // (per worker: the top of a large tree)
            size_t partChunk = 0;
            auto binRange = [&](size_t begin, size_t end, std::array<AABB, kObjectBins>& bx, std::array<size_t, kObjectBins>& cn) {
                for (size_t i = begin; i < end; i++) {
                    const PrimRef& r = refs[i];
                    int b = binOf(centerOn(r.box, axis), nodeBox.min[axis], invWidth, kObjectBins);
                    bx[b].fit(r.box);
                    cn[b]++;
                }
            };
            if (numRefs >= kParallelBinning && WorkerPool::get().threads() > 1 && !WorkerPool::insideTask() && !sequentialBuild()) {
                // the top of a large tree: every worker bins a range into bins of its own (min / max and counts: the merged bins do not depend on the split)
                WorkerPool& pool = WorkerPool::get();
                const size_t parts = pool.threads() * 2, chunk = (numRefs + parts - 1) / parts; // (two ranges per thread: whoever is faster takes more)
                std::vector<std::array<AABB, kObjectBins>> pbox(parts);
                pcount.resize(parts);
                partChunk = chunk;
                for (auto& c : pcount)
                    c.fill(0);
                pool.parallelFor(parts, 1, [&](size_t p0, size_t p1) {
                    for (size_t p = p0; p < p1; p++)
                        binRange(std::min(p * chunk, numRefs), std::min((p + 1) * chunk, numRefs), pbox[p], pcount[p]);
                });
                for (size_t p = 0; p < parts; p++)
                    for (int b = 0; b < kObjectBins; b++) {
                        box[b].fit(pbox[p][b]);
                        count[b] += pcount[p][b];
                    }
            } else {
                binRange(0, numRefs, box, count);
            }
            uint32_t occupied = 0;
            for (int b = 0; b < kObjectBins; b++)
                occupied |= count[b] ? 1u << b : 0u;
            // what lies right of each occupied bin (boxes / counts accumulated from the top down), then a forward sweep
            std::array<AABB, kObjectBins> rightBox; // [b]: the bins above b
            std::array<size_t, kObjectBins> rightCount;
            AABB acc;
            size_t n = 0;
            for (uint32_t m = occupied; m;) {
                const int b = 31 - __builtin_clz(m);
                m &= ~(1u << b);
                rightBox[b] = acc;
                rightCount[b] = n;
                acc.fit(box[b]);
                n += count[b];
            }
            AABB left;
            size_t nl = 0;
            for (uint32_t m = occupied; m; m &= m - 1) {
                const int b = __builtin_ctz(m);
                left.fit(box[b]);
                nl += count[b];
                if (rightCount[b] == 0)
                    break; // the last occupied bin: nothing on the right
                float cost = (float)nl * left.surfaceArea() + (float)rightCount[b] * rightBox[b].surfaceArea();
                if (!best.valid || cost < best.cost) {
                    best.valid = true;
                    best.axis = axis;
                    best.plane = b + 1;
                    best.cost = cost;
                    best.leftBox = left;
                    best.rightBox = rightBox[b];
                    best.leftCount = nl;
                    best.rightCount = rightCount[b];
                    best.partChunk = partChunk;
                    best.leftOfPart.assign(pcount.size(), 0);
                    for (size_t p = 0; p < pcount.size(); p++)
                        for (int lb = 0; lb <= b; lb++)
                            best.leftOfPart[p] += pcount[p][lb];
                }
            }
        }
        return best;
    }

    void applyObjectSplit(const AABB& nodeBox, const Candidate& c, std::vector<PrimRef>& refs, std::vector<PrimRef>& left, std::vector<PrimRef>& right)
    {
        const float invWidth = (float)kObjectBins / nodeBox.extent()[c.axis];
        left.reserve(c.leftCount);
        right.reserve(c.rightCount);
        for (PrimRef& r : refs) {
            int b = binOf(r.box.center()[c.axis], nodeBox.min[c.axis], invWidth, kObjectBins);
            (b < c.plane ? left : right).push_back(r);
        }
    }

    // ---- triangle clipping (Sutherland-Hodgman against one axis-aligned half space) ---------
    struct Polygon {
        vec3 p[12];
        int n = 0;
    };

    void clipHalfSpace(Polygon& poly, int axis, float plane, bool keepGreater)
    {
        Polygon out;
        for (int i = 0; i < poly.n; i++) {
            vec3 a = poly.p[i], b = poly.p[(i + 1) % poly.n];
            bool ina = keepGreater ? a[axis] >= plane : a[axis] <= plane;
            bool inb = keepGreater ? b[axis] >= plane : b[axis] <= plane;
            if (ina)
                out.p[out.n++] = a;
            if (ina != inb) {
                float t = (plane - a[axis]) / (b[axis] - a[axis]);
                vec3 x = a + (b - a) * t;
                x[axis] = plane; // exact on the plane
                out.p[out.n++] = x;
            }
        }
        poly = out;
    }

    // Bounds of triangle (a,b,c) clipped to `clip`; false if nothing is left.
    bool clippedBounds(vec3 a, vec3 b, vec3 c, const AABB& clip, AABB& out)
    {
        Polygon poly;
        poly.p[0] = a, poly.p[1] = b, poly.p[2] = c;
        poly.n = 3;
        for (int axis = 0; axis < 3 && poly.n >= 3; axis++) {
            clipHalfSpace(poly, axis, clip.min[axis], true);
            if (poly.n < 3)
                break;
            clipHalfSpace(poly, axis, clip.max[axis], false);
        }
        if (poly.n < 3)
            return false;
        AABB r;
        for (int i = 0; i < poly.n; i++)
            r.fit(poly.p[i]);
        // guard against round-off pushing the result outside the clip box
        r.min = vmax(r.min, clip.min);
        r.max = vmin(r.max, clip.max);
        out = r;
        return true;
    }

    struct Builder {
        const pt_vertex* verts;
        const pt_triangle* tris;
        BvhBuilder kind;
        std::vector<SubBVHNode> nodes;
        std::vector<PrimRef> leafRefs;
        float rootArea = 0.0f;
        uint32_t maxDepthSeen = 0;

        uint32_t allocPair()
        {
            uint32_t first = (uint32_t)nodes.size();
            SubBVHNode blank;
            std::memset(&blank, 0, sizeof(blank));
            storeBox(blank, AABB());
            nodes.push_back(blank);
            nodes.push_back(blank);
            if (deferred)
                pairEpoch.push_back((uint32_t)deferred->size());
            return first;
        }

        void triangle(uint32_t prim, vec3& a, vec3& b, vec3& c) const
        {
            a = position(verts[tris[prim].indices[0]]);
            b = position(verts[tris[prim].indices[1]]);
            c = position(verts[tris[prim].indices[2]]);
        }

        // ---- spatial split search: chopped-triangle binning with enter/exit counters -------
        Candidate findSpatialSplit(const AABB& nodeBox, const std::vector<PrimRef>& refs) const
        {
            Candidate best;
            vec3 ext = nodeBox.extent();
            for (int axis = 0; axis < 3; axis++) {
                if (!(ext[axis] > std::numeric_limits<float>::min()))
                    continue;
                const float width = ext[axis] / (float)kSpatialBins;
                const float invWidth = (float)kSpatialBins / ext[axis];
                float planes[kSpatialBins + 1];
                for (int i = 0; i <= kSpatialBins; i++)
                    planes[i] = nodeBox.min[axis] + (float)i * width;
                planes[kSpatialBins] = nodeBox.max[axis];

                std::array<AABB, kSpatialBins> box;
                std::array<size_t, kSpatialBins> enter {}, leave {};
                for (const PrimRef& r : refs) {
                    int lo = binOf(r.box.min[axis], nodeBox.min[axis], invWidth, kSpatialBins);
                    int hi = binOf(r.box.max[axis], nodeBox.min[axis], invWidth, kSpatialBins);
                    if (lo == hi) {
                        box[lo].fit(r.box);
                        enter[lo]++;
                        leave[lo]++;
                        continue;
                    }
                    vec3 a, b, c;
                    triangle(r.prim, a, b, c);
                    int first = kSpatialBins, last = -1;
                    for (int bin = lo; bin <= hi; bin++) {
                        AABB clip = r.box;
                        clip.min[axis] = std::max(clip.min[axis], planes[bin]);
                        clip.max[axis] = std::min(clip.max[axis], planes[bin + 1]);
                        AABB piece;
                        if (clip.min[axis] <= clip.max[axis] && clippedBounds(a, b, c, clip, piece)) {
                            box[bin].fit(piece);
                            first = std::min(first, bin);
                            last = std::max(last, bin);
                        }
                    }
                    if (first <= last) {
                        enter[first]++;
                        leave[last]++;
                    }
                }
                std::array<AABB, kSpatialBins> rightBox;
                std::array<size_t, kSpatialBins> rightCount {};
                AABB acc;
                size_t n = 0;
                for (int b = kSpatialBins - 1; b >= 1; b--) {
                    acc.fit(box[b]);
                    n += leave[b];
                    rightBox[b] = acc;
                    rightCount[b] = n;
                }
                AABB left;
                size_t nl = 0;
                for (int plane = 1; plane < kSpatialBins; plane++) {
                    left.fit(box[plane - 1]);
                    nl += enter[plane - 1];
                    if (nl == 0 || rightCount[plane] == 0)
                        continue;
                    float cost = (float)nl * left.surfaceArea() + (float)rightCount[plane] * rightBox[plane].surfaceArea();
                    if (!best.valid || cost < best.cost) {
                        best.valid = true;
                        best.axis = axis;
                        best.plane = plane;
                        best.cost = cost;
                        best.leftBox = left;
                        best.rightBox = rightBox[plane];
                        best.leftCount = nl;
                        best.rightCount = rightCount[plane];
                    }
                }
            }
            return best;
        }

        // Distribute references around the chosen plane; straddlers are either duplicated with
        // clipped boxes or, when cheaper, kept whole on one side (SBVH paper 4.4, "unsplitting").
        bool applySpatialSplit(const AABB& nodeBox, const Candidate& c, const std::vector<PrimRef>& refs,
            std::vector<PrimRef>& left, std::vector<PrimRef>& right, AABB& leftBox, AABB& rightBox) const
        {
            const int axis = c.axis;
            float pos = nodeBox.min[axis] + (float)c.plane * (nodeBox.extent()[axis] / (float)kSpatialBins);
            const float saL = c.leftBox.surfaceArea(), saR = c.rightBox.surfaceArea();
            const float splitCost = saL * (float)c.leftCount + saR * (float)c.rightCount;
            leftBox = AABB();
            rightBox = AABB();
            for (const PrimRef& r : refs) {
                if (r.box.max[axis] <= pos) {
                    left.push_back(r);
                    leftBox.fit(r.box);
                } else if (r.box.min[axis] >= pos) {
                    right.push_back(r);
                    rightBox.fit(r.box);
                } else {
                    float onlyLeft = c.leftBox.merged(r.box).surfaceArea() * (float)c.leftCount + saR * (float)(c.rightCount - 1);
                    float onlyRight = saL * (float)(c.leftCount - 1) + c.rightBox.merged(r.box).surfaceArea() * (float)c.rightCount;
                    if (std::min(onlyLeft, onlyRight) < splitCost) {
                        if (onlyLeft < onlyRight) {
                            left.push_back(r);
                            leftBox.fit(r.box);
                        } else {
                            right.push_back(r);
                            rightBox.fit(r.box);
                        }
                        continue;
                    }
                    vec3 a, b, cc;
                    triangle(r.prim, a, b, cc);
                    AABB clipL = r.box, clipR = r.box, piece;
                    clipL.max[axis] = pos;
                    clipR.min[axis] = pos;
                    bool any = false;
                    if (clippedBounds(a, b, cc, clipL, piece)) {
                        left.push_back({ r.prim, piece });
                        leftBox.fit(piece);
                        any = true;
                    }
                    if (clippedBounds(a, b, cc, clipR, piece)) {
                        right.push_back({ r.prim, piece });
                        rightBox.fit(piece);
                        any = true;
                    }
                    if (!any) { // numerically degenerate sliver: keep it whole on the left
                        left.push_back(r);
                        leftBox.fit(r.box);
                    }
                }
            }
            return !left.empty() && !right.empty() && left.size() < refs.size() && right.size() < refs.size();
        }

        void medianSplit(const AABB& nodeBox, std::vector<PrimRef>& refs, std::vector<PrimRef>& left, std::vector<PrimRef>& right, AABB& lb, AABB& rb)
        {
            int axis = longestAxis(nodeBox.extent());
            size_t mid = refs.size() / 2;
            std::nth_element(refs.begin(), refs.begin() + mid, refs.end(), [axis](const PrimRef& x, const PrimRef& y) {
                return x.box.center()[axis] < y.box.center()[axis];
            });
            left.assign(refs.begin(), refs.begin() + mid);
            right.assign(refs.begin() + mid, refs.end());
            for (auto& r : left) lb.fit(r.box);
            for (auto& r : right) rb.fit(r.box);
        }

        struct Work {
            uint32_t node;
            uint32_t depth;
            std::vector<PrimRef> refs;
        };
        // the object-split builders work in place (as the reference's buildBVHInPlace, src/bvh/bvh_build.cpp:195-215): a node is a range of ONE array
        struct Range {
            uint32_t node, depth;
            size_t begin, end;
        };
        PrimRef* all = nullptr; // the references, and room of the same size for the right side of a partition (which keeps the order on both sides:
        PrimRef* spare = nullptr; // the tree is the one the list-based build made, byte for byte)

        void makeLeaf(uint32_t node, const PrimRef* refs, size_t count)
        {
            nodes[node].leftChildOrFirstTriangle = (uint32_t)leafRefs.size();
            nodes[node].triangleCount = (uint32_t)count;
            leafRefs.insert(leafRefs.end(), refs, refs + count);
            if (deferred)
                topLeaves.push_back({ node, (uint32_t)deferred->size() });
        }
        void makeLeaf(uint32_t node, const std::vector<PrimRef>& refs) { makeLeaf(node, refs.data(), refs.size()); }

        // A subtree set aside by the top phase of a parallel build: the slot of its root (already holding its box), and where the sequential build's arrays
        // stood when it would have been built -- everything it allocates comes right there, ahead of what the top phase allocated afterwards.
        struct Deferred {
            uint32_t slot, depth;
            std::vector<PrimRef> refs; // (spatial splits: a list of its own)
            size_t nodesBefore, leavesBefore; // sizes of the top phase's arrays at that moment
            size_t begin = 0, end = 0; // (object splits: its range of `all`)
        };
        std::vector<uint32_t> pairEpoch; // per pair of `nodes` (top phase): how many subtrees had been set aside when it was allocated
        std::vector<std::pair<uint32_t, uint32_t>> topLeaves; // (node, epoch) of the leaves the top phase made itself
        std::vector<Deferred>* deferred = nullptr;
        size_t deferBelow = 0;

        void build(std::vector<Work>& stack)
        {
            while (!stack.empty()) {
                Work w = std::move(stack.back());
                stack.pop_back();
                if (deferred && w.depth > 0 && w.refs.size() <= deferBelow && w.refs.size() > kLeafSize) { // a worker thread's share (buildParallel)
                    deferred->push_back({ w.node, w.depth, std::move(w.refs), nodes.size(), leafRefs.size() });
                    continue;
                }
                maxDepthSeen = std::max(maxDepthSeen, w.depth);
                const AABB nodeBox = loadBox(nodes[w.node]);
                const size_t n = w.refs.size();
                if (n <= kLeafSize || w.depth >= (uint32_t)kMaxBvhDepth) {
                    makeLeaf(w.node, w.refs);
                    continue;
                }

                std::vector<PrimRef> left, right;
                AABB leftBox, rightBox;
                bool split = false;
                if (w.depth >= (uint32_t)kMedianFallbackDepth) {
                    medianSplit(nodeBox, w.refs, left, right, leftBox, rightBox);
                    split = !left.empty() && !right.empty();
                } else {
                    const float leafCost = (float)n * kIntersectCost;
                    const float area = nodeBox.surfaceArea();
                    auto sah = [&](const Candidate& c) { return kTraversalCost + c.cost * kIntersectCost / area; };
                    static const int allAxes[3] = { 0, 1, 2 };
                    int one = longestAxis(nodeBox.extent());
                    Candidate obj = (kind == BvhBuilder::BinnedFast) ? findObjectSplit(nodeBox, w.refs.data(), n, &one, 1) : findObjectSplit(nodeBox, w.refs.data(), n, allAxes, 3);
                    bool objOk = obj.valid && area > 0.0f && sah(obj) < leafCost;
                    bool done = false;
                    if (kind == BvhBuilder::SpatialSplit && area > 0.0f) {
                        bool trySpatial = !objOk;
                        if (objOk) {
                            float overlap = obj.leftBox.intersection(obj.rightBox).surfaceArea();
                            trySpatial = overlap / rootArea > kSpatialAlpha;
                        }
                        if (trySpatial) {
                            Candidate sp = findSpatialSplit(nodeBox, w.refs);
                            if (sp.valid && sah(sp) < leafCost && (!objOk || sp.cost < obj.cost)) {
                                if (applySpatialSplit(nodeBox, sp, w.refs, left, right, leftBox, rightBox)) {
                                    split = done = true;
                                } else {
                                    left.clear();
                                    right.clear();
                                }
                            }
                        }
                    }
                    if (!done && objOk) {
                        applyObjectSplit(nodeBox, obj, w.refs, left, right);
                        leftBox = obj.leftBox;
                        rightBox = obj.rightBox;
                        split = !left.empty() && !right.empty();
                    }
                }
                if (!split) {
                    makeLeaf(w.node, w.refs);
                    continue;
                }
                uint32_t pair = allocPair();
                nodes[w.node].leftChildOrFirstTriangle = pair;
                nodes[w.node].triangleCount = 0;
                storeBox(nodes[pair], leftBox);
                storeBox(nodes[pair + 1], rightBox);
                w.refs.clear();
                w.refs.shrink_to_fit();
                stack.push_back({ pair, w.depth + 1, std::move(left) });
                stack.push_back({ pair + 1, w.depth + 1, std::move(right) });
            }
        }

        // binned object splits only (BinnedSAH, BinnedFast): the same decisions as build(), node by node and in the same order, on ranges
        void buildRanges(std::vector<Range>& stack)
        {
            while (!stack.empty()) {
                const Range w = stack.back();
                stack.pop_back();
                const size_t n = w.end - w.begin;
                if (deferred && w.depth > 0 && n <= deferBelow && n > kLeafSize) { // a worker thread's share (buildParallel)
                    deferred->push_back({ w.node, w.depth, {}, nodes.size(), leafRefs.size(), w.begin, w.end });
                    continue;
                }
                maxDepthSeen = std::max(maxDepthSeen, w.depth);
                const AABB nodeBox = loadBox(nodes[w.node]);
                PrimRef* refs = all + w.begin;
                if (n <= kLeafSize || w.depth >= (uint32_t)kMaxBvhDepth) {
                    makeLeaf(w.node, refs, n);
                    continue;
                }
                AABB leftBox, rightBox;
                size_t numLeft = 0;
                if (w.depth >= (uint32_t)kMedianFallbackDepth) {
                    const int axis = longestAxis(nodeBox.extent());
                    numLeft = n / 2;
                    std::nth_element(refs, refs + numLeft, refs + n, [axis](const PrimRef& x, const PrimRef& y) {
                        return x.box.center()[axis] < y.box.center()[axis];
                    });
                    for (size_t i = 0; i < numLeft; i++) leftBox.fit(refs[i].box);
                    for (size_t i = numLeft; i < n; i++) rightBox.fit(refs[i].box);
                } else {
                    const float area = nodeBox.surfaceArea();
                    static const int allAxes[3] = { 0, 1, 2 };
                    const int one = longestAxis(nodeBox.extent());
                    const Candidate obj = (kind == BvhBuilder::BinnedFast) ? findObjectSplit(nodeBox, refs, n, &one, 1) : findObjectSplit(nodeBox, refs, n, allAxes, 3);
                    if (obj.valid && area > 0.0f && kTraversalCost + obj.cost * kIntersectCost / area < (float)n * kIntersectCost) {
                        const float invWidth = (float)kObjectBins / nodeBox.extent()[obj.axis], lo = nodeBox.min[obj.axis];
                        PrimRef* right = spare + w.begin;
                        if (!obj.leftOfPart.empty()) {
                            // the workers' ranges again: each scatters its references to where the ranges before it leave off (left and right side: the
                            // order inside both sides is the sequential partition's), then copies its share back
                            const size_t parts = obj.leftOfPart.size(), chunk = obj.partChunk;
                            std::vector<size_t> leftAt(parts + 1, 0), rightAt(parts + 1, 0);
                            for (size_t p = 0; p < parts; p++) {
                                const size_t len = std::min((p + 1) * chunk, n) - std::min(p * chunk, n);
                                leftAt[p + 1] = leftAt[p] + obj.leftOfPart[p], rightAt[p + 1] = rightAt[p] + (len - obj.leftOfPart[p]);
                            }
                            numLeft = leftAt[parts];
                            WorkerPool& pool = WorkerPool::get();
                            pool.parallelFor(parts, 1, [&](size_t p0, size_t p1) {
                                for (size_t p = p0; p < p1; p++) {
                                    size_t l = leftAt[p], r = numLeft + rightAt[p];
                                    for (size_t i = std::min(p * chunk, n), e = std::min((p + 1) * chunk, n); i < e; i++)
                                        right[binOf(centerOn(refs[i].box, obj.axis), lo, invWidth, kObjectBins) < obj.plane ? l++ : r++] = refs[i];
                                }
                            });
                            pool.parallelFor(parts, 1, [&](size_t p0, size_t p1) {
                                const size_t b = std::min(p0 * chunk, n), e = std::min(p1 * chunk, n);
                                std::copy(right + b, right + e, refs + b);
                            });
                        } else {
                            size_t numRight = 0;
                            for (size_t i = 0; i < n; i++) {
                                if (binOf(centerOn(refs[i].box, obj.axis), lo, invWidth, kObjectBins) < obj.plane)
                                    refs[numLeft++] = refs[i];
                                else
                                    right[numRight++] = refs[i];
                            }
                            std::copy(right, right + numRight, refs + numLeft);
                        }
                        leftBox = obj.leftBox;
                        rightBox = obj.rightBox;
                    }
                }
                if (numLeft == 0 || numLeft == n) {
                    makeLeaf(w.node, refs, n);
                    continue;
                }
                const uint32_t pair = allocPair();
                nodes[w.node].leftChildOrFirstTriangle = pair;
                nodes[w.node].triangleCount = 0;
                storeBox(nodes[pair], leftBox);
                storeBox(nodes[pair + 1], rightBox);
                stack.push_back({ pair, w.depth + 1, w.begin, w.begin + numLeft });
                stack.push_back({ pair + 1, w.depth + 1, w.begin + numLeft, w.end });
            }
        }

        void run(std::vector<PrimRef>&& refs)
        {
            AABB rootBox;
            for (auto& r : refs) rootBox.fit(r.box);
            rootArea = rootBox.surfaceArea();
            // Large meshes: the top of the tree here, the subtrees below ~1/12 of the references on the worker threads, each into arrays of its own; the
            // result is put together in the order the sequential build allocates (a subtree's nodes and leaves are contiguous there: the build is depth
            // first), so the arrays are the same bytes whatever the thread count -- PTAMD_BUILD_THREADS=1 builds sequentially (tests compare the two).
            const size_t n = refs.size();
            const bool parallel = n >= 8192 && WorkerPool::get().threads() > 1 && !sequentialBuild();
            std::vector<Deferred> subtrees;
            if (parallel) {
                deferred = &subtrees;
                deferBelow = std::max<size_t>(1024, n / 12);
            }
            uint32_t root = allocPair();
            storeBox(nodes[root], rootBox);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<PrimRef> room;
            if (kind == BvhBuilder::SpatialSplit) {
                std::vector<Work> stack;
                stack.push_back({ root, 0, std::move(refs) });
                build(stack);
            } else {
                room.resize(n);
                all = refs.data(), spare = room.data();
                std::vector<Range> stack { { root, 0, 0, n } };
                buildRanges(stack);
            }
            const auto t1 = std::chrono::steady_clock::now();
            deferred = nullptr;
            if (!subtrees.empty())
                finishParallel(subtrees);
            if (std::getenv("PTAMD_BUILD_TIMING"))
                fprintf(stderr, "[ptamd_host] buildBVH: %zu references, top phase %.3f ms (%zu subtrees set aside), subtrees + assembly %.3f ms\n", n,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), subtrees.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        }

        void finishParallel(std::vector<Deferred>& subtrees)
        {
            const size_t numSub = subtrees.size();
            std::vector<Builder> local(numSub, Builder { verts, tris, kind });
            std::atomic<size_t> next { 0 };
            WorkerPool& pool = WorkerPool::get();
            pool.parallelFor(pool.threads(), 1, [&](size_t, size_t) {
                for (size_t k; (k = next.fetch_add(1)) < numSub;) {
                    Builder& b = local[k];
                    Deferred& d = subtrees[k];
                    const size_t count = kind == BvhBuilder::SpatialSplit ? d.refs.size() : d.end - d.begin;
                    b.rootArea = rootArea;
                    b.nodes.reserve(count + 2);
                    b.leafRefs.reserve(count + count / 4);
                    b.allocPair(); // [0]: a copy of the subtree's root slot, [1]: unused
                    b.nodes[0] = nodes[d.slot];
                    if (kind == BvhBuilder::SpatialSplit) {
                        std::vector<Work> stack;
                        stack.push_back({ 0u, d.depth, std::move(d.refs) });
                        b.build(stack);
                    } else { // (the subtrees' ranges of the two arrays are disjoint)
                        b.all = all, b.spare = spare;
                        std::vector<Range> stack { { 0u, d.depth, d.begin, d.end } };
                        b.buildRanges(stack);
                    }
                }
            });
            // where everything goes: a subtree's nodes behind what the top phase had allocated when it set the subtree aside, plus the earlier subtrees
            std::vector<size_t> nodesOfEarlier(numSub + 1, 0), leavesOfEarlier(numSub + 1, 0);
            for (size_t k = 0; k < numSub; k++) {
                nodesOfEarlier[k + 1] = nodesOfEarlier[k] + local[k].nodes.size() - 2;
                leavesOfEarlier[k + 1] = leavesOfEarlier[k] + local[k].leafRefs.size();
                maxDepthSeen = std::max(maxDepthSeen, local[k].maxDepthSeen);
            }
            std::vector<SubBVHNode> allNodes(nodes.size() + nodesOfEarlier[numSub]);
            std::vector<PrimRef> allLeaves(leafRefs.size() + leavesOfEarlier[numSub]);
            std::vector<uint8_t> isSubtreeRoot(nodes.size(), 0);
            for (const Deferred& d : subtrees)
                isSubtreeRoot[d.slot] = 1;
            auto topIndex = [&](uint32_t i) { return i + (uint32_t)nodesOfEarlier[pairEpoch[i / 2]]; };
            for (const auto& [node, epoch] : topLeaves) { // the top phase's own leaves
                SubBVHNode& nd = nodes[node];
                const uint32_t first = nd.leftChildOrFirstTriangle;
                std::copy(leafRefs.begin() + first, leafRefs.begin() + first + nd.triangleCount, allLeaves.begin() + first + leavesOfEarlier[epoch]);
                nd.leftChildOrFirstTriangle = first + (uint32_t)leavesOfEarlier[epoch];
            }
            for (uint32_t i = 0; i < nodes.size(); i++) {
                SubBVHNode nd = nodes[i];
                if (!isSubtreeRoot[i] && nd.triangleCount == 0 && nd.leftChildOrFirstTriangle != 0)
                    nd.leftChildOrFirstTriangle = topIndex(nd.leftChildOrFirstTriangle);
                allNodes[topIndex(i)] = nd;
            }
            next = 0;
            pool.parallelFor(pool.threads(), 1, [&](size_t, size_t) { // (disjoint ranges of the two arrays)
                for (size_t k; (k = next.fetch_add(1)) < numSub;) {
                    const Builder& b = local[k];
                    const size_t nodeBase = subtrees[k].nodesBefore + nodesOfEarlier[k], leafBase = subtrees[k].leavesBefore + leavesOfEarlier[k];
                    auto moved = [&](SubBVHNode nd) {
                        if (nd.triangleCount == 0)
                            nd.leftChildOrFirstTriangle = (uint32_t)(nodeBase + nd.leftChildOrFirstTriangle - 2);
                        else
                            nd.leftChildOrFirstTriangle = (uint32_t)(leafBase + nd.leftChildOrFirstTriangle);
                        return nd;
                    };
                    allNodes[topIndex(subtrees[k].slot)] = moved(b.nodes[0]);
                    for (size_t i = 2; i < b.nodes.size(); i++)
                        allNodes[nodeBase + i - 2] = moved(b.nodes[i]);
                    std::copy(b.leafRefs.begin(), b.leafRefs.end(), allLeaves.begin() + leafBase);
                }
            });
            nodes = std::move(allNodes);
            leafRefs = std::move(allLeaves);
        }
    };

} // namespace

BvhBuildResult buildBVH(const VertexSceneData* vertices, size_t numVertices, const TriangleSceneData* triangles, size_t numTriangles, BvhBuilder kind)
{
    if (numTriangles == 0)
        throw std::invalid_argument("buildBVH: empty mesh");
    std::vector<PrimRef> refs(numTriangles);
    const size_t perThread = sequentialBuild() || numTriangles < 8192 ? numTriangles : 1024; // (small meshes and PTAMD_BUILD_THREADS=1: on the caller)
    WorkerPool::get().parallelFor(numTriangles, perThread, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            for (int k = 0; k < 3; k++) {
                if (triangles[i].indices[k] >= numVertices)
                    throw std::invalid_argument("buildBVH: vertex index out of range");
                refs[i].box.fit(position(vertices[triangles[i].indices[k]]));
            }
            refs[i].prim = (uint32_t)i;
        }
    });
    Builder b { vertices, triangles, kind };
    b.nodes.reserve(numTriangles);
    b.leafRefs.reserve(numTriangles + numTriangles / 4);
    b.run(std::move(refs));

    BvhBuildResult out;
    out.rootNode = 0;
    out.nodes = std::move(b.nodes);
    out.triangles.resize(b.leafRefs.size());
    out.originalTriangle.resize(b.leafRefs.size());
    WorkerPool::get().parallelFor(b.leafRefs.size(), perThread, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            out.triangles[i] = triangles[b.leafRefs[i].prim];
            out.originalTriangle[i] = b.leafRefs[i].prim;
        }
    });
    return out;
}

void refitBVH(std::vector<SubBVHNode>& nodes, uint32_t root, const std::vector<TriangleSceneData>& triangles, const std::vector<VertexSceneData>& vertices)
{
    // post-order without recursion: children always have larger indices than their parent
    // (pairs are allocated after the parent), so a reverse index sweep visits children first.
    (void)root;
    for (size_t idx = nodes.size(); idx-- > 0;) {
        SubBVHNode& n = nodes[idx];
        if (idx == 1)
            continue; // pad
        AABB box;
        if (n.triangleCount != 0) {
            for (uint32_t t = 0; t < n.triangleCount; t++)
                for (int k = 0; k < 3; k++)
                    box.fit(position(vertices[triangles[n.leftChildOrFirstTriangle + t].indices[k]]));
        } else {
            if (n.leftChildOrFirstTriangle == 0)
                continue; // blank node
            box = loadBox(nodes[n.leftChildOrFirstTriangle]).merged(loadBox(nodes[n.leftChildOrFirstTriangle + 1]));
        }
        storeBox(n, box);
    }
}

BvhStats checkBVH(const BvhBuildResult& bvh, const VertexSceneData* vertices, size_t numInputTriangles, bool exactLeafContainment)
{
    BvhStats s;
    std::vector<uint8_t> seen(numInputTriangles, 0);
    // spatial splits: per input triangle the union of (leaf box x triangle box) over the leaves that reference it.  The clipped pieces add up
    // to the triangle, so that union has the triangle's own extent; a leaf whose box was shrunk (a doctored cache file: found by
    // tools/fuzz_loaders.py) leaves a side uncovered -- rays would pass the triangle there.
    std::vector<AABB> cover(exactLeafContainment ? 0 : numInputTriangles), triBox(exactLeafContainment ? 0 : numInputTriangles);
    struct Item {
        uint32_t node, depth;
    };
    std::vector<Item> stack { { bvh.rootNode, 0 } };
    const float eps = 0.0f;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        const SubBVHNode& n = bvh.nodes[it.node];
        AABB box = loadBox(n);
        s.numNodes++;
        s.maxDepth = std::max(s.maxDepth, it.depth);
        if (n.triangleCount != 0) {
            s.numLeaves++;
            s.numTriangleRefs += n.triangleCount;
            s.maxLeafSize = std::max(s.maxLeafSize, n.triangleCount);
            for (uint32_t t = 0; t < n.triangleCount; t++) {
                uint32_t ti = n.leftChildOrFirstTriangle + t;
                if (ti >= bvh.triangles.size()) {
                    s.trianglesInsideLeaves = false;
                    continue;
                }
                seen[bvh.originalTriangle[ti]] = 1;
                AABB tb;
                for (int k = 0; k < 3; k++)
                    tb.fit(position(vertices[bvh.triangles[ti].indices[k]]));
                if (exactLeafContainment) {
                    if (!box.contains(tb))
                        s.trianglesInsideLeaves = false;
                } else { // spatial splits hold clipped references: the triangle must at least overlap the leaf ...
                    AABB x = box.intersection(tb);
                    if (x.min.x > x.max.x + eps || x.min.y > x.max.y + eps || x.min.z > x.max.z + eps)
                        s.trianglesInsideLeaves = false;
                    else // ... and the leaves that reference it must, between them, reach every side of its box (checked below)
                        cover[bvh.originalTriangle[ti]].fit(x), triBox[bvh.originalTriangle[ti]] = tb;
                }
            }
        } else {
            uint32_t l = n.leftChildOrFirstTriangle;
            for (uint32_t c = l; c <= l + 1; c++) {
                if (!box.contains(loadBox(bvh.nodes[c])))
                    s.childrenInsideParents = false;
                stack.push_back({ c, it.depth + 1 });
            }
        }
    }
    for (uint8_t f : seen)
        if (!f)
            s.allTrianglesReferenced = false;
    for (size_t t = 0; t < cover.size(); t++)
        if (seen[t] && !(cover[t].min.x <= triBox[t].min.x && cover[t].min.y <= triBox[t].min.y && cover[t].min.z <= triBox[t].min.z
                         && cover[t].max.x >= triBox[t].max.x && cover[t].max.y >= triBox[t].max.y && cover[t].max.z >= triBox[t].max.z))
            s.trianglesInsideLeaves = false;
    return s;
}

} // namespace raytracer
