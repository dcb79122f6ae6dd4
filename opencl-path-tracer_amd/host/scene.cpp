#include "scene.h"
#include "parallel.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>

namespace raytracer {

SceneNode& Scene::addNode(const std::shared_ptr<IMesh>& mesh, const Transform& transform, SceneNode* parent)
{
    if (!parent)
        parent = &m_root;
    uint32_t id;
    auto it = m_meshIds.find(mesh.get());
    if (it != m_meshIds.end()) {
        id = it->second;
    } else {
        id = (uint32_t)m_meshes.size();
        m_meshes.push_back({ mesh, 0, 0, 0 });
        m_meshIds.emplace(mesh.get(), id);
    }
    auto node = std::make_unique<SceneNode>();
    node->parent = parent;
    node->bounds = mesh->getBounds();
    node->transform = transform;
    node->meshID = id;
    node->subBvhRootID = mesh->getBvhRootNode();
    parent->bounds.fit(mesh->getBounds());
    parent->children.push_back(std::move(node));
    m_numInstances++;
    return *parent->children.back();
}

namespace {
    void setBox(TopBVHNode& n, const AABB& b)
    {
        n.min[0] = b.min.x, n.min[1] = b.min.y, n.min[2] = b.min.z, n.min[3] = 0;
        n.max[0] = b.max.x, n.max[1] = b.max.y, n.max[2] = b.max.z, n.max[3] = 0;
    }
    AABB getBox(const TopBVHNode& n) { return { { n.min[0], n.min[1], n.min[2] }, { n.max[0], n.max[1], n.max[2] } }; }

    AABB transformedBounds(const AABB& b, const mat4& m)
    {
        AABB r;
        for (int corner = 0; corner < 8; corner++) {
            vec3 p(corner & 1 ? b.max.x : b.min.x, corner & 2 ? b.max.y : b.min.y, corner & 4 ? b.max.z : b.min.z);
            r.fit((m * vec4(p, 1.0f)).xyz());
        }
        return r;
    }

    template <typename F>
    void walk(const SceneNode& node, const mat4& parentWorld, F&& visit)
    {
        mat4 world = parentWorld * node.transform.matrix();
        if (node.meshID)
            visit(node, world);
        for (const auto& c : node.children)
            walk(*c, world, visit);
    }
}

namespace {
    // the instances in scene-graph order (the walk carries the world matrices down the graph): made once per flattenDynamic, for the lights and the top level
    struct Instance {
        const SceneNode* node;
        mat4 world;
    };
    std::vector<Instance> collectInstances(const SceneNode& root)
    {
        std::vector<Instance> instances;
        walk(root, mat4(), [&](const SceneNode& n, const mat4& world) { instances.push_back({ &n, world }); });
        return instances;
    }
    TopBvhBuildResult buildTopBVHOver(const std::vector<Instance>& instances, const std::vector<uint32_t>& meshBvhOffsets);
}

TopBvhBuildResult buildTopBVH(const SceneNode& root, const std::vector<uint32_t>& meshBvhOffsets) { return buildTopBVHOver(collectInstances(root), meshBvhOffsets); }

namespace {
TopBvhBuildResult buildTopBVHOver(const std::vector<Instance>& instances, const std::vector<uint32_t>& meshBvhOffsets)
{
    TopBvhBuildResult out;
    std::vector<uint32_t> active; // cluster roots still to be merged
    // every leaf on its own: world bounds, the inverse of the world matrix -- ten thousand instances moved per tick are ten thousand 4 x 4 inversions: on the
    // worker pool above 512 instances
    const bool timing = std::getenv("PTAMD_BUILD_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now(), t1 = t0;
    out.nodes.resize(instances.size());
    active.resize(instances.size());
    WorkerPool::get().parallelFor(instances.size(), 256, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            const SceneNode& n = *instances[i].node;
            TopBVHNode leaf;
            std::memset(&leaf, 0, sizeof(leaf));
            setBox(leaf, transformedBounds(n.bounds, instances[i].world));
            mat4 inv = inverse(instances[i].world);
            std::memcpy(leaf.invTransform, inv.data(), sizeof(leaf.invTransform));
            leaf.a = *n.subBvhRootID + meshBvhOffsets[*n.meshID];
            leaf.b = 0;
            leaf.isLeaf = 1;
            active[i] = (uint32_t)i;
            out.nodes[i] = leaf;
        }
    });
    if (active.empty())
        throw std::invalid_argument("buildTopBVH: scene has no mesh instances");
    const auto t2 = std::chrono::steady_clock::now();
    struct Report {
        bool on;
        std::chrono::steady_clock::time_point t0, t1, t2;
        size_t n;
        ~Report()
        {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            if (on)
                fprintf(stderr, "[ptamd_host] buildTopBVH: %zu instances, leaves %.3f ms, tree %.3f\n", n, ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
        }
    } report { timing, t0, t1, t2, instances.size() };
    // The agglomerative clustering below is the reference's (top_bvh_build.cpp:42-93): every merge searches ALL clusters still active
    // for a best partner -- O(n) per search, O(n^2) and worse per tree: 0.03 ms for 14 instances, 16 ms for 1 000, 240 ms for 4 000, every
    // tick.  It is kept where it is cheap (the reference's scenes hold a handful of instances: the identical tree); larger scenes get a
    // top-down build over the instance boxes -- 16-bin SAH on the centroids along the longest axis, median where the bins do not separate
    // -- which is O(n log n).  Either tree is just as valid an input to the traversal (inner nodes after their children, root last).
    if (active.size() > kAgglomerativeMaxInstances) {
        std::vector<uint32_t> order = active;
        // the leaves' boxes and centroids side by side (a TopBVHNode is 112 bytes, most of it the matrix): what the build reads thirteen times per leaf
        std::vector<AABB> leafBox(order.size());
        std::vector<vec3> leafCentre(order.size());
        for (size_t i = 0; i < order.size(); i++)
            leafBox[i] = getBox(out.nodes[i]), leafCentre[i] = leafBox[i].center();
        // A SAH split that peels one box off per level (geometrically spaced instances) makes a tree as deep as the instance count, which the
        // device library then refuses for its traversal stack -- a balanced tree over the same boxes would do.  Lop-sided splits (fewer than an
        // eighth on one side) are only taken while the depth stays within 2 log2(n); beyond that the median decides.
        uint32_t depthBudget = 2;
        for (size_t n = order.size(); n > 1; n >>= 1)
            depthBudget += 2;
        // A subtree over `count` leaves has count - 1 inner nodes, and the recursion this replaces pushed them in post-order (left subtree, right subtree,
        // parent): the subtree whose first inner node is `base` owns [base, base + count - 1) and its root is the last of them -- every node's index is
        // known before anything below it is built, so the subtrees below ~1/32 of the instances are built side by side on the worker pool and the
        // array is the sequential build's byte for byte.
        struct Item {
            uint32_t first, count, depth, base;
        };
        const uint32_t numLeaves = (uint32_t)order.size();
        out.nodes.resize((size_t)numLeaves * 2 - 1);
        auto rootIndex = [&](const Item& it) { return it.count == 1 ? order[it.first] : it.base + it.count - 2; };
        auto splitOne = [&](const Item& r, Item& left, Item& right) {
            const uint32_t depth = r.depth;
            AABB cb, nb;
            for (uint32_t k = 0; k < r.count; k++) {
                cb.fit(leafCentre[order[r.first + k]]);
                nb.fit(leafBox[order[r.first + k]]);
            }
            const vec3 ext = cb.extent();
            const int axis = ext.x >= ext.y ? (ext.x >= ext.z ? 0 : 2) : (ext.y >= ext.z ? 1 : 2);
            auto comp = [&](const vec3& v) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); };
            const float lo = comp(cb.min), width = comp(ext);
            uint32_t mid = r.count / 2;
            bool split = false;
            if (width > 0.0f && r.count > 4) {
                constexpr int kBins = 16;
                AABB binBox[kBins];
                uint32_t binCount[kBins] = {};
                auto binOf = [&](uint32_t node) { return std::min(kBins - 1, (int)((comp(leafCentre[node]) - lo) / width * kBins)); };
                for (uint32_t k = 0; k < r.count; k++) {
                    const int b = binOf(order[r.first + k]);
                    binBox[b].fit(leafBox[order[r.first + k]]);
                    binCount[b]++;
                }
                float rightArea[kBins];
                AABB acc;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.fit(binBox[b]);
                    rightArea[b] = acc.surfaceArea();
                }
                acc = AABB();
                uint32_t nLeft = 0;
                float best = std::numeric_limits<float>::max();
                int bestBin = -1;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.fit(binBox[b]);
                    nLeft += binCount[b];
                    if (nLeft == 0 || nLeft == r.count)
                        continue;
                    const float cost = acc.surfaceArea() * (float)nLeft + rightArea[b + 1] * (float)(r.count - nLeft);
                    if (cost < best)
                        best = cost, bestBin = b;
                }
                if (bestBin >= 0) {
                    auto it = std::partition(order.begin() + r.first, order.begin() + r.first + r.count, [&](uint32_t node) { return binOf(node) <= bestBin; });
                    mid = (uint32_t)(it - (order.begin() + r.first));
                    split = mid > 0 && mid < r.count;
                    if (split && depth > depthBudget / 2 && std::min(mid, r.count - mid) < r.count / 8)
                        split = false; // deep already, and this split makes little progress: the median instead
                }
            }
            if (!split) { // all centroids in one place (or a handful of boxes): the median along the axis
                mid = r.count / 2;
                std::nth_element(order.begin() + r.first, order.begin() + r.first + mid, order.begin() + r.first + r.count,
                    [&](uint32_t x, uint32_t y) { return comp(leafCentre[x]) < comp(leafCentre[y]); });
            }
            left = { r.first, mid, depth + 1, r.base };
            right = { r.first + mid, r.count - mid, depth + 1, r.base + (mid - 1) };
            TopBVHNode inner;
            std::memset(&inner, 0, sizeof(inner));
            setBox(inner, nb);
            mat4 identity;
            std::memcpy(inner.invTransform, identity.data(), sizeof(inner.invTransform));
            inner.a = rootIndex(left);
            inner.b = rootIndex(right);
            inner.isLeaf = 0;
            out.nodes[r.base + r.count - 2] = inner;
        };
        const uint32_t grain = std::max<uint32_t>(256, numLeaves / 32);
        std::vector<Item> tasks;
        auto buildBelow = [&](Item top, std::vector<Item>* setAside) {
            std::vector<Item> stack { top };
            while (!stack.empty()) {
                const Item it = stack.back();
                stack.pop_back();
                if (it.count == 1)
                    continue;
                if (setAside && it.depth > 0 && it.count <= grain) {
                    setAside->push_back(it);
                    continue;
                }
                Item l, r;
                splitOne(it, l, r);
                stack.push_back(r);
                stack.push_back(l);
            }
        };
        const Item all { 0, numLeaves, 0, numLeaves };
        WorkerPool& pool = WorkerPool::get();
        const bool pooled = pool.threads() > 1 && numLeaves >= 2048;
        buildBelow(all, pooled ? &tasks : nullptr);
        if (!tasks.empty()) {
            std::atomic<size_t> next { 0 };
            pool.parallelFor(pool.threads(), 1, [&](size_t, size_t) {
                for (size_t t; (t = next.fetch_add(1)) < tasks.size();)
                    buildBelow(tasks[t], nullptr);
            });
        }
        out.rootNode = rootIndex(all);
        return out;
    }

    auto mergedArea = [&](uint32_t x, uint32_t y) { return getBox(out.nodes[x]).merged(getBox(out.nodes[y])).surfaceArea(); };
    auto bestPartner = [&](uint32_t x) {
        uint32_t best = x;
        float bestArea = std::numeric_limits<float>::max();
        for (uint32_t y : active) {
            if (y == x)
                continue;
            float a = mergedArea(x, y);
            if (a < bestArea) {
                bestArea = a;
                best = y;
            }
        }
        return best;
    };
    // mutual-nearest-neighbour chain: follow best partners until A's partner's partner is A
    uint32_t a = active.back();
    while (active.size() > 1) {
        uint32_t b = bestPartner(a);
        uint32_t c = bestPartner(b);
        if (c != a) {
            a = b;
            continue;
        }
        TopBVHNode inner;
        std::memset(&inner, 0, sizeof(inner));
        setBox(inner, getBox(out.nodes[a]).merged(getBox(out.nodes[b])));
        mat4 identity;
        std::memcpy(inner.invTransform, identity.data(), sizeof(inner.invTransform));
        inner.a = a;
        inner.b = b;
        inner.isLeaf = 0;
        active.erase(std::remove_if(active.begin(), active.end(), [&](uint32_t v) { return v == a || v == b; }), active.end());
        a = (uint32_t)out.nodes.size();
        out.nodes.push_back(inner);
        active.push_back(a);
    }
    out.rootNode = (uint32_t)out.nodes.size() - 1;
    return out;
}

} // namespace

void flattenStatic(Scene& scene, FlattenedScene& out)
{
    size_t numV = 0, numM = 0, numT = 0, numN = 0;
    for (const MeshBvhPair& pair : scene.getMeshes()) {
        const IMesh& mesh = *pair.meshPtr;
        numV += mesh.getVertices().size(), numM += mesh.getMaterials().size(), numT += mesh.getTriangles().size(), numN += mesh.getBvhNodes().size();
    }
    out.vertices.resize(numV);
    out.materials.resize(numM);
    out.triangles.resize(numT);
    out.subBvhNodes.resize(numN);
    uint32_t v0 = 0, m0 = 0, t0 = 0, n0 = 0;
    for (MeshBvhPair& pair : scene.getMeshes()) {
        const IMesh& mesh = *pair.meshPtr;
        std::copy(mesh.getVertices().begin(), mesh.getVertices().end(), out.vertices.begin() + v0);
        size_t i = m0;
        for (const Material& m : mesh.getMaterials())
            out.materials[i++] = m;
        i = t0;
        for (TriangleSceneData t : mesh.getTriangles()) {
            t.indices[0] += v0, t.indices[1] += v0, t.indices[2] += v0;
            t.materialIndex += m0;
            out.triangles[i++] = t;
        }
        i = n0;
        for (SubBVHNode n : mesh.getBvhNodes()) {
            n.leftChildOrFirstTriangle += (n.triangleCount > 0) ? t0 : n0;
            out.subBvhNodes[i++] = n;
        }
        pair.bvhIndexOffset = n0;
        pair.vertexIndexOffset = v0;
        v0 += (uint32_t)mesh.getVertices().size(), m0 += (uint32_t)mesh.getMaterials().size(), t0 += (uint32_t)mesh.getTriangles().size(), n0 += (uint32_t)mesh.getBvhNodes().size();
    }
    // A mesh whose arrays change under the scene (isDynamic(): the reference's MeshSequence rebuilds or refits its tree in buildBvh, src/model/mesh_sequence.cpp:81-97)
    // may have moved its root and its bounds since addNode looked at them: the nodes that carry it take the current ones, so that the top level built from
    // them (flattenDynamic) encloses the frame that is about to be rendered.
    struct Refresh {
        static void walk(SceneNode& node, const std::vector<MeshBvhPair>& meshes)
        {
            if (node.meshID && meshes[*node.meshID].meshPtr->isDynamic()) {
                node.bounds = meshes[*node.meshID].meshPtr->getBounds();
                node.subBvhRootID = meshes[*node.meshID].meshPtr->getBvhRootNode();
            }
            for (auto& c : node.children)
                walk(*c, meshes);
        }
    };
    Refresh::walk(scene.getRootNode(), scene.getMeshes());
}

void flattenDynamic(const Scene& scene, FlattenedScene& out)
{
    out.emissiveTriangles.clear();
    const std::vector<Instance> instances = collectInstances(scene.getRootNode());
    for (const Instance& inst : instances) {
        const SceneNode& n = *inst.node;
        const mat4& world = inst.world;
        const IMesh& mesh = *scene.getMeshes()[*n.meshID].meshPtr;
        for (uint32_t ti : mesh.getEmissiveTriangles()) {
            if (out.emissiveTriangles.size() >= kMaxNumLights)
                break;
            const TriangleSceneData& tri = mesh.getTriangles()[ti];
            pt_emissive_triangle e;
            std::memset(&e, 0, sizeof(e));
            for (int k = 0; k < 3; k++) {
                const float* p = mesh.getVertices()[tri.indices[k]].vertex;
                vec4 w = world * vec4(p[0], p[1], p[2], 1.0f);
                e.vertices[k][0] = w.x, e.vertices[k][1] = w.y, e.vertices[k][2] = w.z, e.vertices[k][3] = w.w;
            }
            e.material = mesh.getMaterials()[tri.materialIndex];
            out.emissiveTriangles.push_back(e);
        }
    }
    std::vector<uint32_t> offsets;
    for (const MeshBvhPair& p : scene.getMeshes())
        offsets.push_back(p.bvhIndexOffset);
    TopBvhBuildResult top = buildTopBVHOver(instances, offsets);
    out.topBvhNodes = std::move(top.nodes);
    out.topBvhRoot = top.rootNode;
}

} // namespace raytracer
