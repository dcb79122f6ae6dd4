#include "scene.h"
#include <algorithm>
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

TopBvhBuildResult buildTopBVH(const SceneNode& root, const std::vector<uint32_t>& meshBvhOffsets)
{
    TopBvhBuildResult out;
    std::vector<uint32_t> active; // cluster roots still to be merged
    walk(root, mat4(), [&](const SceneNode& n, const mat4& world) {
        TopBVHNode leaf;
        std::memset(&leaf, 0, sizeof(leaf));
        setBox(leaf, transformedBounds(n.bounds, world));
        mat4 inv = inverse(world);
        std::memcpy(leaf.invTransform, inv.data(), sizeof(leaf.invTransform));
        leaf.a = *n.subBvhRootID + meshBvhOffsets[*n.meshID];
        leaf.b = 0;
        leaf.isLeaf = 1;
        active.push_back((uint32_t)out.nodes.size());
        out.nodes.push_back(leaf);
    });
    if (active.empty())
        throw std::invalid_argument("buildTopBVH: scene has no mesh instances");
    // The agglomerative clustering below is the reference's (top_bvh_build.cpp:42-93): every merge searches ALL clusters still active
    // for a best partner -- O(n) per search, O(n^2) and worse per tree: 0.03 ms for 14 instances, 16 ms for 1 000, 240 ms for 4 000, every
    // tick.  It is kept where it is cheap (the reference's scenes hold a handful of instances: the identical tree); larger scenes get a
    // top-down build over the instance boxes -- 16-bin SAH on the centroids along the longest axis, median where the bins do not separate
    // -- which is O(n log n).  Either tree is just as valid an input to the traversal (inner nodes after their children, root last).
    if (active.size() > kAgglomerativeMaxInstances) {
        struct Range {
            uint32_t first, count;
        };
        std::vector<uint32_t> order = active;
        // A SAH split that peels one box off per level (geometrically spaced instances) makes a tree as deep as the instance count, which the
        // device library then refuses for its traversal stack -- a balanced tree over the same boxes would do.  Lop-sided splits (fewer than an
        // eighth on one side) are only taken while the depth stays within 2 log2(n); beyond that the median decides.
        uint32_t depthBudget = 2;
        for (size_t n = order.size(); n > 1; n >>= 1)
            depthBudget += 2;
        std::function<uint32_t(Range, uint32_t)> build = [&](Range r, uint32_t depth) -> uint32_t {
            if (r.count == 1)
                return order[r.first];
            AABB cb, nb;
            for (uint32_t k = 0; k < r.count; k++) {
                const AABB b = getBox(out.nodes[order[r.first + k]]);
                cb.fit(b.center());
                nb.fit(b);
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
                auto binOf = [&](uint32_t node) { return std::min(kBins - 1, (int)((comp(getBox(out.nodes[node]).center()) - lo) / width * kBins)); };
                for (uint32_t k = 0; k < r.count; k++) {
                    const int b = binOf(order[r.first + k]);
                    binBox[b].fit(getBox(out.nodes[order[r.first + k]]));
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
                    [&](uint32_t x, uint32_t y) { return comp(getBox(out.nodes[x]).center()) < comp(getBox(out.nodes[y]).center()); });
            }
            const uint32_t l = build({ r.first, mid }, depth + 1), rr = build({ r.first + mid, r.count - mid }, depth + 1);
            TopBVHNode inner;
            std::memset(&inner, 0, sizeof(inner));
            setBox(inner, nb);
            mat4 identity;
            std::memcpy(inner.invTransform, identity.data(), sizeof(inner.invTransform));
            inner.a = l;
            inner.b = rr;
            inner.isLeaf = 0;
            out.nodes.push_back(inner);
            return (uint32_t)out.nodes.size() - 1;
        };
        out.rootNode = build({ 0, (uint32_t)order.size() }, 0);
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
    walk(scene.getRootNode(), mat4(), [&](const SceneNode& n, const mat4& world) {
        const IMesh& mesh = *scene.getMeshes()[*n.meshID].meshPtr;
        for (uint32_t ti : mesh.getEmissiveTriangles()) {
            if (out.emissiveTriangles.size() >= kMaxNumLights)
                return;
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
    });
    std::vector<uint32_t> offsets;
    for (const MeshBvhPair& p : scene.getMeshes())
        offsets.push_back(p.bvhIndexOffset);
    TopBvhBuildResult top = buildTopBVH(scene.getRootNode(), offsets);
    out.topBvhNodes = std::move(top.nodes);
    out.topBvhRoot = top.rootNode;
}

} // namespace raytracer
