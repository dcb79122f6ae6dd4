// Scene graph with mesh de-duplication for instancing (API of the reference's src/scene.h:13-52),
// the top-level BVH builder (src/bvh/top_bvh_build.h:12) and the flattening step that produces the
// device arrays (src/raytracer.cpp:201-287, 497-621).
#pragma once
#include "mesh.h"
#include "transform.h"
#include <memory>
#include <optional>
#include <unordered_map>
#include <vector>

namespace raytracer {

struct SceneNode {
    const SceneNode* parent = nullptr;
    std::vector<std::unique_ptr<SceneNode>> children;
    AABB bounds; // object-space bounds of the attached mesh
    Transform transform;
    std::optional<uint32_t> meshID;
    std::optional<uint32_t> subBvhRootID;
};

struct MeshBvhPair {
    std::shared_ptr<IMesh> meshPtr;
    uint32_t bvhIndexOffset; // offset of this mesh's nodes in the global sub-BVH array
    uint32_t vertexIndexOffset = 0; // ... of its vertices in the global vertex array (what pt_refit_vertices addresses a deformed mesh by)
    uint64_t uploadedGeneration = 0; // IMesh::generation() the device library last saw (RayTracer::updateGeometry)
};

class Scene {
public:
    SceneNode& addNode(const std::shared_ptr<IMesh>& mesh, const Transform& transform = {}, SceneNode* parent = nullptr);
    SceneNode& getRootNode() { return m_root; }
    const SceneNode& getRootNode() const { return m_root; }
    std::vector<MeshBvhPair>& getMeshes() { return m_meshes; }
    const std::vector<MeshBvhPair>& getMeshes() const { return m_meshes; }
    size_t numInstances() const { return m_numInstances; }

private:
    SceneNode m_root;
    std::unordered_map<const IMesh*, uint32_t> m_meshIds;
    std::vector<MeshBvhPair> m_meshes;
    size_t m_numInstances = 0;
};

// Top-level BVH over the scene-graph nodes that carry a mesh: one leaf per instance with
// world-space bounds, inverse(world) and the global sub-BVH root; leaves are merged by greedy
// agglomerative clustering on merged surface area (Walter et al. 2008, as the reference's
// src/bvh/top_bvh_build.cpp:16-68).  The root is the LAST node.
struct TopBvhBuildResult {
    uint32_t rootNode = 0;
    std::vector<TopBVHNode> nodes;
};
// more instances than this: a top-down SAH build instead of the reference's O(n^2) agglomerative clustering (scene.cpp)
constexpr size_t kAgglomerativeMaxInstances = 256;
TopBvhBuildResult buildTopBVH(const SceneNode& root, const std::vector<uint32_t>& meshBvhOffsets);

// Everything pt_upload_static / pt_upload_dynamic consume.
struct FlattenedScene {
    std::vector<VertexSceneData> vertices;
    std::vector<TriangleSceneData> triangles; // global vertex / material indices
    std::vector<pt_material> materials;
    std::vector<SubBVHNode> subBvhNodes; // global triangle / node indices
    std::vector<pt_emissive_triangle> emissiveTriangles; // world space
    std::vector<TopBVHNode> topBvhNodes;
    uint32_t topBvhRoot = 0;
};
constexpr uint32_t kMaxNumLights = 256; // MAX_NUM_LIGHTS, src/raytracer.cpp:39

void flattenStatic(Scene& scene, FlattenedScene& out); // also records MeshBvhPair::bvhIndexOffset
void flattenDynamic(const Scene& scene, FlattenedScene& out); // lights + top-level BVH
inline FlattenedScene flattenScene(Scene& scene)
{
    FlattenedScene f;
    flattenStatic(scene, f);
    flattenDynamic(scene, f);
    return f;
}

} // namespace raytracer
