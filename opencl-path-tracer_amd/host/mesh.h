// Mesh = triangle soup + materials + bottom-level BVH.  Interface of the reference's IMesh
// (src/model/imesh.h:9-26) and Mesh (src/model/mesh.h); Assimp import is replaced by construction
// from arrays and a small ASCII/binary PLY reader (asset I/O is outside the hot path).
#pragma once
#include "aabb.h"
#include "bvh_build.h"
#include "image.h"
#include "material.h"
#include "transform.h"
#include <memory>
#include <string>
#include <vector>

namespace raytracer {

class IMesh {
public:
    virtual ~IMesh() = default;
    virtual const std::vector<VertexSceneData>& getVertices() const = 0;
    virtual const std::vector<TriangleSceneData>& getTriangles() const = 0; // leaf order
    virtual const std::vector<Material>& getMaterials() const = 0;
    virtual const std::vector<SubBVHNode>& getBvhNodes() const = 0;
    virtual const std::vector<uint32_t>& getEmissiveTriangles() const = 0;
    virtual AABB getBounds() const = 0;
    virtual bool isDynamic() const = 0;
    virtual uint32_t maxNumVertices() const = 0;
    virtual uint32_t maxNumTriangles() const = 0;
    virtual uint32_t maxNumMaterials() const = 0;
    virtual uint32_t maxNumBvhNodes() const = 0;
    virtual void buildBvh() = 0;
    virtual uint32_t getBvhRootNode() const = 0;
    // bumped whenever the arrays above change (a refit, a rebuild): what a flattened copy of them is held against.  An implementation that does not keep
    // count returns kUntracked: RayTracer::updateGeometry then takes such a mesh for changed whenever it isDynamic() (the reference's MeshSequence is such
    // a class: its arrays change under goToNextFrame, src/model/mesh_sequence.cpp:81-97) and hands the re-flattened arrays over as the reference does.
    static constexpr uint64_t kUntracked = ~0ull;
    virtual uint64_t generation() const { return kUntracked; }
};

class Mesh : public IMesh {
public:
    // positions: 3*numVertices floats; normals (3*nv) and texCoords (2*nv) may be null -- missing
    // normals are generated area-weighted per vertex; indices: 3*numTriangles; materialIndex per
    // triangle (null = 0).
    Mesh(const float* positions, const float* normals, const float* texCoords, size_t numVertices,
        const uint32_t* indices, const uint32_t* materialIndex, size_t numTriangles,
        const std::vector<Material>& materials, BvhBuilder builder = BvhBuilder::SpatialSplit, const std::string& bvhCacheFile = "");

    static std::shared_ptr<Mesh> fromPLY(const std::string& path, const Material& material, BvhBuilder builder = BvhBuilder::SpatialSplit);
    // Wavefront OBJ (+ MTL) the way the reference imports a model file (src/model/mesh.cpp:36-200, Assimp):
    // polygons are triangulated, points and lines dropped, identical (position, uv, normal) corners welded,
    // `offset` is baked into the positions and its inverse-transpose into the normals (mesh.cpp:76-82), missing
    // normals are generated smooth.  Materials: `overrideMaterial` for everything, else per `usemtl` group
    // Ke != 0 -> Material::Emissive(Ke), otherwise Material::Diffuse(Kd), or Material::Diffuse(textures->add(map_Kd file), Kd)
    // when the material has a diffuse map and a texture registry is given (mesh.cpp:51-70; paths relative to the OBJ's
    // folder).  `bvhCacheFile`: see storeBvh.
    static std::shared_ptr<Mesh> fromOBJ(const std::string& path, const Material* overrideMaterial = nullptr, const Transform& offset = Transform(),
        BvhBuilder builder = BvhBuilder::SpatialSplit, const std::string& bvhCacheFile = "", UniqueTextureFiles* textures = nullptr);

    const std::vector<VertexSceneData>& getVertices() const override { return m_vertices; }
    const std::vector<TriangleSceneData>& getTriangles() const override { return m_bvh.triangles; }
    const std::vector<Material>& getMaterials() const override { return m_materials; }
    const std::vector<SubBVHNode>& getBvhNodes() const override { return refittedBvh().nodes; }
    const std::vector<uint32_t>& getEmissiveTriangles() const override { return m_emissive; }
    AABB getBounds() const override { return m_bounds; }
    bool isDynamic() const override { return false; }
    uint32_t maxNumVertices() const override { return (uint32_t)m_vertices.size(); }
    uint32_t maxNumTriangles() const override { return (uint32_t)m_bvh.triangles.size(); }
    uint32_t maxNumMaterials() const override { return (uint32_t)m_materials.size(); }
    uint32_t maxNumBvhNodes() const override { return (uint32_t)m_bvh.nodes.size(); }
    void buildBvh() override {}
    uint32_t getBvhRootNode() const override { return m_bvh.rootNode; }

    // The reference's on-disk BVH cache (src/model/mesh.cpp:202-263), byte for byte: u32 version = 1, u32 root,
    // u32 numNodes, numNodes x 48-B SubBVHNode, u32 numTriangles, numTriangles x 16-B TriangleSceneData, '\n'.
    // A constructor given `bvhCacheFile` loads it when it exists AND validates (the reference trusts the file;
    // here sizes, indices, the triangle set and the BvhTester invariants are checked, and a file that fails is
    // rebuilt and overwritten) and stores the freshly built tree otherwise (mesh.cpp:175-196).
    void storeBvh(const std::string& fileName) const;
    bool loadBvh(const std::string& fileName);
    bool bvhFromCache() const { return m_bvhFromCache; }

    // new positions (3 * numVertices floats) and normals (null: regenerated smooth) for the same topology: the BVH is refitted.
    // The boxes are recomputed (refitBVH, bottom-up on one thread) only when somebody asks for the nodes: RayTracer::updateGeometry hands the
    // device library the vertices alone and the device refits its own copy of the tree (pt_refit_vertices), so a frame loop never pays for them here.
    void refit(const float* positions, const float* normals);
    bool boxesCurrent() const { return !m_boxesStale; }

    const BvhBuildResult& getBvh() const { return refittedBvh(); }
    size_t numInputTriangles() const { return m_inputTriangles.size(); }
    BvhBuilder builder() const { return m_builder; }
    uint64_t generation() const override { return m_generation; }

private:
    void generateSmoothNormals();
    const BvhBuildResult& refittedBvh() const;
    // vertex -> incident input triangles (once per corner, ascending): the smooth-normal pass gathers per vertex in the order the
    // scatter over the triangles would have added (same sums, bit for bit), which lets it run on several threads
    std::vector<uint32_t> m_cornerStart, m_cornerTri;
    std::vector<vec3> m_faceNormal; // scratch of generateSmoothNormals
    mutable bool m_boxesStale = false; // the nodes' boxes are older than the vertices (refit): recomputed by refittedBvh()
    std::vector<VertexSceneData> m_vertices;
    std::vector<TriangleSceneData> m_inputTriangles;
    std::vector<Material> m_materials;
    std::vector<uint32_t> m_emissive;
    mutable BvhBuildResult m_bvh;
    AABB m_bounds;
    BvhBuilder m_builder;
    bool m_bvhFromCache = false;
    uint64_t m_generation = 0;
};

} // namespace raytracer
