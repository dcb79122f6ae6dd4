// C entry points over the host-side scene library (meshes, BVH builders, scene graph, camera) so
// that non-C++ callers -- the Python parity tests and bench.py -- can build the flattened scene
// arrays that pt_upload_static / pt_upload_dynamic consume.  Declared in include/ptamd_host.h.
#include "../../include/ptamd_host.h"
#include "camera.h"
#include "image.h"
#include "scene.h"
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

using namespace raytracer;

namespace {
thread_local std::string g_error;

struct MeshHandle {
    std::shared_ptr<Mesh> mesh;
};
struct SceneHandle {
    Scene scene;
    std::vector<SceneNode*> nodes; // by id
    FlattenedScene flat;
    bool flattened = false; // `flat` is current
    bool staticFlattened = false; // its static arrays are (a moved node leaves them valid; a new node does not)
    std::vector<uint64_t> meshGenerations; // IMesh::generation() of every mesh when the static arrays were made (a refit leaves them stale)
    std::vector<uint64_t> generationsNow() const
    {
        std::vector<uint64_t> g;
        for (const MeshBvhPair& p : scene.getMeshes())
            g.push_back(p.meshPtr->generation());
        return g;
    }
};

template <typename F>
int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
}

extern "C" {

const char* pth_last_error(void) { return g_error.c_str(); }

pth_mesh* pth_mesh_create(const float* positions, const float* normals, const float* texCoords, size_t numVertices,
    const uint32_t* indices, const uint32_t* materialIndex, size_t numTriangles, const pt_material* materials,
    size_t numMaterials, int builder)
{
    MeshHandle* h = nullptr;
    int rc = guarded([&] {
        std::vector<Material> mats(numMaterials);
        for (size_t i = 0; i < numMaterials; i++)
            static_cast<pt_material&>(mats[i]) = materials[i];
        auto m = std::make_shared<Mesh>(positions, normals, texCoords, numVertices, indices, materialIndex, numTriangles, mats, (BvhBuilder)builder);
        h = new MeshHandle { m };
    });
    return rc == 0 ? (pth_mesh*)h : nullptr;
}

pth_mesh* pth_mesh_create_cached(const float* positions, const float* normals, const float* texCoords, size_t numVertices,
    const uint32_t* indices, const uint32_t* materialIndex, size_t numTriangles, const pt_material* materials,
    size_t numMaterials, int builder, const char* bvhCacheFile)
{
    MeshHandle* h = nullptr;
    int rc = guarded([&] {
        std::vector<Material> mats(numMaterials);
        for (size_t i = 0; i < numMaterials; i++)
            static_cast<pt_material&>(mats[i]) = materials[i];
        auto m = std::make_shared<Mesh>(positions, normals, texCoords, numVertices, indices, materialIndex, numTriangles, mats, (BvhBuilder)builder,
            bvhCacheFile ? std::string(bvhCacheFile) : std::string());
        h = new MeshHandle { m };
    });
    return rc == 0 ? (pth_mesh*)h : nullptr;
}

int pth_mesh_store_bvh(const pth_mesh* m, const char* path)
{
    return guarded([&] { ((const MeshHandle*)m)->mesh->storeBvh(path); });
}

int pth_mesh_bvh_from_cache(const pth_mesh* m) { return m && ((const MeshHandle*)m)->mesh->bvhFromCache() ? 1 : 0; }

pth_mesh* pth_mesh_from_ply(const char* path, const pt_material* material, int builder)
{
    MeshHandle* h = nullptr;
    int rc = guarded([&] {
        Material m;
        static_cast<pt_material&>(m) = *material;
        h = new MeshHandle { Mesh::fromPLY(path, m, (BvhBuilder)builder) };
    });
    return rc == 0 ? (pth_mesh*)h : nullptr;
}

pth_texture_files* pth_texture_files_create(void) { return (pth_texture_files*)new UniqueTextureFiles(); }
void pth_texture_files_destroy(pth_texture_files* t) { delete (UniqueTextureFiles*)t; }
int pth_texture_files_add(pth_texture_files* t, const char* path, int isLinear, float brightnessMultiplier)
{
    int id = -1;
    guarded([&] { id = ((UniqueTextureFiles*)t)->add(path, isLinear != 0, brightnessMultiplier); });
    return id;
}
int pth_texture_files_count(const pth_texture_files* t) { return t ? (int)((const UniqueTextureFiles*)t)->files().size() : 0; }
const char* pth_texture_files_path(const pth_texture_files* t, int index, int* isLinear, float* brightnessMultiplier)
{
    const auto& files = ((const UniqueTextureFiles*)t)->files();
    if (!t || index < 0 || (size_t)index >= files.size())
        return nullptr;
    if (isLinear)
        *isLinear = files[index].isLinear ? 1 : 0;
    if (brightnessMultiplier)
        *brightnessMultiplier = files[index].brightnessMultiplier;
    return files[index].path.c_str();
}

pth_mesh* pth_mesh_from_obj(const char* path, const pt_material* overrideMaterial, const float location[3], const float orientation_wxyz[4],
    const float scale[3], int builder, const char* bvhCacheFile)
{
    return pth_mesh_from_obj_textured(path, overrideMaterial, location, orientation_wxyz, scale, builder, bvhCacheFile, nullptr);
}

pth_mesh* pth_mesh_from_obj_textured(const char* path, const pt_material* overrideMaterial, const float location[3], const float orientation_wxyz[4],
    const float scale[3], int builder, const char* bvhCacheFile, pth_texture_files* textures)
{
    MeshHandle* h = nullptr;
    int rc = guarded([&] {
        Material m;
        if (overrideMaterial)
            static_cast<pt_material&>(m) = *overrideMaterial;
        Transform t;
        if (location)
            t.location = vec3(location[0], location[1], location[2]);
        if (orientation_wxyz)
            t.orientation = quat { orientation_wxyz[0], orientation_wxyz[1], orientation_wxyz[2], orientation_wxyz[3] };
        if (scale)
            t.scale = vec3(scale[0], scale[1], scale[2]);
        h = new MeshHandle { Mesh::fromOBJ(path, overrideMaterial ? &m : nullptr, t, (BvhBuilder)builder, bvhCacheFile ? std::string(bvhCacheFile) : std::string(),
            (UniqueTextureFiles*)textures) };
    });
    return rc == 0 ? (pth_mesh*)h : nullptr;
}

int pth_image_hdr_info(const char* path, uint32_t* width, uint32_t* height)
{
    return guarded([&] {
        const ImageRGBAF img = loadRadianceHDR(path);
        *width = img.width, *height = img.height;
    });
}

int pth_image_load_hdr(const char* path, uint32_t width, uint32_t height, float brightnessMultiplier, float* rgba_out)
{
    return guarded([&] {
        const ImageRGBAF img = loadSkydomeLayer(path, width, height, brightnessMultiplier);
        std::memcpy(rgba_out, img.rgba.data(), img.rgba.size() * sizeof(float));
    });
}

int pth_image_png_info(const char* path, uint32_t* width, uint32_t* height)
{
    return guarded([&] {
        const ImageRGBA8 img = loadPNG(path);
        *width = img.width, *height = img.height;
    });
}

int pth_image_load_png_rgba8(const char* path, uint8_t* rgba_out)
{
    return guarded([&] {
        const ImageRGBA8 img = loadPNG(path);
        std::memcpy(rgba_out, img.rgba.data(), img.rgba.size());
    });
}

int pth_image_load_material_png(const char* path, uint32_t width, uint32_t height, int isLinear, float* rgba_out)
{
    return guarded([&] {
        const ImageRGBAF img = loadMaterialLayer(path, width, height, isLinear != 0);
        std::memcpy(rgba_out, img.rgba.data(), img.rgba.size() * sizeof(float));
    });
}

int pth_mesh_refit(pth_mesh* m, const float* positions, const float* normals)
{
    return guarded([&] {
        if (!m || !positions)
            throw std::invalid_argument("pth_mesh_refit: null argument");
        ((MeshHandle*)m)->mesh->refit(positions, normals);
    });
}

int pth_image_load_material_png_bgra8(const char* path, uint32_t width, uint32_t height, int isLinear, uint8_t* bgra_out)
{
    return guarded([&] {
        if (!path || !bgra_out)
            throw std::invalid_argument("pth_image_load_material_png_bgra8: null argument");
        const ImageRGBA8 img = loadMaterialLayerBGRA8(path, width, height, isLinear != 0);
        std::memcpy(bgra_out, img.rgba.data(), img.rgba.size());
    });
}

void pth_mesh_destroy(pth_mesh* m) { delete (MeshHandle*)m; }

int pth_mesh_info(const pth_mesh* m, pth_mesh_stats* out)
{
    return guarded([&] {
        const Mesh& mesh = *((const MeshHandle*)m)->mesh;
        BvhStats s = checkBVH(mesh.getBvh(), mesh.getVertices().data(), mesh.numInputTriangles(), mesh.builder() != BvhBuilder::SpatialSplit);
        out->num_vertices = (uint32_t)mesh.getVertices().size();
        out->num_input_triangles = (uint32_t)mesh.numInputTriangles();
        out->num_triangle_refs = (uint32_t)mesh.getTriangles().size();
        out->num_nodes = (uint32_t)mesh.getBvhNodes().size();
        out->num_leaves = s.numLeaves;
        out->max_depth = s.maxDepth;
        out->max_leaf_size = s.maxLeafSize;
        out->children_inside_parents = s.childrenInsideParents;
        out->triangles_inside_leaves = s.trianglesInsideLeaves;
        out->all_triangles_referenced = s.allTrianglesReferenced;
        out->reachable_triangle_refs = s.numTriangleRefs;
        out->reachable_nodes = s.numNodes;
    });
}

int pth_mesh_copy_geometry(const pth_mesh* m, pt_vertex* vertices, pt_material* materials, uint32_t* numMaterials)
{
    return guarded([&] {
        const Mesh& mesh = *((const MeshHandle*)m)->mesh;
        if (vertices)
            std::memcpy(vertices, mesh.getVertices().data(), mesh.getVertices().size() * sizeof(pt_vertex));
        if (materials)
            for (size_t i = 0; i < mesh.getMaterials().size(); i++)
                materials[i] = mesh.getMaterials()[i];
        if (numMaterials)
            *numMaterials = (uint32_t)mesh.getMaterials().size();
    });
}

int pth_mesh_copy_bvh(const pth_mesh* m, pt_sub_bvh_node* nodes, pt_triangle* triangles, uint32_t* originalTriangle)
{
    return guarded([&] {
        const Mesh& mesh = *((const MeshHandle*)m)->mesh;
        if (nodes) std::memcpy(nodes, mesh.getBvhNodes().data(), mesh.getBvhNodes().size() * sizeof(pt_sub_bvh_node));
        if (triangles) std::memcpy(triangles, mesh.getTriangles().data(), mesh.getTriangles().size() * sizeof(pt_triangle));
        if (originalTriangle) std::memcpy(originalTriangle, mesh.getBvh().originalTriangle.data(), mesh.getBvh().originalTriangle.size() * sizeof(uint32_t));
    });
}

pth_scene* pth_scene_create(void) { return (pth_scene*)new SceneHandle(); }
void pth_scene_destroy(pth_scene* s) { delete (SceneHandle*)s; }

int pth_scene_add_node(pth_scene* s, const pth_mesh* m, const float location[3], const float orientation_wxyz[4], const float scale[3], int parent)
{
    int id = -1;
    int rc = guarded([&] {
        SceneHandle& h = *(SceneHandle*)s;
        Transform t;
        if (location) t.location = vec3(location[0], location[1], location[2]);
        if (orientation_wxyz) t.orientation = quat(orientation_wxyz[0], orientation_wxyz[1], orientation_wxyz[2], orientation_wxyz[3]);
        if (scale) t.scale = vec3(scale[0], scale[1], scale[2]);
        SceneNode* p = nullptr;
        if (parent >= 0) {
            if ((size_t)parent >= h.nodes.size())
                throw std::invalid_argument("pth_scene_add_node: bad parent id");
            p = h.nodes[parent];
        }
        SceneNode& n = h.scene.addNode(((const MeshHandle*)m)->mesh, t, p);
        id = (int)h.nodes.size();
        h.nodes.push_back(&n);
        h.flattened = h.staticFlattened = false;
    });
    return rc == 0 ? id : -1;
}

int pth_scene_set_transform(pth_scene* s, int node, const float location[3], const float orientation_wxyz[4], const float scale[3])
{
    return guarded([&] {
        SceneHandle& h = *(SceneHandle*)s;
        if (node < 0 || (size_t)node >= h.nodes.size())
            throw std::invalid_argument("pth_scene_set_transform: bad node id");
        Transform& t = h.nodes[node]->transform;
        if (location) t.location = vec3(location[0], location[1], location[2]);
        if (orientation_wxyz) t.orientation = quat(orientation_wxyz[0], orientation_wxyz[1], orientation_wxyz[2], orientation_wxyz[3]);
        if (scale) t.scale = vec3(scale[0], scale[1], scale[2]);
        h.flattened = false;
    });
}

int pth_scene_flatten(pth_scene* s, pth_scene_counts* counts)
{
    return guarded([&] {
        SceneHandle& h = *(SceneHandle*)s;
        h.flat = flattenScene(h.scene);
        h.flattened = h.staticFlattened = true;
        h.meshGenerations = h.generationsNow();
        counts->num_vertices = (uint32_t)h.flat.vertices.size();
        counts->num_triangles = (uint32_t)h.flat.triangles.size();
        counts->num_materials = (uint32_t)h.flat.materials.size();
        counts->num_sub_nodes = (uint32_t)h.flat.subBvhNodes.size();
        counts->num_lights = (uint32_t)h.flat.emissiveTriangles.size();
        counts->num_top_nodes = (uint32_t)h.flat.topBvhNodes.size();
        counts->top_root = h.flat.topBvhRoot;
        counts->num_instances = (uint32_t)h.scene.numInstances();
    });
}

int pth_scene_flatten_dynamic(pth_scene* s, pth_scene_counts* counts)
{
    return guarded([&] {
        SceneHandle& h = *(SceneHandle*)s;
        if (!h.staticFlattened)
            throw std::logic_error("pth_scene_flatten_dynamic: call pth_scene_flatten first (and again after adding nodes)");
        if (h.meshGenerations != h.generationsNow()) // (the lights and the top-level boxes below would follow the new mesh, the vertices and sub-BVH boxes not)
            throw std::logic_error("pth_scene_flatten_dynamic: a mesh of the scene was refitted since pth_scene_flatten: its vertices and sub-BVH boxes in the "
                                   "static arrays are stale -- call pth_scene_flatten again");
        flattenDynamic(h.scene, h.flat);
        h.flattened = true;
        counts->num_vertices = (uint32_t)h.flat.vertices.size();
        counts->num_triangles = (uint32_t)h.flat.triangles.size();
        counts->num_materials = (uint32_t)h.flat.materials.size();
        counts->num_sub_nodes = (uint32_t)h.flat.subBvhNodes.size();
        counts->num_lights = (uint32_t)h.flat.emissiveTriangles.size();
        counts->num_top_nodes = (uint32_t)h.flat.topBvhNodes.size();
        counts->top_root = h.flat.topBvhRoot;
        counts->num_instances = (uint32_t)h.scene.numInstances();
    });
}

int pth_scene_flatten_dynamic_only(pth_scene* s, pth_scene_counts* counts)
{
    return guarded([&] {
        SceneHandle& h = *(SceneHandle*)s;
        if (!h.staticFlattened)
            throw std::logic_error("pth_scene_flatten_dynamic_only: call pth_scene_flatten first (and again after adding nodes)");
        flattenDynamic(h.scene, h.flat);
        h.flattened = true; // (lights and top level are; the static arrays may be older than a refitted mesh: the caller said it does not want them)
        *counts = pth_scene_counts {};
        counts->num_lights = (uint32_t)h.flat.emissiveTriangles.size();
        counts->num_top_nodes = (uint32_t)h.flat.topBvhNodes.size();
        counts->top_root = h.flat.topBvhRoot;
        counts->num_instances = (uint32_t)h.scene.numInstances();
    });
}

int pth_scene_mesh_offsets(const pth_scene* s, const pth_mesh* m, uint32_t* firstVertex, uint32_t* firstNode)
{
    int found = -1;
    guarded([&] {
        const SceneHandle& h = *(const SceneHandle*)s;
        if (!h.staticFlattened)
            throw std::logic_error("pth_scene_mesh_offsets: call pth_scene_flatten first");
        for (const MeshBvhPair& p : h.scene.getMeshes())
            if (p.meshPtr.get() == ((const MeshHandle*)m)->mesh.get()) {
                if (firstVertex) *firstVertex = p.vertexIndexOffset;
                if (firstNode) *firstNode = p.bvhIndexOffset;
                found = 0;
            }
    });
    return found;
}

const pt_vertex* pth_mesh_vertices(const pth_mesh* m, uint32_t* count)
{
    if (!m)
        return nullptr;
    const auto& v = ((const MeshHandle*)m)->mesh->getVertices();
    if (count)
        *count = (uint32_t)v.size();
    return v.data();
}

int pth_scene_copy(const pth_scene* s, pt_vertex* v, pt_triangle* t, pt_material* m, pt_sub_bvh_node* n, pt_emissive_triangle* l, pt_top_bvh_node* top)
{
    return guarded([&] {
        const SceneHandle& h = *(const SceneHandle*)s;
        if (!h.flattened)
            throw std::logic_error("pth_scene_copy: call pth_scene_flatten first");
        auto cp = [](auto* dst, const auto& src) {
            if (dst && !src.empty())
                std::memcpy(dst, src.data(), src.size() * sizeof(src[0]));
        };
        cp(v, h.flat.vertices);
        cp(t, h.flat.triangles);
        cp(m, h.flat.materials);
        cp(n, h.flat.subBvhNodes);
        cp(l, h.flat.emissiveTriangles);
        cp(top, h.flat.topBvhNodes);
    });
}

int pth_camera_data(const pth_camera_params* p, pt_camera* out)
{
    return guarded([&] {
        Transform t(vec3(p->location[0], p->location[1], p->location[2]),
            quat(p->orientation_wxyz[0], p->orientation_wxyz[1], p->orientation_wxyz[2], p->orientation_wxyz[3]));
        Camera cam(t, p->horizontal_fov_deg, p->aspect_ratio, p->focal_distance);
        if (p->focal_length_mm > 0) cam.m_focalLengthMm = p->focal_length_mm;
        if (p->aperture_fstops > 0) cam.m_aperture = p->aperture_fstops;
        if (p->shutter_time > 0) cam.m_shutterTime = p->shutter_time;
        if (p->iso > 0) cam.m_iso = p->iso;
        cam.m_thinLens = p->thin_lens != 0;
        *out = cam.get_camera_data();
    });
}

} // extern "C"
