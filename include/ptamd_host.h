/* ptamd_host.h -- C entry points of the host-side scene library (CPU only, no HIP):
 * meshes + bottom-level BVH builders, scene graph with instancing, top-level BVH, flattening into
 * the device arrays of ptamd.h, and camera parameter derivation.  These mirror, for non-C++
 * callers, the C++ classes in opencl-path-tracer_amd/host/ which keep the reference's API
 * (Scene::addNode src/scene.h:39, Mesh src/model/mesh.h, buildTopBVH src/bvh/top_bvh_build.h:12,
 * Camera::get_camera_data src/camera.h:33).  Calls return 0 / a handle on success, -1 / NULL on
 * failure with the message in pth_last_error() (thread-local). */
#ifndef PTAMD_HOST_H
#define PTAMD_HOST_H
#include "ptamd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_mesh pth_mesh;
typedef struct pth_scene pth_scene;

enum { PTH_BVH_BINNED_SAH = 0, PTH_BVH_BINNED_FAST = 1, PTH_BVH_SPATIAL_SPLIT = 2 };

typedef struct {
    uint32_t num_vertices, num_input_triangles, num_triangle_refs, num_nodes, num_leaves, max_depth, max_leaf_size;
    uint32_t children_inside_parents, triangles_inside_leaves, all_triangles_referenced; /* BvhTester invariants, src/bvh/bvh_test.cpp:117-139 */
    uint32_t reachable_triangle_refs, reachable_nodes;
} pth_mesh_stats;

typedef struct {
    uint32_t num_vertices, num_triangles, num_materials, num_sub_nodes, num_lights, num_top_nodes, top_root, num_instances;
} pth_scene_counts;

typedef struct {
    float location[3];
    float orientation_wxyz[4];
    float horizontal_fov_deg, aspect_ratio, focal_distance;
    float focal_length_mm, aperture_fstops, shutter_time, iso; /* <= 0: reference defaults 50, 8, 1/32, 1200 (src/camera.cpp:5-15) */
    int thin_lens;
} pth_camera_params;

const char* pth_last_error(void);

pth_mesh* pth_mesh_create(const float* positions, const float* normals, const float* texCoords, size_t numVertices,
    const uint32_t* indices, const uint32_t* materialIndex, size_t numTriangles, const pt_material* materials,
    size_t numMaterials, int builder);
/* Same, with the reference's .bvh cache file (src/model/mesh.cpp:175-263: u32 version 1, u32 root, u32 numNodes,
 * nodes[48 B], u32 numTriangles, triangles[16 B], '\n'): loaded if present and valid for this mesh, else the
 * tree is built and the file written.  pth_mesh_bvh_from_cache tells which happened. */
pth_mesh* pth_mesh_create_cached(const float* positions, const float* normals, const float* texCoords, size_t numVertices,
    const uint32_t* indices, const uint32_t* materialIndex, size_t numTriangles, const pt_material* materials,
    size_t numMaterials, int builder, const char* bvhCacheFile);
int pth_mesh_store_bvh(const pth_mesh* m, const char* path); /* Mesh::storeBvh, mesh.cpp:202-225 */
int pth_mesh_bvh_from_cache(const pth_mesh* m);
pth_mesh* pth_mesh_from_ply(const char* path, const pt_material* material, int builder);
/* Wavefront OBJ + MTL, imported the way the reference imports model files (src/model/mesh.cpp:36-200): polygons
 * triangulated, corners welded, the TRS `offset` baked in (null = identity), Ke != 0 -> Emissive(Ke) else Diffuse(Kd),
 * or one override material for everything (null = use the MTL). */
pth_mesh* pth_mesh_from_obj(const char* path, const pt_material* overrideMaterial, const float location[3], const float orientation_wxyz[4],
    const float scale[3], int builder, const char* bvhCacheFile);
/* UniqueTextureArray (src/opencl/texture.h:18-31): the texture files a scene's materials use, each once; the index is the
 * material's tex_id = the layer of the material texture array.  pth_mesh_from_obj_textured registers the map_Kd file of
 * every non-emissive MTL material there and makes it Material::Diffuse(tex_id, Kd) (src/model/mesh.cpp:61-66); load the
 * layers with pth_image_load_material_png at the array's size (1024 x 1024 in the reference, src/raytracer.cpp:284). */
typedef struct pth_texture_files pth_texture_files;
pth_texture_files* pth_texture_files_create(void);
void pth_texture_files_destroy(pth_texture_files* t);
int pth_texture_files_add(pth_texture_files* t, const char* path, int isLinear, float brightnessMultiplier); /* the id; same file, same id */
int pth_texture_files_count(const pth_texture_files* t);
const char* pth_texture_files_path(const pth_texture_files* t, int index, int* isLinear, float* brightnessMultiplier);
pth_mesh* pth_mesh_from_obj_textured(const char* path, const pt_material* overrideMaterial, const float location[3], const float orientation_wxyz[4],
    const float scale[3], int builder, const char* bvhCacheFile, pth_texture_files* textures);
void pth_mesh_destroy(pth_mesh* m);
int pth_mesh_info(const pth_mesh* m, pth_mesh_stats* out);
int pth_mesh_copy_bvh(const pth_mesh* m, pt_sub_bvh_node* nodes, pt_triangle* triangles, uint32_t* originalTriangle);
/* A deformed frame of the same mesh (MeshSequence + refitBVH, reference src/model/mesh_sequence.cpp:81-97, src/bvh/refit_bvh.cpp):
 * 3 * num_vertices new positions, normals likewise or NULL (regenerated smooth); topology and leaf order stay, boxes are refitted. */
int pth_mesh_refit(pth_mesh* m, const float* positions, const float* normals);
int pth_mesh_copy_geometry(const pth_mesh* m, pt_vertex* vertices, pt_material* materials, uint32_t* numMaterials); /* vertices: num_vertices entries */

pth_scene* pth_scene_create(void);
void pth_scene_destroy(pth_scene* s);
/* returns the new node id (>= 0); parent = -1 for the root */
int pth_scene_add_node(pth_scene* s, const pth_mesh* m, const float location[3], const float orientation_wxyz[4], const float scale[3], int parent);
int pth_scene_set_transform(pth_scene* s, int node, const float location[3], const float orientation_wxyz[4], const float scale[3]);
int pth_scene_flatten(pth_scene* s, pth_scene_counts* counts);
/* The per-tick half alone -- lights and top-level BVH from the scene graph as it stands (flattenDynamic: what RayTracer::frameTick does
 * on the host before pt_upload_dynamic_async; reference transferDynamicData, src/raytracer.cpp:497-509,569-595); the static arrays of the
 * last pth_scene_flatten stay as they are.  Copy out with pth_scene_copy (NULL for the arrays that are not wanted).
 * Fails (-1) when a node was added or a mesh of the scene was refitted (pth_mesh_refit) since that pth_scene_flatten: the static arrays
 * are then stale, and lights / top-level boxes made from the new mesh would not belong to them -- call pth_scene_flatten again. */
int pth_scene_flatten_dynamic(pth_scene* s, pth_scene_counts* counts);
/* The same without the check that the static arrays are current: for callers that hand deformed meshes to the device library themselves
 * (pt_refit_vertices) and want nothing but the lights and the top level of the scene as it stands. */
int pth_scene_flatten_dynamic_only(pth_scene* s, pth_scene_counts* counts);
/* where mesh `m` lies in the arrays of the last pth_scene_flatten: its first vertex / sub-BVH node; -1 when the mesh is not part of the scene */
int pth_scene_mesh_offsets(const pth_scene* s, const pth_mesh* m, uint32_t* firstVertex, uint32_t* firstNode);
/* the mesh's vertex array in place (num_vertices records of 48 bytes; valid until the mesh is destroyed, contents change with pth_mesh_refit) */
const pt_vertex* pth_mesh_vertices(const pth_mesh* m, uint32_t* count);
int pth_scene_copy(const pth_scene* s, pt_vertex* v, pt_triangle* t, pt_material* m, pt_sub_bvh_node* n, pt_emissive_triangle* l, pt_top_bvh_node* top);

int pth_camera_data(const pth_camera_params* p, pt_camera* out);

/* Radiance .hdr -> one RGBA32F layer of a texture array, as CLTextureArray::loadImage prepares it
 * (src/opencl/texture.cpp:72-120): rescaled to width x height (Lanczos-3), alpha 1, colours x brightnessMultiplier,
 * rows bottom-up (FreeImage order).  rgba_out: width*height*4 floats = what pt_upload_texture_array takes. */
int pth_image_hdr_info(const char* path, uint32_t* width, uint32_t* height);
int pth_image_load_hdr(const char* path, uint32_t width, uint32_t height, float brightnessMultiplier, float* rgba_out);
/* PNG -> one layer of the 8-bit material texture array, as CLTextureArray::loadImage prepares it
 * (src/opencl/texture.cpp:72-92,112-131: rescale to the layer size, FreeImage_AdjustGamma(1/2.2) unless isLinear, 32 bits
 * per texel, rows bottom-up), returned as the RGBA floats read_imagef yields (byte / 255) for pt_upload_texture_array
 * kind 0.  width / height 0 = keep the file's size.  pth_image_load_png_rgba8: the decoded file itself, top-down. */
int pth_image_png_info(const char* path, uint32_t* width, uint32_t* height);
int pth_image_load_png_rgba8(const char* path, uint8_t* rgba_out);
int pth_image_load_material_png(const char* path, uint32_t width, uint32_t height, int isLinear, float* rgba_out);
/* the same layer as bytes b g r a, rows bottom-up: the bitmap the reference uploads (PT_TEX_BGRA8_UNORM) */
int pth_image_load_material_png_bgra8(const char* path, uint32_t width, uint32_t height, int isLinear, uint8_t* bgra_out);

#ifdef __cplusplus
}
#endif
#endif
