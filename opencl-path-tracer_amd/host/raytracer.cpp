#include "raytracer.h"
#include <cstring>

namespace raytracer {

void RayTracer::check(int rc, const char* what)
{
    if (rc != PT_OK)
        throw std::runtime_error(std::string(what) + ": " + pt_last_error(m_ctx));
}

RayTracer::RayTracer(int width, int height, std::shared_ptr<Scene> scene, const TextureArray& materialTextures, const TextureArray& skydomeTextures,
    int device, uint32_t seed)
    : m_scene(std::move(scene)), m_width(width), m_height(height)
{
    pt_config cfg {};
    cfg.width = (uint32_t)width;
    cfg.height = (uint32_t)height;
    cfg.device = device;
    cfg.seed = seed;
    cfg.samples_in_flight = 1; // rayTrace() adds exactly one sample per call, like the reference
    if (pt_create(&cfg, &m_ctx) != PT_OK)
        throw std::runtime_error(std::string("pt_create: ") + pt_last_error(nullptr));
    try {
        upload(materialTextures, skydomeTextures);
    } catch (...) { // the destructor does not run for a partially constructed object: release the context here
        pt_destroy(m_ctx);
        m_ctx = nullptr;
        throw;
    }
}

void RayTracer::upload(const TextureArray& materialTextures, const TextureArray& skydomeTextures)
{
    // static geometry once (reference: initBuffersAndTransferStaticData, src/raytracer.cpp:201-287) ...
    flattenStatic(*m_scene, m_flat);
    for (MeshBvhPair& pair : m_scene->getMeshes())
        pair.uploadedGeneration = pair.meshPtr->generation();
    check(pt_upload_static(m_ctx, m_flat.vertices.data(), (uint32_t)m_flat.vertices.size(), m_flat.triangles.data(), (uint32_t)m_flat.triangles.size(),
              m_flat.materials.data(), (uint32_t)m_flat.materials.size(), m_flat.subBvhNodes.data(), (uint32_t)m_flat.subBvhNodes.size()),
        "pt_upload_static");
    if (materialTextures.layers)
        check(pt_upload_texture_array(m_ctx, 0, materialTextures.width, materialTextures.height, materialTextures.layers,
                  materialTextures.storeAsFloat() ? PT_TEX_RGBA32F : PT_TEX_BGRA8_UNORM, materialTextures.data()),
            "pt_upload_texture_array(material)");
    if (skydomeTextures.layers)
        check(pt_upload_texture_array(m_ctx, 1, skydomeTextures.width, skydomeTextures.height, skydomeTextures.layers,
                  skydomeTextures.storeAsFloat() ? PT_TEX_RGBA32F : PT_TEX_BGRA8_UNORM, skydomeTextures.data()),
            "pt_upload_texture_array(skydome)");
    // ... then the dynamic part (lights + top-level BVH), as the reference's constructor ends with frameTick()
    frameTick();
}

RayTracer::~RayTracer() { pt_destroy(m_ctx); }

// transferDynamicData + the flip (reference src/raytracer.cpp:183-189,497-595): lights and top-level BVH from the scene graph as
// it stands, converted and copied into the inactive device buffers on the copy stream while frames enqueued so far keep
// rendering, then adopted by everything enqueued from here on.  No wait on the host.
void RayTracer::frameTick()
{
    flattenDynamic(*m_scene, m_flat);
    check(pt_upload_dynamic_async(m_ctx, m_flat.emissiveTriangles.data(), (uint32_t)m_flat.emissiveTriangles.size(), m_flat.topBvhNodes.data(),
              (uint32_t)m_flat.topBvhNodes.size(), m_flat.topBvhRoot),
        "pt_upload_dynamic_async");
    check(pt_frame_tick(m_ctx), "pt_frame_tick");
}

// Deformed meshes (Mesh::refit: same topology, new vertices -- the reference rewrites the dynamic tail of its vertex and sub-BVH buffers in
// transferDynamicData, :510-568, after refitting the boxes on the host, src/bvh/refit_bvh.cpp:6-34): every mesh that changed since the last
// call hands its vertices to the device library, which refits its own copy of the trees (pt_refit_vertices) -- no node array travels, the
// host never runs refitBVH in a frame loop (Mesh::refit leaves the boxes to whoever asks for them).  The following frameTick adopts it.
void RayTracer::updateGeometry()
{
    bool hostRoute = false;
    for (MeshBvhPair& pair : m_scene->getMeshes()) {
        const uint64_t gen = pair.meshPtr->generation();
        if (gen == IMesh::kUntracked) { // the mesh does not say when it changes: a dynamic one has changed, a static one never does
            if (pair.meshPtr->isDynamic()) {
                hostRoute = true;
                break;
            }
            continue;
        }
        if (gen == pair.uploadedGeneration)
            continue;
        const auto& verts = pair.meshPtr->getVertices();
        const int rc = pt_refit_vertices(m_ctx, pair.vertexIndexOffset, verts.data(), (uint32_t)verts.size());
        if (rc == PT_ERR_UNSUPPORTED) { // (roots of the sub-BVH array that share a subtree: the bottom-up pass on the device does not apply)
            hostRoute = true;
            break;
        }
        check(rc, "pt_refit_vertices");
        pair.uploadedGeneration = gen;
    }
    if (hostRoute) { // the reference's way: boxes refitted on the host, the whole vertex and node arrays handed over
        flattenStatic(*m_scene, m_flat);
        check(pt_update_geometry(m_ctx, m_flat.vertices.data(), (uint32_t)m_flat.vertices.size(), m_flat.subBvhNodes.data(), (uint32_t)m_flat.subBvhNodes.size()),
            "pt_update_geometry");
        for (MeshBvhPair& pair : m_scene->getMeshes())
            pair.uploadedGeneration = pair.meshPtr->generation();
    }
}

// A rebuilt scene (meshes replaced, topology changed -- the else branch of MeshSequence::buildBvh, reference src/model/mesh_sequence.cpp:89-96):
// everything static is flattened again and handed to the device library's second static set on the copy stream; frames keep rendering the old scene
// until the next frameTick, which uploads the new scene's lights and top level and adopts both.
void RayTracer::rebuildGeometry()
{
    flattenStatic(*m_scene, m_flat);
    for (MeshBvhPair& pair : m_scene->getMeshes())
        pair.uploadedGeneration = pair.meshPtr->generation();
    check(pt_upload_static_async(m_ctx, m_flat.vertices.data(), (uint32_t)m_flat.vertices.size(), m_flat.triangles.data(), (uint32_t)m_flat.triangles.size(),
              m_flat.materials.data(), (uint32_t)m_flat.materials.size(), m_flat.subBvhNodes.data(), (uint32_t)m_flat.subBvhNodes.size()),
        "pt_upload_static_async");
}

void RayTracer::rayTrace(const Camera& camera)
{
    const CameraData cam = camera.get_camera_data();
    if (!m_haveCamera || std::memcmp(&cam, &m_prevCamera, sizeof(CameraData)) != 0) { // src/raytracer.cpp:99-105
        m_prevCamera = cam;
        m_haveCamera = true;
        check(pt_set_camera(m_ctx, &cam), "pt_set_camera");
        check(pt_clear(m_ctx), "pt_clear");
    }
    if (getSamplesPerPixel() >= getMaxSamplesPerPixel())
        return;
    check(pt_render(m_ctx, 1), "pt_render");
    check(pt_synchronize(m_ctx), "pt_synchronize"); // queue.finish(), src/raytracer.cpp:117-118
}

int RayTracer::getSamplesPerPixel() const { return (int)pt_samples_per_pixel(m_ctx); }

std::vector<float> RayTracer::getOutput()
{
    std::vector<float> out((size_t)m_width * m_height * 4);
    check(pt_resolve(m_ctx, out.data()), "pt_resolve");
    return out;
}

std::vector<float> RayTracer::getAccumulator()
{
    std::vector<float> out((size_t)m_width * m_height * 4);
    check(pt_read_accum(m_ctx, out.data()), "pt_read_accum");
    return out;
}

pt_stats RayTracer::getStats()
{
    pt_stats s;
    check(pt_stats_get(m_ctx, &s), "pt_stats_get");
    return s;
}

} // namespace raytracer
