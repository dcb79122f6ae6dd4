// A REBUILT tree per frame through the kept C++ API -- the other branch of the reference's MeshSequence::buildBvh (src/model/mesh_sequence.cpp:89-96:
// `buildBinnedFastBVH` on the frame's triangles instead of a refit), whose arrays transferDynamicData re-uploads every tick (src/raytracer.cpp:510-568).
// A sequence-like IMesh makes a new Mesh (fast binned builder) per frame; RayTracer::rebuildGeometry hands the re-flattened scene to the device library's
// second static set (pt_upload_static_async: the host converts the topology, the device makes the records), RayTracer::frameTick adopts it with the new
// top level, frames keep rendering meanwhile.  Prints the median milliseconds of every stage and checks the last frame against a RayTracer that only ever
// saw that frame: the two accumulators must be identical.
//      usage: rebuild_loop [level] [frames]        (icosphere level: 5 = 20 480 triangles, bench.py's `rebuild_20k`)
#include "../opencl-path-tracer_amd/host/raytracer.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

using namespace raytracer;
using Clock = std::chrono::steady_clock;

static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

// what the reference's MeshSequence is to its RayTracer: an IMesh whose arrays are another frame's after goToFrame
class RebuiltMesh : public IMesh {
public:
    RebuiltMesh(std::vector<uint32_t> indices, Material material)
        : m_indices(std::move(indices)), m_material(material) {}
    void goToFrame(const std::vector<float>& positions)
    {
        m_mesh = std::make_shared<Mesh>(positions.data(), nullptr, nullptr, positions.size() / 3, m_indices.data(), nullptr, m_indices.size() / 3,
            std::vector<Material> { m_material }, BvhBuilder::BinnedFast);
    }
    const std::vector<VertexSceneData>& getVertices() const override { return m_mesh->getVertices(); }
    const std::vector<TriangleSceneData>& getTriangles() const override { return m_mesh->getTriangles(); }
    const std::vector<Material>& getMaterials() const override { return m_mesh->getMaterials(); }
    const std::vector<SubBVHNode>& getBvhNodes() const override { return m_mesh->getBvhNodes(); }
    const std::vector<uint32_t>& getEmissiveTriangles() const override { return m_mesh->getEmissiveTriangles(); }
    AABB getBounds() const override { return m_mesh->getBounds(); }
    bool isDynamic() const override { return true; }
    uint32_t maxNumVertices() const override { return m_mesh->maxNumVertices(); }
    uint32_t maxNumTriangles() const override { return m_mesh->maxNumTriangles(); }
    uint32_t maxNumMaterials() const override { return m_mesh->maxNumMaterials(); }
    uint32_t maxNumBvhNodes() const override { return m_mesh->maxNumBvhNodes(); }
    void buildBvh() override {}
    uint32_t getBvhRootNode() const override { return m_mesh->getBvhRootNode(); }

private:
    std::vector<uint32_t> m_indices;
    Material m_material;
    std::shared_ptr<Mesh> m_mesh;
};

static std::shared_ptr<Mesh> quadMesh(vec3 a, vec3 b, vec3 c, vec3 d, const Material& m)
{
    const float pos[12] = { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z };
    const uint32_t idx[6] = { 0, 1, 2, 0, 2, 3 };
    return std::make_shared<Mesh>(pos, nullptr, nullptr, 4, idx, nullptr, 2, std::vector<Material> { m }, BvhBuilder::BinnedSAH);
}

// unit icosphere, `level` subdivisions (20 * 4^level triangles)
static void icosphere(int level, std::vector<float>& pos, std::vector<uint32_t>& idx)
{
    const float t = (1.0f + std::sqrt(5.0f)) / 2.0f;
    const float v[12][3] = { { -1, t, 0 }, { 1, t, 0 }, { -1, -t, 0 }, { 1, -t, 0 }, { 0, -1, t }, { 0, 1, t }, { 0, -1, -t }, { 0, 1, -t }, { t, 0, -1 }, { t, 0, 1 }, { -t, 0, -1 }, { -t, 0, 1 } };
    const uint32_t f[20][3] = { { 0, 11, 5 }, { 0, 5, 1 }, { 0, 1, 7 }, { 0, 7, 10 }, { 0, 10, 11 }, { 1, 5, 9 }, { 5, 11, 4 }, { 11, 10, 2 }, { 10, 7, 6 }, { 7, 1, 8 },
        { 3, 9, 4 }, { 3, 4, 2 }, { 3, 2, 6 }, { 3, 6, 8 }, { 3, 8, 9 }, { 4, 9, 5 }, { 2, 4, 11 }, { 6, 2, 10 }, { 8, 6, 7 }, { 9, 8, 1 } };
    auto push = [&](float x, float y, float z) {
        const float n = std::sqrt(x * x + y * y + z * z);
        pos.push_back(x / n), pos.push_back(y / n), pos.push_back(z / n);
        return (uint32_t)(pos.size() / 3 - 1);
    };
    pos.clear(), idx.clear();
    for (const auto& p : v)
        push(p[0], p[1], p[2]);
    for (const auto& tri : f)
        idx.insert(idx.end(), tri, tri + 3);
    for (int l = 0; l < level; l++) {
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> mid;
        auto middle = [&](uint32_t a, uint32_t b) {
            const auto key = std::make_pair(std::min(a, b), std::max(a, b));
            auto it = mid.find(key);
            if (it != mid.end())
                return it->second;
            const uint32_t m = push(pos[3 * a] + pos[3 * b], pos[3 * a + 1] + pos[3 * b + 1], pos[3 * a + 2] + pos[3 * b + 2]);
            mid.emplace(key, m);
            return m;
        };
        std::vector<uint32_t> next;
        for (size_t i = 0; i < idx.size(); i += 3) {
            const uint32_t a = idx[i], b = idx[i + 1], c = idx[i + 2], ab = middle(a, b), bc = middle(b, c), ca = middle(c, a);
            const uint32_t t4[12] = { a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca };
            next.insert(next.end(), t4, t4 + 12);
        }
        idx.swap(next);
    }
}

static void deformed(const std::vector<float>& unit, int k, std::vector<float>& pos)
{
    pos.resize(unit.size());
    for (size_t i = 0; i < unit.size(); i += 3) {
        const float s = 0.5f * (1.0f + 0.1f * std::sin((float)k + 1.0f + 5.0f * unit[i]));
        pos[i] = unit[i] * s, pos[i + 1] = unit[i + 1] * s, pos[i + 2] = unit[i + 2] * s;
    }
}

int main(int argc, char** argv)
{
    const int level = argc > 1 ? std::atoi(argv[1]) : 5, frames = argc > 2 ? std::atoi(argv[2]) : 12;
    const int W = 1280, H = 720;
    try {
        std::vector<float> unit, pos;
        std::vector<uint32_t> idx;
        icosphere(level, unit, idx);
        const Material white = Material::Diffuse(vec3(0.73f)), blobMaterial = Material::PBRDielectric(vec3(0.75f, 0.2f, 0.15f), 0.7f);
        auto makeScene = [&](std::shared_ptr<RebuiltMesh>* out, int frame) {
            auto scene = std::make_shared<Scene>();
            scene->addNode(quadMesh({ -1, 0, -1 }, { -1, 0, 1 }, { 1, 0, 1 }, { 1, 0, -1 }, white));
            scene->addNode(quadMesh({ -1, 0, 1 }, { -1, 2, 1 }, { 1, 2, 1 }, { 1, 0, 1 }, white));
            scene->addNode(quadMesh({ -0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, 0.25f }, { -0.25f, 1.98f, 0.25f },
                Material::Emissive(vec3(1.0f, 0.92f, 0.8f), 12.0f)));
            auto seq = std::make_shared<RebuiltMesh>(idx, blobMaterial);
            deformed(unit, frame, pos);
            seq->goToFrame(pos);
            Transform t;
            t.location = vec3(0.0f, 0.8f, 0.1f);
            t.scale = vec3(1.2f);
            scene->addNode(seq, t);
            *out = seq;
            return scene;
        };
        TextureArray noTextures, sky;
        const float grey[4] = { 0.4f, 0.4f, 0.4f, 1.0f };
        sky.add(grey, 1, 1);
        Transform camT(vec3(0.0f, 1.0f, -3.9f));
        Camera camera(camT, 50.0f, (float)W / H, 3.9f);
        camera.m_thinLens = false;
        auto check = [](int rc, pt_ctx* c) {
            if (rc != PT_OK)
                throw std::runtime_error(pt_last_error(c));
        };

        std::shared_ptr<RebuiltMesh> seq;
        RayTracer rt(W, H, makeScene(&seq, 0), noTextures, sky);
        rt.rayTrace(camera);
        std::vector<double> tBuild, tRebuild, tTick, tAdopted;
        for (int k = 1; k <= frames + 2; k++) {
            check(pt_render(rt.context(), 1), rt.context()); // a frame of the old scene is in flight while the host builds
            deformed(unit, k, pos);
            const auto t0 = Clock::now();
            seq->goToFrame(pos);
            const auto t1 = Clock::now();
            check(pt_render(rt.context(), 1), rt.context()); // ... and another one while the host converts
            rt.rebuildGeometry();
            const auto t2 = Clock::now();
            rt.frameTick();
            const auto t3 = Clock::now();
            check(pt_render(rt.context(), 1), rt.context()); // the first frame of the new scene
            check(pt_synchronize(rt.context()), rt.context());
            const auto t4 = Clock::now();
            if (k > 2) // the first two rebuilds size the two static sets
                tBuild.push_back(ms(t0, t1)), tRebuild.push_back(ms(t1, t2)), tTick.push_back(ms(t2, t3)), tAdopted.push_back(ms(t0, t4));
        }
        // the last frame against a RayTracer that only ever saw it
        check(pt_clear(rt.context()), rt.context());
        for (int s = 0; s < 4; s++)
            rt.rayTrace(camera);
        const std::vector<float> got = rt.getAccumulator();
        std::shared_ptr<RebuiltMesh> seq2;
        RayTracer fresh(W, H, makeScene(&seq2, frames + 2), noTextures, sky);
        for (int s = 0; s < 4; s++)
            fresh.rayTrace(camera);
        const std::vector<float> want = fresh.getAccumulator();
        const bool same = got.size() == want.size() && std::memcmp(got.data(), want.data(), got.size() * sizeof(float)) == 0;
        std::printf("triangles=%zu frames=%d mesh_build_ms=%.3f rebuild_geometry_ms=%.3f frame_tick_ms=%.3f until_first_new_frame_ms=%.3f same_as_fresh=%d\n", idx.size() / 3, frames,
            median(tBuild), median(tRebuild), median(tTick), median(tAdopted), same ? 1 : 0);
        return same ? 0 : 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
