// An IMesh that is NOT raytracer::Mesh and does not count its generations -- what the reference's MeshSequence is to its RayTracer
// (src/model/mesh_sequence.h: isDynamic() true, arrays that change under goToNextFrame, src/model/mesh_sequence.cpp:81-97).
// RayTracer::updateGeometry must take such a mesh for changed on every call and hand the re-flattened arrays over (the reference's route,
// transferDynamicData src/raytracer.cpp:510-568); ADVICE r5: it was skipped silently and the next frame showed stale geometry.
// Renders the deformed state through updateGeometry + frameTick and through a fresh RayTracer: the two accumulators must be identical.
//      usage: untracked_mesh          prints stale_differs=<0|1> same_as_fresh=<0|1>
#include "../opencl-path-tracer_amd/host/raytracer.h"
#include <cmath>
#include <cstdio>
#include <cstring>

using namespace raytracer;

// a sequence-like mesh: forwards everything to a Mesh it owns, deforms through it, reports no generation
class SequenceLikeMesh : public IMesh {
public:
    explicit SequenceLikeMesh(std::shared_ptr<Mesh> m) : m_mesh(std::move(m)) {}
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
    // (no generation(): IMesh's default, kUntracked)
    void goToFrame(const std::vector<float>& positions) { m_mesh->refit(positions.data(), nullptr); }

private:
    std::shared_ptr<Mesh> m_mesh;
};

static std::shared_ptr<Mesh> quadMesh(vec3 a, vec3 b, vec3 c, vec3 d, const Material& m)
{
    const float pos[12] = { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z };
    const uint32_t idx[6] = { 0, 1, 2, 0, 2, 3 };
    return std::make_shared<Mesh>(pos, nullptr, nullptr, 4, idx, nullptr, 2, std::vector<Material> { m }, BvhBuilder::BinnedSAH);
}

static void blob(int n, float phase, std::vector<float>& pos, std::vector<uint32_t>* idx)
{
    pos.resize((size_t)n * n * 3);
    const float twoPi = 6.28318530718f;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const float u = twoPi * i / n, v = twoPi * j / n, r = 0.2f * (1.0f + 0.3f * std::sin(3 * u + phase));
            float* p = &pos[((size_t)i * n + j) * 3];
            p[0] = (0.45f + r * std::cos(v)) * std::cos(u), p[1] = r * std::sin(v), p[2] = (0.45f + r * std::cos(v)) * std::sin(u);
        }
    if (!idx)
        return;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const uint32_t a = (uint32_t)(i * n + j), b = (uint32_t)(((i + 1) % n) * n + j), c = (uint32_t)(((i + 1) % n) * n + (j + 1) % n), d = (uint32_t)(i * n + (j + 1) % n);
            const uint32_t t[6] = { a, c, b, a, d, c };
            idx->insert(idx->end(), t, t + 6);
        }
}

int main()
{
    const int W = 160, H = 90, N = 24, SPP = 4;
    try {
        std::vector<float> pos0, pos1;
        std::vector<uint32_t> idx;
        blob(N, 0.0f, pos0, &idx);
        blob(N, 1.3f, pos1, nullptr);
        auto makeScene = [&](const std::vector<float>& pos, std::shared_ptr<SequenceLikeMesh>* out) {
            auto scene = std::make_shared<Scene>();
            scene->addNode(quadMesh({ -1, 0, -1 }, { -1, 0, 1 }, { 1, 0, 1 }, { 1, 0, -1 }, Material::Diffuse(vec3(0.73f))));
            scene->addNode(quadMesh({ -0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, 0.25f }, { -0.25f, 1.98f, 0.25f },
                Material::Emissive(vec3(1.0f, 0.92f, 0.8f), 12.0f)));
            auto inner = std::make_shared<Mesh>(pos.data(), nullptr, nullptr, (size_t)N * N, idx.data(), nullptr, idx.size() / 3,
                std::vector<Material> { Material::Diffuse(vec3(0.8f, 0.3f, 0.2f)) }, BvhBuilder::BinnedSAH);
            auto seq = std::make_shared<SequenceLikeMesh>(inner);
            Transform t;
            t.location = vec3(0.0f, 0.8f, 0.1f);
            scene->addNode(seq, t);
            if (out)
                *out = seq;
            return scene;
        };
        TextureArray noTextures, sky;
        const float grey[4] = { 0.4f, 0.4f, 0.4f, 1.0f };
        sky.add(grey, 1, 1);
        Transform camT(vec3(0.0f, 1.0f, -3.9f));
        Camera camera(camT, 50.0f, (float)W / H, 3.9f);
        camera.m_thinLens = false;
        auto render = [&](RayTracer& rt) {
            if (pt_clear(rt.context()) != PT_OK)
                throw std::runtime_error(pt_last_error(rt.context()));
            for (int s = 0; s < SPP; s++)
                rt.rayTrace(camera);
            return rt.getAccumulator();
        };

        std::shared_ptr<SequenceLikeMesh> seq;
        RayTracer rt(W, H, makeScene(pos0, &seq), noTextures, sky);
        const std::vector<float> first = render(rt);
        seq->goToFrame(pos1);
        rt.updateGeometry();
        rt.frameTick();
        const std::vector<float> updated = render(rt);

        // the yardstick: the same mesh (same tree topology: built on the first frame's positions) deformed BEFORE a RayTracer ever sees it
        std::shared_ptr<SequenceLikeMesh> seq2;
        auto scene2 = makeScene(pos0, &seq2);
        seq2->goToFrame(pos1);
        RayTracer fresh(W, H, scene2, noTextures, sky);
        const std::vector<float> want = render(fresh);
        const bool staleDiffers = std::memcmp(first.data(), updated.data(), first.size() * sizeof(float)) != 0;
        const bool same = std::memcmp(want.data(), updated.data(), want.size() * sizeof(float)) == 0;
        std::printf("stale_differs=%d same_as_fresh=%d\n", staleDiffers ? 1 : 0, same ? 1 : 0);
        return staleDiffers && same ? 0 : 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
