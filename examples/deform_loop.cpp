// A deforming mesh per frame through the kept C++ API, the way the reference's MeshSequence drives it (src/model/mesh_sequence.cpp:81-97,
// src/bvh/refit_bvh.cpp:6-34, transferDynamicData src/raytracer.cpp:510-568): every tick Mesh::refit (new vertex positions, boxes refitted
// bottom-up, topology kept), RayTracer::updateGeometry (pt_update_geometry), RayTracer::frameTick (lights + top level, pt_upload_dynamic_async +
// pt_frame_tick), then one sample per pixel.  Prints the median milliseconds of every stage: the tick without any binding in between.
//      usage: deform_loop [nu nv] [frames]        (2 * nu * nv triangles on a lumpy torus; 135 x 135 = 36 450, the lab report's mesh size)
#include "../opencl-path-tracer_amd/host/raytracer.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace raytracer;
using Clock = std::chrono::steady_clock;

static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

static std::shared_ptr<Mesh> quadMesh(vec3 a, vec3 b, vec3 c, vec3 d, const Material& m)
{
    const float pos[12] = { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z };
    const uint32_t idx[6] = { 0, 1, 2, 0, 2, 3 };
    return std::make_shared<Mesh>(pos, nullptr, nullptr, 4, idx, nullptr, 2, std::vector<Material> { m }, BvhBuilder::BinnedSAH);
}

static void torus(int nu, int nv, float phase, std::vector<float>& pos)
{
    pos.resize((size_t)nu * nv * 3);
    const float twoPi = 6.28318530718f;
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            const float u = twoPi * i / nu, v = twoPi * j / nv;
            const float r = 0.16f * (1.0f + 0.25f * std::sin(5 * u + phase) * std::cos(3 * v));
            float* p = &pos[((size_t)i * nv + j) * 3];
            p[0] = (0.42f + r * std::cos(v)) * std::cos(u), p[1] = r * std::sin(v) * (1.0f + 0.1f * std::sin(phase)), p[2] = (0.42f + r * std::cos(v)) * std::sin(u);
        }
}

int main(int argc, char** argv)
{
    const int nu = argc > 2 ? std::atoi(argv[1]) : 135, nv = argc > 2 ? std::atoi(argv[2]) : 135;
    const int frames = argc > 3 ? std::atoi(argv[3]) : (argc == 2 ? std::atoi(argv[1]) : 12);
    const int W = 640, H = 360;
    try {
        auto scene = std::make_shared<Scene>();
        const Material white = Material::Diffuse(vec3(0.73f));
        scene->addNode(quadMesh({ -1, 0, -1 }, { -1, 0, 1 }, { 1, 0, 1 }, { 1, 0, -1 }, white));
        scene->addNode(quadMesh({ -1, 0, 1 }, { -1, 2, 1 }, { 1, 2, 1 }, { 1, 0, 1 }, white));
        scene->addNode(quadMesh({ -0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, 0.25f }, { -0.25f, 1.98f, 0.25f },
            Material::Emissive(vec3(1.0f, 0.92f, 0.8f), 12.0f)));
        std::vector<float> pos;
        torus(nu, nv, 0.0f, pos);
        std::vector<uint32_t> idx;
        for (int i = 0; i < nu; i++)
            for (int j = 0; j < nv; j++) {
                const uint32_t a = (uint32_t)(i * nv + j), b = (uint32_t)(((i + 1) % nu) * nv + j), c = (uint32_t)(((i + 1) % nu) * nv + (j + 1) % nv),
                               d = (uint32_t)(i * nv + (j + 1) % nv);
                const uint32_t t[6] = { a, c, b, a, d, c };
                idx.insert(idx.end(), t, t + 6);
            }
        auto mesh = std::make_shared<Mesh>(pos.data(), nullptr, nullptr, (size_t)nu * nv, idx.data(), nullptr, idx.size() / 3,
            std::vector<Material> { Material::PBRMetal(vec3(0.955f, 0.638f, 0.538f), 0.8f) }, BvhBuilder::SpatialSplit);
        Transform t;
        t.location = vec3(0.0f, 0.8f, 0.1f);
        scene->addNode(mesh, t);

        TextureArray noTextures, sky;
        const float grey[4] = { 0.4f, 0.4f, 0.4f, 1.0f };
        sky.add(grey, 1, 1);
        RayTracer rt(W, H, scene, noTextures, sky);
        Transform camT(vec3(0.0f, 1.0f, -3.9f));
        Camera camera(camT, 50.0f, (float)W / H, 3.9f);
        camera.m_thinLens = false;
        rt.rayTrace(camera);

        std::vector<double> tRefit, tGeom, tTick, tAdopted, tRender;
        double firstMean = 0, lastMean = 0;
        for (int k = 0; k < frames + 2; k++) {
            torus(nu, nv, 0.35f * (k + 1), pos);
            const auto t0 = Clock::now();
            mesh->refit(pos.data(), nullptr);
            const auto t1 = Clock::now();
            rt.updateGeometry();
            const auto t2 = Clock::now();
            rt.frameTick();
            const auto t3 = Clock::now();
            if (pt_synchronize(rt.context()) != PT_OK)
                throw std::runtime_error(pt_last_error(rt.context()));
            const auto t4 = Clock::now();
            rt.rayTrace(camera); // (the geometry changed, not the camera: samples accumulate over the deforming mesh like in the reference's viewer)
            const auto t5 = Clock::now();
            if (k >= 2) // the first two ticks size the two buffer sets
                tRefit.push_back(ms(t0, t1)), tGeom.push_back(ms(t1, t2)), tTick.push_back(ms(t2, t3)), tAdopted.push_back(ms(t0, t4)), tRender.push_back(ms(t4, t5));
            if (k == 0 || k == frames + 1) {
                const std::vector<float> acc = rt.getAccumulator();
                double m = 0;
                for (size_t i = 0; i < acc.size(); i += 4)
                    m += acc[i] + acc[i + 1] + acc[i + 2];
                (k == 0 ? firstMean : lastMean) = m / (3.0 * W * H) / rt.getSamplesPerPixel();
            }
        }
        std::printf("triangles=%zu frames=%d refit_ms=%.3f update_geometry_ms=%.3f frame_tick_ms=%.3f until_adopted_ms=%.3f render_1spp_ms=%.3f mean_first=%.5f mean_last=%.5f spp=%d\n",
            idx.size() / 3, frames, median(tRefit), median(tGeom), median(tTick), median(tAdopted), median(tRender), firstMean, lastMean, rt.getSamplesPerPixel());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
