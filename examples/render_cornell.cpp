// The reference's main loop (src/main.cpp:34-121) without the window: build a Scene through the kept
// Scene / Mesh / Material / Camera / RayTracer API, call rayTrace() N times, write the tone-mapped
// image as a PPM.     usage: render_cornell [spp] [out.ppm]
#include "../opencl-path-tracer_amd/host/raytracer.h"
#include <cstdio>
#include <cstdlib>

using namespace raytracer;

static std::shared_ptr<Mesh> quadMesh(vec3 a, vec3 b, vec3 c, vec3 d, const Material& m)
{
    const float pos[12] = { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z };
    const uint32_t idx[6] = { 0, 1, 2, 0, 2, 3 };
    return std::make_shared<Mesh>(pos, nullptr, nullptr, 4, idx, nullptr, 2, std::vector<Material> { m }, BvhBuilder::BinnedSAH);
}

int main(int argc, char** argv)
{
    const int spp = argc > 1 ? std::atoi(argv[1]) : 64;
    const char* out = argc > 2 ? argv[2] : "cornell.ppm";
    const int W = 256, H = 256;
    try {
        auto scene = std::make_shared<Scene>();
        const Material white = Material::Diffuse(vec3(0.73f)), green = Material::Diffuse(vec3(0.12f, 0.45f, 0.15f)),
                       red = Material::Diffuse(vec3(0.65f, 0.05f, 0.05f));
        scene->addNode(quadMesh({ -1, 0, -1 }, { -1, 0, 1 }, { 1, 0, 1 }, { 1, 0, -1 }, white)); // floor
        scene->addNode(quadMesh({ -1, 2, -1 }, { 1, 2, -1 }, { 1, 2, 1 }, { -1, 2, 1 }, white)); // ceiling
        scene->addNode(quadMesh({ -1, 0, 1 }, { -1, 2, 1 }, { 1, 2, 1 }, { 1, 0, 1 }, white)); // back
        scene->addNode(quadMesh({ -1, 0, -1 }, { -1, 2, -1 }, { -1, 2, 1 }, { -1, 0, 1 }, green)); // left
        scene->addNode(quadMesh({ 1, 0, -1 }, { 1, 0, 1 }, { 1, 2, 1 }, { 1, 2, -1 }, red)); // right
        scene->addNode(quadMesh({ -0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, -0.25f }, { 0.25f, 1.98f, 0.25f }, { -0.25f, 1.98f, 0.25f },
            Material::Emissive(vec3(1.0f, 0.92f, 0.8f), 12.0f)));
        // a copper sphere-ish blob would come from Mesh::fromPLY(".../bun_zipper.ply", Material::PBRMetal(...)) (src/main.cpp:143-155)
        auto box = quadMesh({ -0.3f, 0.6f, -0.3f }, { -0.3f, 0.6f, 0.3f }, { 0.3f, 0.6f, 0.3f }, { 0.3f, 0.6f, -0.3f },
            Material::PBRMetal(vec3(0.955f, 0.638f, 0.538f), 0.8f));
        Transform t;
        t.location = vec3(0.2f, 0.0f, 0.1f);
        scene->addNode(box, t);

        TextureArray noTextures, sky;
        const float grey[4] = { 0.4f, 0.4f, 0.4f, 1.0f };
        sky.add(grey, 1, 1);
        RayTracer rt(W, H, scene, noTextures, sky);

        Transform camT(vec3(0.0f, 1.0f, -3.9f)); // identity orientation looks down +z
        Camera camera(camT, 40.0f, (float)W / H, 3.9f);
        camera.m_thinLens = false;
        for (int i = 0; i < spp; i++)
            rt.rayTrace(camera);
        std::vector<float> img = rt.getOutput();
        pt_stats st = rt.getStats();
        FILE* f = std::fopen(out, "wb");
        if (!f)
            throw std::runtime_error("cannot write output");
        std::fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (int i = 0; i < W * H; i++)
            for (int k = 0; k < 3; k++)
                std::fputc((int)(std::fmin(1.0f, std::fmax(0.0f, img[i * 4 + k])) * 255.0f + 0.5f), f);
        std::fclose(f);
        double mean = 0;
        for (int i = 0; i < W * H; i++)
            mean += img[i * 4] + img[i * 4 + 1] + img[i * 4 + 2];
        std::printf("spp=%d rays=%llu mean=%.5f -> %s\n", rt.getSamplesPerPixel(), (unsigned long long)(st.rays_extension + st.rays_shadow),
            mean / (3.0 * W * H), out);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
