// ThreadSanitizer harness for the host library's worker pool (threaded smooth normals, refit, lazy box refit inside flatten), several ticks.  Build container:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -Iinclude tools/tsan_host.cpp opencl-path-tracer_amd/host/{bvh_build,mesh,scene,image,capi}.cpp -o /tmp/tsan_host -lz && /tmp/tsan_host
#include "ptamd.h"
#include "ptamd_host.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
int main()
{
    const int nu = 160, nv = 120;
    std::vector<float> p(3 * nu * nv);
    std::vector<uint32_t> idx, mat;
    auto fill = [&](float t) {
        for (int i = 0; i < nu; i++)
            for (int j = 0; j < nv; j++) {
                const float u = i * 6.2831853f / nu, v = j * 6.2831853f / nv, r = 0.16f * (1.f + 0.25f * std::sin(5 * u + t) * std::cos(3 * v));
                float* q = &p[3 * (i * nv + j)];
                q[0] = (0.42f + r * std::cos(v)) * std::cos(u), q[1] = r * std::sin(v), q[2] = (0.42f + r * std::cos(v)) * std::sin(u);
            }
    };
    fill(0.f);
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            const uint32_t a = i * nv + j, b = ((i + 1) % nu) * nv + j, c = ((i + 1) % nu) * nv + (j + 1) % nv, d = i * nv + (j + 1) % nv;
            const uint32_t t[6] = { a, c, b, a, d, c };
            idx.insert(idx.end(), t, t + 6);
            mat.push_back(0), mat.push_back(0);
        }
    pt_material m;
    std::memset(&m, 0, sizeof m);
    pth_mesh* mesh = pth_mesh_create(p.data(), nullptr, nullptr, (size_t)nu * nv, idx.data(), mat.data(), idx.size() / 3, &m, 1, 0);
    if (!mesh) { std::printf("create failed: %s\n", pth_last_error()); return 1; }
    pth_scene* s = pth_scene_create();
    const float loc[3] = { 0, 0, 0 }, q[4] = { 1, 0, 0, 0 }, sc[3] = { 1, 1, 1 };
    pth_scene_add_node(s, mesh, loc, q, sc, -1);
    pth_scene_counts cnt;
    for (int k = 0; k < 6; k++) {
        fill(0.3f * k);
        if (pth_mesh_refit(mesh, p.data(), nullptr) != 0) { std::printf("refit failed: %s\n", pth_last_error()); return 1; }
        if (pth_scene_flatten(s, &cnt) != 0) { std::printf("flatten failed: %s\n", pth_last_error()); return 1; }
    }
    std::printf("ok: %d triangles, 6 ticks\n", (int)(idx.size() / 3));
    pth_scene_destroy(s);
    pth_mesh_destroy(mesh);
    return 0;
}
