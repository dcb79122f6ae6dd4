// Host BVH builders, sequential (PTAMD_BUILD_THREADS=1) against the worker pool, same arrays required: g++ -std=c++17 -O2 -pthread -Iinclude -Iopencl-path-tracer_amd/host tools/micro/bvh_build_bench.cpp opencl-path-tracer_amd/host/bvh_build.cpp -o /tmp/bvh_build_bench
#include "bvh_build.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace raytracer;
int main(int argc, char** argv)
{
    const int nu = argc > 1 ? atoi(argv[1]) : 160, nv = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<pt_vertex> v(nu * nv);
    std::vector<pt_triangle> t;
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            const float u = i * 6.2831853f / nu, w = j * 6.2831853f / nv, r = 0.16f * (1.f + 0.25f * std::sin(5 * u) * std::cos(3 * w));
            pt_vertex& q = v[i * nv + j];
            std::memset(&q, 0, sizeof q);
            q.vertex[0] = (0.42f + r * std::cos(w)) * std::cos(u), q.vertex[1] = r * std::sin(w), q.vertex[2] = (0.42f + r * std::cos(w)) * std::sin(u);
        }
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            const uint32_t a = i * nv + j, b = ((i + 1) % nu) * nv + j, c = ((i + 1) % nu) * nv + (j + 1) % nv, d = i * nv + (j + 1) % nv;
            pt_triangle x;
            std::memset(&x, 0, sizeof x);
            x.indices[0] = a, x.indices[1] = c, x.indices[2] = b;
            t.push_back(x);
            x.indices[0] = a, x.indices[1] = d, x.indices[2] = c;
            t.push_back(x);
        }
    for (int kind = 0; kind < 3; kind++) {
        BvhBuildResult ref;
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0) setenv("PTAMD_BUILD_THREADS", "1", 1); else unsetenv("PTAMD_BUILD_THREADS");
            double best = 1e9;
            BvhBuildResult r;
            for (int k = 0; k < 5; k++) {
                auto t0 = std::chrono::steady_clock::now();
                r = buildBVH(v.data(), v.size(), t.data(), t.size(), (BvhBuilder)kind);
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            bool same = true;
            if (mode == 0) ref = r;
            else same = ref.nodes.size() == r.nodes.size() && !memcmp(ref.nodes.data(), r.nodes.data(), r.nodes.size() * sizeof(r.nodes[0]))
                && ref.originalTriangle == r.originalTriangle && !memcmp(ref.triangles.data(), r.triangles.data(), r.triangles.size() * sizeof(r.triangles[0]));
            printf("kind %d %s: %zu tris %.2f ms, %zu nodes%s\n", kind, mode ? "parallel  " : "sequential", t.size(), best, r.nodes.size(), mode ? (same ? " identical" : " DIFFERENT") : "");
        }
    }
}
