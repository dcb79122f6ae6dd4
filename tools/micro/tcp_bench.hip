// Micro-benchmark (diagnostic, not part of the product): what does a divergent 64-byte node fetch cost
// on the CU's vector-memory path, per wave-step, for different ways of issuing it?
//   A  every lane: 4 x global_load_dwordx4 of its own 64-B node (what k_trace does)
//   B  quad-cooperative: in round r the 4 lanes of a quad fetch the 4 chunks of the node wanted by the
//      quad's r-th lane (one 64-B request per quad per round); 4 rounds
//   C  every lane: 1 x dwordx4 (16-B "node"), D: 2 x dwordx4 (32 B)
// Dependent chain: the next node index is derived from the loaded data (like a traversal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256, 4) k(const uint4* __restrict__ nodes, uint32_t mask, int iters, uint32_t* out, int alu)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v;
        if (MODE == 0) {
            const uint4* p = nodes + (size_t)idx * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            v = a.x ^ b.y ^ c.z ^ d.w;
        } else if (MODE == 1) {
            uint32_t part[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t want = __shfl(idx, (lane & ~3u) | r); // ds_bpermute; a real kernel would use DPP quad_perm
                const uint4 a = nodes[(size_t)want * 4 + (lane & 3u)];
                part[r] = a.x ^ a.y ^ a.z ^ a.w;
            }
            // every lane needs "its" node: combine the 4 partial values of round (lane&3) across the quad
            uint32_t mine = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t s = part[r];
                s ^= __shfl_xor(s, 1);
                s ^= __shfl_xor(s, 2);
                mine = (lane & 3u) == (uint32_t)r ? s : mine;
            }
            v = mine;
        } else if (MODE == 2) {
            const uint4 a = nodes[(size_t)idx * 4];
            v = a.x ^ a.w;
        } else {
            const uint4* p = nodes + (size_t)idx * 4;
            const uint4 a = p[0], b = p[1];
            v = a.x ^ b.y;
        }
        for (int q = 0; q < alu; q++) // stand-in for the slab tests
            v = v * 1664525u + 1013904223u;
        acc += v;
        idx = mix(v + it) & mask;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
    const int log2Nodes = argc > 1 ? atoi(argv[1]) : 18; // 2^18 x 64 B = 16 MiB
    const int iters = 2000;
    const size_t n = (size_t)1 << log2Nodes;
    std::vector<uint4> host(n * 4);
    for (size_t i = 0; i < host.size(); i++) host[i] = make_uint4(rand(), rand(), rand(), rand());
    uint4* dn; uint32_t* dout;
    CHECK(hipMalloc(&dn, host.size() * 16));
    CHECK(hipMemcpy(dn, host.data(), host.size() * 16, hipMemcpyHostToDevice));
    const int blocks = 256 * 4;
    CHECK(hipMalloc(&dout, blocks * 256 * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = { "A 4x16B per lane", "B quad-cooperative", "C 1x16B per lane", "D 2x16B per lane" };
    for (int alu = 0; alu <= 100; alu += 100)
        for (int mode = 0; mode < 4; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, dn, (uint32_t)(n - 1), iters, dout, alu);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, dn, (uint32_t)(n - 1), iters, dout, alu);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, dn, (uint32_t)(n - 1), iters, dout, alu);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, dn, (uint32_t)(n - 1), iters, dout, alu);
                hipEventRecord(e1);
                CHECK(hipEventSynchronize(e1));
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double steps = (double)blocks * 4 * iters; // wave-steps
            printf("nodes 2^%d alu %3d  %-20s %8.3f ms  %7.1f ns per wave-step per CU  (%.2f G lane-steps/s)\n", log2Nodes, alu, names[mode], ms,
                ms * 1e6 / (steps / 256.0), steps * 64 / ms / 1e6);
        }
    return 0;
}
