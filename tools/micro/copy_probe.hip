// GPU box: which float4 device copy reaches what MI355X_MICROARCH.md quotes (6.29 TB/s)?  Variants: accesses per thread in flight, temporal hint, grid, block.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/copy_probe.hip -o tools/micro/copy_probe && tools/micro/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(1024) k_copy(const f4v* __restrict__ src, f4v* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT)
                __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else
                dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride)
        dst[i] = src[i];
}
template <int U, bool NT>
static void run(const f4v* s, f4v* d, size_t n, int blocks, int block, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(block), 0, 0, s, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best)
            best = ms;
    }
    printf("%-28s blocks %6d x %4d : %7.1f GB/s\n", name, blocks, block, 2.0 * n * 16 / (best * 1e-3) / 1e9);
}
int main()
{
    const size_t bytes = 2ull << 30, n = bytes / 16;
    f4v *s, *d;
    hipMalloc(&s, bytes), hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int per : { 4, 8, 16, 32 })
        for (int block : { 256, 512, 1024 }) {
            run<1, false>(s, d, n, cus * per, block, "1 access, plain");
            run<1, true>(s, d, n, cus * per, block, "1 access, nontemporal");
            run<4, false>(s, d, n, cus * per, block, "4 accesses, plain");
            run<4, true>(s, d, n, cus * per, block, "4 accesses, nontemporal");
            run<8, true>(s, d, n, cus * per, block, "8 accesses, nontemporal");
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("hipMemcpyAsync D2D: %.1f GB/s\n", 2.0 * bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
