// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
// Serial 2-D NDRange driver for the reference's `accumulate` kernel
// (/root/reference/assets/cl/accumulate.cl:6-34), launch shape of raytracer.cpp:432-450.
#include <cstddef>
#include <cstdint>

extern "C" {
extern thread_local size_t ref_global_id[3];
extern thread_local size_t ref_local_id[3];
void accumulate(void* outputImage, void* input, const void* kernelData, uint32_t n, uint32_t scrWidth);

void ref_accumulate(uint32_t width, uint32_t height, void* outputImage, void* input, const void* kd, uint32_t n)
{
    for (uint32_t y = 0; y < height; y++) {
        for (uint32_t x = 0; x < width; x++) {
            ref_global_id[0] = x;
            ref_global_id[1] = y;
            ref_global_id[2] = 0;
            ref_local_id[0] = ref_local_id[1] = ref_local_id[2] = 0;
            accumulate(outputImage, input, kd, n, width);
        }
    }
}
}
