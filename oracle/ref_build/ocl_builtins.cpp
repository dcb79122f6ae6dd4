// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
//
// The reference's device code (/root/reference/assets/cl/*.cl) is OpenCL C.  ROCm clang
// compiles it unmodified for x86-64 (`-x cl -D__local=`), leaving the OpenCL C *language
// built-ins* it calls (work-item functions, math, geometric, image reads) undefined, because
// those are normally supplied by whichever OpenCL implementation JITs the kernel.
// This file defines exactly those built-ins, by their OpenCL 1.2 specification semantics, so
// that the reference kernels can be executed one work-item at a time on the host CPU
// (SURVEY.md section 8(c)).  It contains no code from the reference.
//
// Build with the SAME clang as the kernels: ext_vector_type(N) then has the ABI of OpenCL floatN.
#include <cmath>
#include <cstdint>
#include <cstring>

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float3_t __attribute__((ext_vector_type(3)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef int int2_t __attribute__((ext_vector_type(2)));

// Image handle used for image2d_array_t (read) and image2d_t (write): float RGBA texels.
struct RefImage {
    int32_t width, height, layers, _pad;
    float* rgba; // [layer][y][x][4]
};

// ---- work-item state (set by ref_driver.cpp before each work-item) -------------------
extern "C" {
thread_local size_t ref_global_id[3] = { 0, 0, 0 };
thread_local size_t ref_local_id[3] = { 0, 0, 0 };
}

#define OCL(name) __asm__(name)

size_t ocl_get_global_id(unsigned d) OCL("_Z13get_global_idj");
size_t ocl_get_global_id(unsigned d) { return d < 3 ? ref_global_id[d] : 0; }
size_t ocl_get_local_id(unsigned d) OCL("_Z12get_local_idj");
size_t ocl_get_local_id(unsigned d) { return d < 3 ? ref_local_id[d] : 0; }
void ocl_barrier(unsigned) OCL("_Z7barrierj");
void ocl_barrier(unsigned) {} // work-group size is 1 in the serial driver
unsigned ocl_atomic_inc(volatile unsigned* p) OCL("_Z10atomic_incPU8CLglobalVj");
unsigned ocl_atomic_inc(volatile unsigned* p)
{
    unsigned old = *p;
    *p = old + 1;
    return old;
}

// ---- scalar math: correctly-rounded / libm single precision -----------------------------
float ocl_cos(float x) OCL("_Z3cosf");
float ocl_cos(float x) { return cosf(x); }
float ocl_sin(float x) OCL("_Z3sinf");
float ocl_sin(float x) { return sinf(x); }
float ocl_tan(float x) OCL("_Z3tanf");
float ocl_tan(float x) { return tanf(x); }
float ocl_acos(float x) OCL("_Z4acosf");
float ocl_acos(float x) { return acosf(x); }
float ocl_atan(float x) OCL("_Z4atanf");
float ocl_atan(float x) { return atanf(x); }
float ocl_atan2(float y, float x) OCL("_Z5atan2ff");
float ocl_atan2(float y, float x) { return atan2f(y, x); }
float ocl_exp(float x) OCL("_Z3expf");
float ocl_exp(float x) { return expf(x); }
float ocl_pow(float x, float y) OCL("_Z3powff");
float ocl_pow(float x, float y) { return powf(x, y); }
float ocl_sqrt(float x) OCL("_Z4sqrtf");
float ocl_sqrt(float x) { return sqrtf(x); }
float ocl_fabs(float x) OCL("_Z4fabsf");
float ocl_fabs(float x) { return fabsf(x); }
float ocl_log1p(float x) OCL("_Z5log1pf");
float ocl_log1p(float x) { return log1pf(x); }
float ocl_log2(float x) OCL("_Z4log2f");
float ocl_log2(float x) { return log2f(x); }
float ocl_fmin(float a, float b) OCL("_Z4fminff");
float ocl_fmin(float a, float b) { return fminf(a, b); }
float ocl_fmax(float a, float b) OCL("_Z4fmaxff");
float ocl_fmax(float a, float b) { return fmaxf(a, b); }
// OpenCL 1.2 s6.12.4: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x
float ocl_min(float a, float b) OCL("_Z3minff");
float ocl_min(float a, float b) { return b < a ? b : a; }
float ocl_max(float a, float b) OCL("_Z3maxff");
float ocl_max(float a, float b) { return a < b ? b : a; }
// mix(x,y,a) = x + (y-x)*a ; clamp(x,lo,hi) = min(max(x,lo),hi)
float ocl_mix(float x, float y, float a) OCL("_Z3mixfff");
float ocl_mix(float x, float y, float a) { return x + (y - x) * a; }
float ocl_clamp(float x, float lo, float hi) OCL("_Z5clampfff");
float ocl_clamp(float x, float lo, float hi) { return ocl_min(ocl_max(x, lo), hi); }

// ---- float3 built-ins -------------------------------------------------------------------
float ocl_dot3(float3_t a, float3_t b) OCL("_Z3dotDv3_fS_");
float ocl_dot3(float3_t a, float3_t b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float3_t ocl_cross3(float3_t a, float3_t b) OCL("_Z5crossDv3_fS_");
float3_t ocl_cross3(float3_t a, float3_t b)
{
    float3_t r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
float3_t ocl_normalize3(float3_t a) OCL("_Z9normalizeDv3_f");
float3_t ocl_normalize3(float3_t a)
{
    float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    float3_t r;
    r.x = a.x / len;
    r.y = a.y / len;
    r.z = a.z / len;
    return r;
}
float3_t ocl_exp3(float3_t a) OCL("_Z3expDv3_f");
float3_t ocl_exp3(float3_t a)
{
    float3_t r;
    r.x = expf(a.x);
    r.y = expf(a.y);
    r.z = expf(a.z);
    return r;
}
float3_t ocl_pow3(float3_t a, float3_t b) OCL("_Z3powDv3_fS_");
float3_t ocl_pow3(float3_t a, float3_t b)
{
    float3_t r;
    r.x = powf(a.x, b.x);
    r.y = powf(a.y, b.y);
    r.z = powf(a.z, b.z);
    return r;
}
float3_t ocl_fabs3(float3_t a) OCL("_Z4fabsDv3_f");
float3_t ocl_fabs3(float3_t a)
{
    float3_t r;
    r.x = fabsf(a.x);
    r.y = fabsf(a.y);
    r.z = fabsf(a.z);
    return r;
}

// ---- images ---------------------------------------------------------------------------
// Every sampler in the reference is NORMALIZED_COORDS_TRUE | ADDRESS_REPEAT | FILTER_LINEAR
// (shading_helper.cl:19-22, skydome.cl:4-7), so the sampler handle carries no information.
void* ocl_translate_sampler(int v) OCL("__translate_sampler_initializer");
void* ocl_translate_sampler(int v) { return (void*)(intptr_t)(v | 0x10000); }

// OpenCL 1.2 s8.2 linear filter with s8.3 repeat addressing, 2D array (layer = clamp(rint(w))).
float4_t ocl_read_imagef_2darray(const RefImage* img, void* sampler, float4_t c)
    OCL("_Z11read_imagef20ocl_image2d_array_ro11ocl_samplerDv4_f");
float4_t ocl_read_imagef_2darray(const RefImage* img, void*, float4_t c)
{
    const int w = img->width, h = img->height;
    float u = (c.x - floorf(c.x)) * (float)w;
    float v = (c.y - floorf(c.y)) * (float)h;
    int i0 = (int)floorf(u - 0.5f), j0 = (int)floorf(v - 0.5f);
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += w;
    if (i1 > w - 1) i1 -= w;
    if (j0 < 0) j0 += h;
    if (j1 > h - 1) j1 -= h;
    float a = (u - 0.5f) - floorf(u - 0.5f);
    float b = (v - 0.5f) - floorf(v - 0.5f);
    int layer = (int)rintf(c.z);
    if (layer < 0) layer = 0;
    if (layer > img->layers - 1) layer = img->layers - 1;
    const float* base = img->rgba + (size_t)layer * w * h * 4;
    const float* t00 = base + ((size_t)j0 * w + i0) * 4;
    const float* t10 = base + ((size_t)j0 * w + i1) * 4;
    const float* t01 = base + ((size_t)j1 * w + i0) * 4;
    const float* t11 = base + ((size_t)j1 * w + i1) * 4;
    float4_t r;
    for (int k = 0; k < 4; k++) {
        r[k] = (1 - a) * (1 - b) * t00[k] + a * (1 - b) * t10[k] + (1 - a) * b * t01[k] + a * b * t11[k];
    }
    return r;
}

void ocl_write_imagef_2d(RefImage* img, int2_t xy, float4_t c) OCL("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void ocl_write_imagef_2d(RefImage* img, int2_t xy, float4_t c)
{
    if (xy.x < 0 || xy.y < 0 || xy.x >= img->width || xy.y >= img->height)
        return;
    float* t = img->rgba + ((size_t)xy.y * img->width + xy.x) * 4;
    t[0] = c.x;
    t[1] = c.y;
    t[2] = c.z;
    t[3] = c.w;
}

// ---- plain-C doors to the built-ins above, for tests/test_ocl_builtins.py ------------------------------------------
// (the kernels reach them through their OpenCL-mangled names with vector arguments in SSE registers, which ctypes
// cannot pass).  The test checks them against values derived from the OpenCL 1.2 specification WITHOUT restating the
// formulas above: what is left unpinned by running the reference's kernels on this shim is the shim itself.
extern "C" {
void ref_test_read_imagef(const RefImage* img, const float* coord4, float* out4)
{
    float4_t c = { coord4[0], coord4[1], coord4[2], coord4[3] };
    float4_t r = ocl_read_imagef_2darray(img, ocl_translate_sampler(0), c);
    for (int k = 0; k < 4; k++)
        out4[k] = r[k];
}
void ref_test_vec3(int op, const float* a3, const float* b3, float* out3)
{
    float3_t a = { a3[0], a3[1], a3[2] }, b = { b3[0], b3[1], b3[2] }, r = { 0, 0, 0 };
    switch (op) {
    case 0: r.x = ocl_dot3(a, b); break;
    case 1: r = ocl_cross3(a, b); break;
    case 2: r = ocl_normalize3(a); break;
    case 3: r = ocl_exp3(a); break;
    case 4: r = ocl_pow3(a, b); break;
    case 5: r = ocl_fabs3(a); break;
    }
    out3[0] = r.x, out3[1] = r.y, out3[2] = r.z;
}
float ref_test_scalar(int op, float x, float y, float z)
{
    switch (op) {
    case 0: return ocl_min(x, y);
    case 1: return ocl_max(x, y);
    case 2: return ocl_mix(x, y, z);
    case 3: return ocl_clamp(x, y, z);
    case 4: return ocl_fmin(x, y);
    case 5: return ocl_fmax(x, y);
    case 6: return ocl_cos(x);
    case 7: return ocl_sin(x);
    case 8: return ocl_tan(x);
    case 9: return ocl_acos(x);
    case 10: return ocl_atan(x);
    case 11: return ocl_atan2(x, y);
    case 12: return ocl_exp(x);
    case 13: return ocl_pow(x, y);
    case 14: return ocl_sqrt(x);
    case 15: return ocl_fabs(x);
    case 16: return ocl_log1p(x);
    case 17: return ocl_log2(x);
    }
    return 0.f;
}
unsigned ref_test_atomic_inc(unsigned* p) { return ocl_atomic_inc(p); }
}
