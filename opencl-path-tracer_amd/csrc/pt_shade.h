// gen_camera_rays, shade (NEE + importance sampling + Russian roulette) and resolve kernels.
//
// Semantics follow the reference's generatePrimaryRays (assets/cl/kernel.cl:24-84, camera.cl:28-77),
// shade (kernel.cl:190-301) with neeIsShading (shading.cl:356-623) and accumulate
// (accumulate.cl:6-34); citations on each block.  CDNA4 specifics: float4-SoA queue records,
// wave-aggregated compaction (one atomicAdd per wave per output queue, ranks from __ballot +
// popcount) with separate live-ray and shadow-ray counters, textures sampled from plain buffers
// (CDNA has no texture units), zero-state counter PRNG.
#pragma once
#include "pt_math.h"

namespace ptd {

constexpr float kPI = 3.14159265359f; // shapes.cl:5
constexpr float kINVPI = 0.31830988618f; // shading_helper.cl:9
constexpr float kEPS = 0.0001f; // shading_helper.cl:15
constexpr float kMaxSmoothness = 0.94f; // shading.cl:9
constexpr float kAirIor = 1.000277f; // scene.cl:46
enum : int { MAT_DIFFUSE = 0, MAT_PBR = 1, MAT_REFRACTIVE = 2, MAT_BASIC_REFRACTIVE = 3, MAT_EMISSIVE = 4 };

// ---- texture fetch: NORMALIZED_COORDS | ADDRESS_REPEAT | FILTER_LINEAR on a 2D array ----------
// (sampler of shading_helper.cl:19-22 / skydome.cl:4-7; OpenCL 1.2 s8.2 linear filter, s8.3 repeat)
__device__ inline float4 sampleLinearRepeat(const Texture& tex, float s, float t, float layerCoord)
{
    const int w = tex.width, h = tex.height;
    const float u = (s - floorf(s)) * (float)w;
    const float v = (t - floorf(t)) * (float)h;
    int i0 = (int)floorf(u - 0.5f), j0 = (int)floorf(v - 0.5f);
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += w;
    if (i1 > w - 1) i1 -= w;
    if (j0 < 0) j0 += h;
    if (j1 > h - 1) j1 -= h;
    const float a = (u - 0.5f) - floorf(u - 0.5f);
    const float b = (v - 0.5f) - floorf(v - 0.5f);
    int layer = (int)rintf(layerCoord);
    layer = max(0, min(layer, tex.layers - 1));
    const size_t base = (size_t)layer * w * h;
    const float4 t00 = fetchTexel(tex, base + (size_t)j0 * w + i0), t10 = fetchTexel(tex, base + (size_t)j0 * w + i1);
    const float4 t01 = fetchTexel(tex, base + (size_t)j1 * w + i0), t11 = fetchTexel(tex, base + (size_t)j1 * w + i1);
    const float w00 = (1 - a) * (1 - b), w10 = a * (1 - b), w01 = (1 - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x, w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
        w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z, w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}

__device__ inline V3 readSkydome(const SceneDev& sc, V3 dir) // skydome.cl:12-26
{
    if (!sc.sky.texels)
        return mk(0.0f);
    float u = 1 + atan2f(dir.x, -dir.z) / kPI;
    const float v = acosf(dir.y) / kPI;
    u /= 2;
    return xyz(sampleLinearRepeat(sc.sky, u, 1.0f - v, 0.0f));
}

// ---- camera (camera.cl:28-77) -----------------------------------------------------------------
// Written as explicit scalar operations under `fp contract(off)`: three kernels generate primary rays (k_gen, and -- where the
// primary rays are not queued -- the packet kernel), and what one of them traces must be bit for bit what another would
// have: with the compiler free to contract a * b + c differently in each inlining context it was not (4 of 423 k rays of a
// test render took another path).  FMAs are spelled out where wanted, and the two divisions and the square root are the
// hardware's 1-ulp v_rcp_f32 / v_rsq_f32 by name (left to the compiler, `1 / x` came out as a bare v_rcp_f32 in k_gen and as a
// range-scaled sequence in the packet kernel).
__device__ inline float fmaE(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline void pinholeRay(const CameraDev& cam, int x, int y, float width, float height, Rng& rng, V3* o, V3* d)
{
#pragma clang fp contract(off)
    const float rw = __builtin_amdgcn_rcpf(width), rh = __builtin_amdgcn_rcpf(height);
    const float ux = cam.u.x * rw, uy = cam.u.y * rw, uz = cam.u.z * rw; // uStep = u / width
    const float vx = cam.v.x * rh, vy = cam.v.y * rh, vz = cam.v.z * rh;
    const float fx = (float)x, fy = (float)y;
    float sx = fmaE(vx, fy, fmaE(ux, fx, cam.screen.x)), sy = fmaE(vy, fy, fmaE(uy, fx, cam.screen.y)), sz = fmaE(vz, fy, fmaE(uz, fx, cam.screen.z));
    const float r1 = rng.u01();
    sx = fmaE(r1, ux, sx), sy = fmaE(r1, uy, sy), sz = fmaE(r1, uz, sz);
    const float r2 = rng.u01();
    sx = fmaE(r2, vx, sx), sy = fmaE(r2, vy, sy), sz = fmaE(r2, vz, sz);
    const float dx = sx - cam.eye.x, dy = sy - cam.eye.y, dz = sz - cam.eye.z;
    const float inv = __builtin_amdgcn_rsqf(fmaE(dz, dz, fmaE(dy, dy, dx * dx)));
    *o = xyz(cam.eye);
    *d = mk(dx * inv, dy * inv, dz * inv);
}
// Parity mode: the reference's operations one by one, each correctly rounded and none contracted (what oracle/_ref and the
// oracle execute: IEEE division and square root, (a + b) + c as written): the render follows the reference's kernels path by path
// only until the first decision flips, and an ulp in a primary ray is enough to flip one within a few thousand paths.
__device__ inline void pinholeRayPrecise(const CameraDev& cam, int x, int y, float width, float height, Rng& rng, V3* o, V3* d)
{
    const float ux = __fdiv_rn(cam.u.x, width), uy = __fdiv_rn(cam.u.y, width), uz = __fdiv_rn(cam.u.z, width);
    const float vx = __fdiv_rn(cam.v.x, height), vy = __fdiv_rn(cam.v.y, height), vz = __fdiv_rn(cam.v.z, height);
    const float fx = (float)x, fy = (float)y;
    float sx = __fadd_rn(__fadd_rn(cam.screen.x, __fmul_rn(ux, fx)), __fmul_rn(vx, fy));
    float sy = __fadd_rn(__fadd_rn(cam.screen.y, __fmul_rn(uy, fx)), __fmul_rn(vy, fy));
    float sz = __fadd_rn(__fadd_rn(cam.screen.z, __fmul_rn(uz, fx)), __fmul_rn(vz, fy));
    const float r1 = rng.u01();
    sx = __fadd_rn(sx, __fmul_rn(r1, ux)), sy = __fadd_rn(sy, __fmul_rn(r1, uy)), sz = __fadd_rn(sz, __fmul_rn(r1, uz));
    const float r2 = rng.u01();
    sx = __fadd_rn(sx, __fmul_rn(r2, vx)), sy = __fadd_rn(sy, __fmul_rn(r2, vy)), sz = __fadd_rn(sz, __fmul_rn(r2, vz));
    const float dx = __fsub_rn(sx, cam.eye.x), dy = __fsub_rn(sy, cam.eye.y), dz = __fsub_rn(sz, cam.eye.z);
    const float len = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    *o = xyz(cam.eye);
    *d = mk(__fdiv_rn(dx, len), __fdiv_rn(dy, len), __fdiv_rn(dz, len));
}
__device__ inline void cameraRayPrecise(const CameraDev& cam, int x, int y, float width, float height, Rng& rng, V3* o, V3* d)
{
    if (!cam.thinLens) {
        pinholeRayPrecise(cam, x, y, width, height, rng, o, d);
        return;
    }
    const float r1 = __fsub_rn(__fmul_rn(rng.u01(), 2.0f), 1.0f); // square aperture (camera.cl:62-66)
    const float r2 = __fsub_rn(__fmul_rn(rng.u01(), 2.0f), 1.0f);
    const float a = cam.apertureRadius;
    const float ox = __fadd_rn(__fmul_rn(__fmul_rn(r1, cam.uN.x), a), __fmul_rn(__fmul_rn(r2, cam.vN.x), a));
    const float oy = __fadd_rn(__fmul_rn(__fmul_rn(r1, cam.uN.y), a), __fmul_rn(__fmul_rn(r2, cam.vN.y), a));
    const float oz = __fadd_rn(__fmul_rn(__fmul_rn(r1, cam.uN.z), a), __fmul_rn(__fmul_rn(r2, cam.vN.z), a));
    V3 po, pd;
    pinholeRayPrecise(cam, x, y, width, height, rng, &po, &pd);
    const float f = cam.focalDistance;
    const V3 focal = mk(__fadd_rn(po.x, __fmul_rn(f, pd.x)), __fadd_rn(po.y, __fmul_rn(f, pd.y)), __fadd_rn(po.z, __fmul_rn(f, pd.z)));
    const V3 lens = mk(__fadd_rn(po.x, ox), __fadd_rn(po.y, oy), __fadd_rn(po.z, oz));
    *o = lens;
    *d = mk(__fsub_rn(focal.x, lens.x), __fsub_rn(focal.y, lens.y), __fsub_rn(focal.z, lens.z)); // not normalised (camera.cl:71-75)
}

__device__ inline void cameraRay(const CameraDev& cam, int x, int y, float width, float height, Rng& rng, V3* o, V3* d)
{
#pragma clang fp contract(off)
    if (!cam.thinLens) {
        pinholeRay(cam, x, y, width, height, rng, o, d);
        return;
    }
    const float r1 = (rng.u01() * 2.0f - 1.0f) * cam.apertureRadius; // square aperture (camera.cl:62-66)
    const float r2 = (rng.u01() * 2.0f - 1.0f) * cam.apertureRadius;
    const float ox = fmaE(r2, cam.vN.x, r1 * cam.uN.x), oy = fmaE(r2, cam.vN.y, r1 * cam.uN.y), oz = fmaE(r2, cam.vN.z, r1 * cam.uN.z);
    V3 po, pd;
    pinholeRay(cam, x, y, width, height, rng, &po, &pd);
    // focal = po + focalDistance * pd, lens = po + offset, direction = focal - lens: not normalised, as the reference
    // (camera.cl:71-75)
    const V3 lens = mk(po.x + ox, po.y + oy, po.z + oz);
    const V3 focal = mk(fmaE(cam.focalDistance, pd.x, po.x), fmaE(cam.focalDistance, pd.y, po.y), fmaE(cam.focalDistance, pd.z, po.z));
    *o = lens;
    *d = mk(focal.x - lens.x, focal.y - lens.y, focal.z - lens.z);
}

// ---- BSDF pieces (pbr_brdf.cl, refract.cl) -------------------------------------------------------
// pow(x, 5) of pbr_brdf.cl as three multiplications (powf is ~50 instructions of exp2/log2 with special cases)
__device__ inline float pow5(float x)
{
    const float x2 = x * x;
    return x2 * x2 * x;
}
__device__ inline V3 F_Schlick(V3 f0, float f90, float u) { return f0 + (mk(f90) - f0) * pow5(1.0f - u); }
__device__ inline float G_SmithBeckmannCorrelated(float VdotM, float NdotV, float alpha)
{
    // 1 / (alpha * tan(acos(NdotV))) with tan(acos c) = sqrt(1 - c^2) / c: same value and the same signs / infinities
    const float a = NdotV / (alpha * sqrtf(1.0f - NdotV * NdotV));
    const float chi = a > 0 ? 1.0f : 0.0f;
    float approx = 1.0f;
    if (a < 1.6f)
        approx = (3.535f * a + 2.181f * a * a) / (1 + 2.276f * a + 2.577f * a * a);
    return chi * VdotM / NdotV * approx;
}
__device__ inline float G_SmithGGX_IncludeFraction(float NdotL, float NdotV, float alphaG)
{
    const float a2 = alphaG * alphaG;
    const float lv = NdotL * sqrtf((-NdotV * a2 + NdotV) * NdotV + a2);
    const float ll = NdotV * sqrtf((-NdotL * a2 + NdotL) * NdotL + a2);
    return 0.5f / (lv + ll);
}
__device__ inline float D_GGX(float NdotH, float alpha)
{
    const float a2 = alpha * alpha;
    const float f = (NdotH * NdotH) * (a2 - 1) + 1;
    return f > kEPS ? a2 / (kPI * f * f) : 1.0f;
}
__device__ inline float Fr_DisneyDiffuse(float NdotV, float NdotL, float LdotH, float linearRoughness)
{
    const float energyBias = 0.0f + (0.5f - 0.0f) * linearRoughness;
    const float energyFactor = 1.0f + (1.0f / 1.51f - 1.0f) * linearRoughness;
    const float fd90 = energyBias + 2.0f * LdotH * LdotH * linearRoughness;
    const float lightScatter = 1.0f + (fd90 - 1.0f) * pow5(1.0f - NdotL);
    const float viewScatter = 1.0f + (fd90 - 1.0f) * pow5(1.0f - NdotV);
    return lightScatter * viewScatter * energyFactor;
}

struct MatView {
    V3 colour;
    float p0, p1; // smoothness | iorBasic ; f0NonMetal | iorRough
    int texId;
    bool metallic;
    int type;
};
// the material as TriFat carries it: (.., colour.xyz) and (params.xyz, bits(type)) of the caller's 48-byte record
__device__ inline MatView loadMaterial(float4 v0c, float4 mat)
{
    MatView v;
    v.colour = mk(v0c.y, v0c.z, v0c.w);
    v.p0 = mat.x;
    v.p1 = mat.y;
    v.texId = (int)asU(mat.x);
    v.metallic = (asU(mat.z) & 0xFFu) != 0u;
    v.type = (int)asU(mat.w);
    return v;
}
__device__ inline V3 pbrF0(const MatView& m) { return m.metallic ? m.colour : mk(m.p1); }

__device__ inline V3 pbrBrdfWithDiffuse(V3 V, V3 L, V3 N, const MatView& m, bool nospecular) // pbr_brdf.cl:131-181
{
    const V3 f0 = pbrF0(m);
    const float roughness = 1.0f - m.p0;
    const float linearRoughness = sqrtf(roughness);
    const float NdotV = fabsf(dot(N, V)) + 1e-5f;
    const V3 H = normalize(V + L);
    const float LdotH = saturate(dot(L, H));
    const float NdotH = saturate(dot(N, H));
    const float NdotL = saturate(dot(N, L));
    const V3 F = F_Schlick(f0, 1.0f, LdotH);
    const float G = G_SmithGGX_IncludeFraction(NdotL, NdotV, roughness);
    const float D = D_GGX(NdotH, roughness);
    const V3 Fr = D * G * F;
    const float Fd = Fr_DisneyDiffuse(NdotV, NdotL, LdotH, linearRoughness) / kPI;
    const V3 diffuseColour = m.metallic ? mk(0.0f) : m.colour;
    const V3 diffuse = (mk(1.0f) - F) * (Fd * diffuseColour);
    return nospecular ? diffuse : Fr + diffuse;
}

__device__ inline float calcWeight(V3 I, V3 N, V3 M, float smoothness, V3 O) // refract.cl:116-134
{
    const float IdotM = fabsf(dot(I, M)), MdotN = fabsf(dot(M, N)), NdotI = fabsf(dot(N, I));
    const float MdotO = fabsf(dot(M, O)), NdotO = fabsf(dot(N, O));
    const float roughness = 1.0f - smoothness;
    float G = G_SmithBeckmannCorrelated(IdotM, NdotI, roughness) * G_SmithBeckmannCorrelated(MdotO, NdotO, roughness);
    G = fmaxf(fminf(G, 4.0f), 0.f);
    const float weight = (IdotM * G) / (NdotI * MdotN);
    return fminf(weight, 4.0f);
}

// M^T (3x3 part of the inverse instance matrix) times v = normal transform (math.cl:22-29).
// Instance rows r_i = (m[i], m[4+i], m[8+i], m[12+i]) so (M^T v)_j = sum_i m[4j+i] v_i.
__device__ inline V3 normalTransform(const Instance& in, V3 v)
{
    const V3 c1 = v.x * mk(in.r0.x, in.r0.y, in.r0.z);
    const V3 c2 = v.y * mk(in.r1.x, in.r1.y, in.r1.z);
    const V3 c3 = v.z * mk(in.r2.x, in.r2.y, in.r2.z);
    return c1 + c2 + c3;
}
__device__ inline V3 orient(V3 sample, V3 normal, V3 tangentSeed, const Instance& in)
{
    const V3 tangent = normalize(cross(normal, tangentSeed));
    const V3 bitangent = cross(normal, tangent);
    const V3 os = sample.x * tangent + sample.y * bitangent + sample.z * normal;
    return normalize(normalTransform(in, os));
}
__device__ inline V3 cosineWeightedDiffuseReflection(V3 normal, V3 edge1, const Instance& in, Rng& rng) // shading_helper.cl:62-90
{
    const float r0 = rng.u01(), r1 = rng.u01();
    const float r = sqrtf(r0);
    const float theta = 2 * kPI * r1;
    return normalize(orient(mk(r * cosf(theta), r * sinf(theta), sqrtf(1 - r0)), normal, edge1, in));
}

struct ShadeResult {
    V3 radiance; // deposited by shade itself
    float pdf; // MIS: solid-angle density with which the continuation direction was sampled (outData->pdf)
    uint32_t flags; // continuation ray flags
    V3 origin, direction, throughput;
    uint32_t shadowFlags;
    V3 shadowOrigin, shadowDirection, shadowContribution;
    float shadowLength;
};

// Options of the general shading kernel (k_shade<PARITY, true>); the default kernel is neeIsShading with uniform light choice.
struct ShadeOpts {
    bool mis; // neeMisShading (shading.cl:35-349) instead of neeIsShading (:356-623)
    bool weightedLights; // weightedRandomPointOnLight (shading_helper.cl:216-259) instead of randomPointOnLight (:261-278)
    V3 rayOrigin; // MIS: origin of the ray that was traced (distance to an emissive hit, shading.cl:78-79)
    float inPdf; // MIS: inData->pdf
};

// material-texture fetch for diffuseColour (shading_helper.cl:280-307); x = -1 marks an alpha-0 texel.  Without a texture
// array the fetch returns opaque white (what the oracle binds by default): only neeMisShading's PBR light sample can get
// here without one, through its DIFFUSE view of a PBR record (see shadeHit).
// (f0.w, f1.w, f2.w) / (f3.x, f3.y, f3.z): texCoord.x / .y of the three vertices (TriFat)
__device__ inline V3 diffuseColourTextured(const SceneDev& sc, const MatView& mat, float4 f0, float4 f1, float4 f2, float4 f3, float u, float v)
{
    if (!sc.materialTex.texels)
        return mk(1.0f);
    const float t0x = f0.w, t0y = f3.x;
    const float tcx = t0x + (f1.w - t0x) * u + (f2.w - t0x) * v;
    const float tcy = t0y + (f3.y - t0y) * u + (f3.z - t0y) * v;
    const float4 c = sampleLinearRepeat(sc.materialTex, tcx, tcy, (float)mat.texId);
    return (c.w == 0.0f) ? mk(-1.0f) : xyz(c);
}

// weightedRandomPointOnLight, shading_helper.cl:216-259: the light is chosen with probability proportional to the solid
// angle its centroid direction gives it (area * cos / dist^2, capped at 2 pi, not clamped at zero), one draw; the colour
// carries weightTotal / numLights.  The reference keeps the weights in a 255-entry private array; here they are computed
// twice (same arithmetic, same values) so that no per-lane array is needed.
__device__ inline float lightWeight(const Light& lt, V3 X)
{
    const V3 centroid = (xyz(lt.v2) + xyz(lt.v1) + xyz(lt.v0)) / 3.0f;
    V3 L = centroid - X;
    const float dist2 = dot(L, L);
    L = L / sqrtf(dist2);
    const float solidAngle = (dot(xyz(lt.normal), -L) * lt.v0.w) / dist2;
    return 2 * kPI < solidAngle ? 2 * kPI : solidAngle; // OpenCL min(2 pi, x)
}
__device__ inline int pickWeightedLight(const SceneDev& sc, V3 X, Rng& rng, float* weightTotalOut)
{
    const int n = (int)sc.numLights;
    float weightTotal = 0;
    for (int i = 0; i < n; i++)
        weightTotal += lightWeight(sc.lights[i], X);
    float randomValue = rng.u01() * weightTotal;
    int li;
    for (li = 0; li < n; ++li) {
        randomValue -= lightWeight(sc.lights[li], X);
        if (randomValue <= 0)
            break;
    }
    *weightTotalOut = weightTotal;
    return li; // == n when the walk falls off the end (negative weights): the caller treats that light as black, like the zeroed
               // record the reference then reads
}

// neeIsShading, shading.cl:356-623; with GENERAL also neeMisShading, :35-349 (the two differ in four places, marked MIS below;
// what is kept and what is fixed of the reference's MIS code: oracle/oracle.cpp, neeShading) and weighted light choice
template <bool GENERAL>
__device__ inline void shadeHit(const SceneDev& sc, V3 X, V3 D, float t, float u, float v, uint32_t prim, uint32_t instIdx,
    V3 throughput, uint32_t inFlags, Rng& rng, ShadeResult& out, const ShadeOpts& opt, float* park = nullptr, uint32_t parkStride = 0)
{
    const bool MIS = GENERAL && opt.mis;
    const TriFat* fp = &sc.triFat[prim];
    const float4 f0 = fp->n0u, f1 = fp->n1u, f2 = fp->n2u, f3 = fp->vvvm, f4 = fp->e1e, f5 = fp->e2v, f6 = fp->v0c, f7 = fp->mat;
    const V3 edge1 = xyz(f4), edge2 = mk(f4.w, f5.x, f5.y);
    const Instance in = sc.instances[instIdx];
    const V3 realNormal = normalize(normalTransform(in, cross(edge1, edge2)));
    const V3 n0 = xyz(f0), n1 = xyz(f1), n2 = xyz(f2);
    const V3 shadingNormal = normalize(n0 + (n1 - n0) * u + (n2 - n0) * v); // object space, not instance-transformed (shading.cl:378)
    V3 raySideNormal = shadingNormal;
    if (dot(raySideNormal, -D) < 0.0f)
        raySideNormal = raySideNormal * -1.0f;
    const MatView mat = loadMaterial(f6, f7);
    out.radiance = mk(0.0f);
    out.flags = 0;
    out.pdf = 0.f;
    out.shadowFlags = FLAG_FINISHED;

    if (mat.type == MAT_EMISSIVE) { // :387-397
        out.flags = FLAG_FINISHED;
        if (inFlags & FLAG_LASTSPECULAR) {
            out.radiance = throughput * mat.colour;
        } else if (MIS) { // shading.cl:69-90: a BSDF-sampled direction found the light: balance heuristic against NEE's density
            const V3 v0 = mk(f5.z, f5.w, f6.x), v1 = v0 + edge1, v2 = v0 + edge2; // object space, as the reference
            const V3 A = v1 - v0, B = v2 - v1, C = v0 - v2;
            const float la = sqrtf(dot(A, A)), lb = sqrtf(dot(B, B)), lc = sqrtf(dot(C, C));
            const float hs = (la + lb + lc) / 2.0f;
            const float lightArea = sqrtf(hs * (hs - la) * (hs - lb) * (hs - lc));
            const V3 distV = X - opt.rayOrigin;
            float solidAngle = (dot(realNormal, -D) * lightArea) / dot(distV, distV);
            solidAngle = 2 * kPI < solidAngle ? 2 * kPI : solidAngle;
            const float pdf2 = opt.inPdf;
            if (solidAngle > kEPS && !(pdf2 < kEPS)) {
                const float pdf1 = 1 / solidAngle;
                out.radiance = throughput * mat.colour * (pdf2 / (pdf1 + pdf2));
            }
        }
        return;
    }

    // diffuse albedo lookup shared by NEE and the continuation (shading_helper.cl:280-307); x=-1: alpha-0 texel
    V3 albedo = mat.colour;
    if (mat.type == MAT_DIFFUSE && mat.texId != -1) {
        const float t0x = f0.w, t0y = f3.x;
        const float tcx = t0x + (f1.w - t0x) * u + (f2.w - t0x) * v;
        const float tcy = t0y + (f3.y - t0y) * u + (f3.z - t0y) * v;
        const float4 c = sampleLinearRepeat(sc.materialTex, tcx, tcy, (float)mat.texId);
        albedo = (c.w == 0.0f) ? mk(-1.0f) : xyz(c);
    }

    V3 BRDF = mk(0.0f);
    if (mat.type != MAT_REFRACTIVE && mat.type != MAT_BASIC_REFRACTIVE) { // NEE, :399-448
        int li;
        float colourScale = 1.0f;
        if (GENERAL && opt.weightedLights) {
            float weightTotal;
            li = pickWeightedLight(sc, X, rng, &weightTotal);
            colourScale = weightTotal / (float)sc.numLights;
        } else {
            li = rng.randomInteger(0, (int)sc.numLights - 1);
        }
        Light lt = sc.lights[min(li, (int)sc.numLights - 1)];
        if (GENERAL && li >= (int)sc.numLights) // past the end (see pickWeightedLight): a zero-area, black light
            lt.colour = lt.v0 = lt.v1 = lt.v2 = make_float4(0.f, 0.f, 0.f, 0.f), lt.normal = make_float4(NAN, NAN, NAN, 0.f);
        const float u1 = rng.u01(), u2 = rng.u01();
        const V3 lightPos = (1 - sqrtf(u1)) * xyz(lt.v0) + (sqrtf(u1) * (1 - u2)) * xyz(lt.v1) + (sqrtf(u1) * u2) * xyz(lt.v2);
        const V3 lightNormal = xyz(lt.normal);
        const V3 lightColour = GENERAL ? xyz(lt.colour) * colourScale : xyz(lt.colour);
        V3 L = lightPos - X;
        const float dist2 = dot(L, L);
        const float dist = sqrtf(dist2);
        L = L / dist;
        if (dot(shadingNormal, L) > kEPS && dot(realNormal, L) > kEPS && dot(lightNormal, -L) > kEPS) {
            float pdf2 = 0.0f; // MIS: density with which the BSDF sampling below would have produced L
            if (mat.type == MAT_PBR && MIS) { // shading.cl:116-146
                const V3 halfway = normalize(-D + L);
                const V3 F = F_Schlick(pbrF0(mat), 1.0f, saturate(dot(L, halfway)));
                const float rand01 = rng.u01();
                if (!mat.metallic && rand01 > F.x)
                    pdf2 = dot(shadingNormal, L) / kPI; // cosine weighted PDF
                else
                    pdf2 = D_GGX(dot(shadingNormal, halfway), 1.0f - mat.p0);
                // sic: diffuseColour() of a PBR record -- tex_id is the bit pattern of `smoothness`, never -1, so this is a
                // material-texture fetch at a clamped layer (pbrBrdf's value, shading.cl:117, is overwritten there)
                const V3 c = diffuseColourTextured(sc, mat, f0, f1, f2, f3, u, v);
                BRDF = (c.x == -1.0f) ? mk(0.0f) : c / kPI;
            } else if (mat.type == MAT_PBR) {
                BRDF = pbrBrdfWithDiffuse(-D, L, shadingNormal, mat, mat.p0 > kMaxSmoothness);
            } else if (mat.type == MAT_DIFFUSE && MIS) { // shading.cl:148-152
                BRDF = albedo / kPI; // sic: no alpha-0 check in this variant
                pdf2 = dot(realNormal, L) / kPI;
            } else if (mat.type == MAT_DIFFUSE) {
                BRDF = (albedo.x == -1.0f) ? mk(0.0f) : albedo / kPI;
            }
            float solidAngle = 2 * kPI;
            if (dist2 > kEPS) {
                solidAngle = (dot(lightNormal, -L) * lt.v0.w) / dist2;
                solidAngle = fminf(fmaxf(solidAngle, 0.0f), 2 * kPI);
            }
            V3 Ld;
            if (MIS) // shading.cl:153-163
                Ld = (float)sc.numLights * lightColour * BRDF * dot(realNormal, L) / (1 / solidAngle + pdf2);
            else
                Ld = (float)sc.numLights * lightColour * BRDF * solidAngle * dot(shadingNormal, L);
            out.shadowFlags = 0;
            if (park) { // the shadow ray waits in LDS for the workgroup's compaction: ten registers less while the continuation is sampled
                const V3 sc_ = Ld * throughput, so_ = X + L * kEPS;
                park[0 * parkStride] = so_.x, park[1 * parkStride] = so_.y, park[2 * parkStride] = so_.z, park[3 * parkStride] = dist - 2 * kEPS;
                park[4 * parkStride] = L.x, park[5 * parkStride] = L.y, park[6 * parkStride] = L.z;
                park[7 * parkStride] = sc_.x, park[8 * parkStride] = sc_.y, park[9 * parkStride] = sc_.z;
            } else {
                out.shadowContribution = Ld * throughput;
                out.shadowOrigin = X + L * kEPS;
                out.shadowDirection = L;
                out.shadowLength = dist - 2 * kEPS;
            }
        }
    }

    bool dospecular = false;
    float PDF = 1.0f, cosineTerm = 1.0f;
    V3 reflection = mk(0.0f);
    if (mat.type == MAT_PBR) { // :456-496
        const V3 f0 = pbrF0(mat);
        const V3 V = -D;
        const float alpha = 1 - mat.p0;
        // ggxWeightedHalfway, shading_helper.cl:92-125
        const float r0 = rng.u01();
        const float phi = 2.0f * kPI * r0;
        const float r1 = rng.u01();
        // theta = acos(c); the reference then takes cos(pi/2 - theta) = sin(theta) and sin(pi/2 - theta) = cos(theta) = c
        const float cosTheta = sqrtf((1.0f - r1) / ((alpha * alpha - 1.0f) * r1 + 1.0f));
        const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
        const V3 halfway = orient(mk(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta), shadingNormal, mk(1.0f, 0.0f, 0.0f), in);
        reflection = normalize(2 * dot(halfway, V) * halfway - V);
        if (MIS) // the density ggxWeightedImportanceDirection reports (shading_helper.cl:162-175); "MIS needs real PDF", shading.cl:210
            out.pdf = D_GGX(dot(halfway, shadingNormal), alpha);
        cosineTerm = dot(shadingNormal, reflection);
        if (cosineTerm < 0.05f || dot(realNormal, reflection) < kEPS) {
            out.flags = FLAG_FINISHED;
            return;
        }
        const float LdotH = saturate(dot(reflection, halfway));
        const V3 F = F_Schlick(f0, 1.0f, LdotH);
        const float rand01 = rng.u01();
        const float roughness = 1.0f - mat.p0;
        if (!mat.metallic && rand01 > F.x) {
            reflection = cosineWeightedDiffuseReflection(shadingNormal, edge1, in, rng);
            PDF = kINVPI;
            cosineTerm = 1.0f;
            if (MIS)
                out.pdf = dot(shadingNormal, reflection) * kINVPI; // MIS needs the real unsimplified PDF (shading.cl:205)
            // diffuseOnly, pbr_brdf.cl:217-237
            const float NdotV = fabsf(dot(shadingNormal, V)) + 1e-5f;
            const float Fd = Fr_DisneyDiffuse(NdotV, saturate(dot(shadingNormal, reflection)), saturate(dot(reflection, halfway)), sqrtf(roughness));
            BRDF = Fd * mat.colour / kPI;
        } else {
            PDF = 1.0f;
            // brdfOnlyNoFresnelNoNDF, pbr_brdf.cl:194-213
            const float NdotV = fabsf(dot(shadingNormal, V)) + 1e-5f;
            const float G = G_SmithGGX_IncludeFraction(saturate(dot(shadingNormal, reflection)), NdotV, roughness);
            BRDF = mk(fminf(G, 10.0f));
            if (mat.metallic)
                BRDF = BRDF * F;
            if (mat.p0 > kMaxSmoothness)
                dospecular = true;
        }
    } else if (mat.type == MAT_BASIC_REFRACTIVE) { // :497-538
        V3 absorptionFactor = mk(1.0f);
        float n1, n2;
        if (dot(realNormal, -D) > kEPS) {
            n1 = kAirIor;
            n2 = mat.p0;
        } else {
            n1 = mat.p0;
            n2 = kAirIor;
            const V3 e = -mat.colour * t;
            absorptionFactor = mk(expf(e.x), expf(e.y), expf(e.z));
        }
        const float cos1 = dot(raySideNormal, -D);
        const float n1n2 = n1 / n2;
        const float K = 1 - (n1n2 * n1n2) * (1 - cos1 * cos1);
        if (K > kEPS) {
            const float rand01 = rng.u01();
            const float f0 = ((n1 - n2) / (n1 + n2)) * ((n1 - n2) / (n1 + n2));
            const V3 F = F_Schlick(mk(f0), 1.0f, dot(raySideNormal, -D));
            if (rand01 < F.x)
                reflection = normalize(-D - 2 * dot(-D, raySideNormal) * raySideNormal); // sic (shading.cl:522)
            else
                reflection = normalize(n1n2 * D + raySideNormal * (n1n2 * cos1 - sqrtf(K)));
        } else {
            reflection = normalize(-D - 2 * dot(-D, raySideNormal) * raySideNormal);
        }
        BRDF = absorptionFactor;
    } else if (mat.type == MAT_REFRACTIVE) { // :539-586
        // beckmannWeightedHalfway, shading_helper.cl:127-160
        const float alpha = (1.2f - 0.2f * sqrtf(fabsf(dot(D, raySideNormal)))) * (1 - mat.p0);
        const float r0 = rng.u01(), r1 = rng.u01();
        const float phi = 2.0f * kPI * r0;
        // theta = atan(x): cos(theta) = 1 / sqrt(1 + x^2), sin(theta) = x / sqrt(1 + x^2)
        const float tanTheta = -alpha * alpha * log1pf(-r1);
        const float cosTheta = 1.0f / sqrtf(1.0f + tanTheta * tanTheta), sinTheta = tanTheta * cosTheta;
        const V3 halfway = orient(mk(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta), raySideNormal, mk(1.0f, 0.0f, 0.0f), in);
        V3 absorptionFactor = mk(1.0f);
        float n_i, n_t;
        if (dot(realNormal, -D) > 0.0f) {
            n_i = kAirIor;
            n_t = mat.p1;
        } else {
            n_i = mat.p1;
            n_t = kAirIor;
            const V3 e = -mat.colour * t;
            absorptionFactor = mk(expf(e.x), expf(e.y), expf(e.z));
        }
        const float f0 = ((n_i - n_t) / (n_i + n_t)) * ((n_i - n_t) / (n_i + n_t));
        const V3 F = F_Schlick(mk(f0), 1.0f, dot(-D, halfway));
        const float rand01 = rng.u01();
        const V3 I = -D;
        bool refract = false;
        float n1n2 = 0.f, cos1 = 0.f, K = 0.f;
        if (!(rand01 < F.x)) {
            n1n2 = n_i / n_t;
            cos1 = dot(halfway, I);
            K = 1 - (n1n2 * n1n2) * (1 - cos1 * cos1);
            refract = K >= 0;
        }
        if (refract)
            reflection = normalize(-n1n2 * I + halfway * (n1n2 * cos1 - sqrtf(K))); // evaluateRefract, refract.cl:142-153
        else
            reflection = normalize(I - 2 * dot(I, halfway) * halfway); // evaluateReflect, refract.cl:136-140
        const float w = calcWeight(I, raySideNormal, halfway, mat.p0, reflection);
        BRDF = mk(w) * absorptionFactor;
    } else if (mat.type == MAT_DIFFUSE) { // :587-601
        if (albedo.x == -1.0f) {
            reflection = D;
            BRDF = mk(1.0f);
        } else {
            reflection = cosineWeightedDiffuseReflection(realNormal, edge1, in, rng);
            if (MIS) // taken from the direction just sampled; the reference reads `reflection` before assigning it (:590-592)
                out.pdf = dot(shadingNormal, reflection) * kINVPI;
            BRDF = albedo;
        }
    }

    // Russian roulette + spawn, :606-622
    out.flags = (mat.type == MAT_REFRACTIVE || mat.type == MAT_BASIC_REFRACTIVE || dospecular) ? FLAG_LASTSPECULAR : 0u;
    const V3 integral = BRDF * cosineTerm / PDF;
    const float survive = saturate(fmaxf(fmaxf(integral.x, integral.y), integral.z));
    const float choice = rng.u01();
    if (survive < kEPS || choice > survive) {
        out.flags = FLAG_FINISHED;
        return;
    }
    out.origin = X + reflection * kEPS;
    out.direction = reflection;
    out.throughput = throughput * integral / survive;
}

// =================================================================================================
// kernels
// =================================================================================================
struct FrameParams {
    CameraDev cam;
    uint32_t width, height;
    uint32_t sample, seed;
    uint32_t maxBounces;
    uint32_t parity; // LFSR113 streams per slot + reference queue semantics
    uint32_t numOwned; // pixels owned by this context
    uint32_t planes; // samples in flight (fixed schedule only; 1 otherwise)
    uint32_t interleave; // k_gen: samples of one pixel in consecutive queue entries (a power of two dividing planes)
    uint32_t interleaveShift; // log2 of it
    float invWidth;
    uint32_t integrator; // 0: neeIsShading; 1: neeMisShading; 2: COMPARE_SHADING -- MIS for the pixels of the left half of the
                         // image, IS for the right half, both halves showing the left half's view (kernel.cl:48-51,248-265)
    uint32_t weightedLights; // NEE picks its light by weightedRandomPointOnLight instead of randomPointOnLight
    float invSpan; // 1 / (numOwned << interleaveShift): entry index -> sample group without an integer division (primaryEntry)
};
enum : uint32_t { INTEGRATOR_IS = 0, INTEGRATOR_MIS = 1, INTEGRATOR_COMPARE = 2 };

// Entry i of the FIRST queue of a batch with several samples in flight: which (pixel, accumulator plane = sample of the batch) it
// is.  Consecutive entries are `interleave` samples of one pixel, then the next owned pixel; after every owned pixel the next
// group of samples (k_gen).  Used by the packet kernel, which generates the camera rays of its packet itself (and queues them for
// k_shade) instead of reading what a k_gen launch would have written first.
__device__ inline void primaryEntry(const FrameParams& fp, const uint32_t* __restrict__ pixelList, uint32_t i, uint32_t* pixel, uint32_t* plane)
{
    const uint32_t span = fp.numOwned << fp.interleaveShift;
    uint32_t group = (uint32_t)((float)i * fp.invSpan); // float estimate, then exact correction
    if (group * span > i)
        group--;
    if ((group + 1u) * span <= i)
        group++;
    const uint32_t r = i - group * span;
    *plane = (group << fp.interleaveShift) + (r & ((1u << fp.interleaveShift) - 1u));
    const uint32_t k = r >> fp.interleaveShift;
    *pixel = pixelList ? pixelList[k] : k;
}
// the camera ray of (pixel, sample fp.sample + plane): origin, direction (un-normalised for a thin lens, camera.cl:71-75)
__device__ inline void primaryRay(const FrameParams& fp, uint32_t pixel, uint32_t plane, V3* o, V3* d)
{
    // pixel -> (x, y) without an integer division: float estimate, then exact correction (pixel < 2^31)
    uint32_t py = (uint32_t)((float)pixel * fp.invWidth);
    if (py * fp.width > pixel)
        py--;
    if ((py + 1u) * fp.width <= pixel)
        py++;
    uint32_t px = pixel - py * fp.width;
    if (fp.integrator == INTEGRATOR_COMPARE && px >= fp.width / 2u) // COMPARE_SHADING, kernel.cl:48-51
        px -= fp.width / 2u;
    Rng rng = rngCounter(pixel, fp.sample + plane, fp.seed, 0u);
    cameraRay(fp.cam, (int)px, (int)py, (float)fp.width, (float)fp.height, rng, o, d);
}

// start of a batch whose primary rays are regenerated where they are needed instead of queued (k_gen's bookkeeping)
__global__ void k_begin_batch(uint32_t* queueCount, uint32_t* generated, uint32_t n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *queueCount = n;
        *generated += n;
    }
}

// generatePrimaryRays, kernel.cl:24-84.  Thread i creates the ray of the i-th (pixel, sample) pair to
// issue into queue slot slotBase + i.  With one sample in flight: pixel = pixelList[first + i] (or the index itself
// when no list is set).  With several: the samples of a pixel are neighbours in the queue (a wave of 64 primary rays
// then walks the tree almost in lock-step, and so do the shadow rays its hits spawn): sample = fp.sample + plane.
__global__ void __launch_bounds__(256) k_gen(FrameParams fp, RayQueue q, const uint32_t* __restrict__ pixelList, uint32_t first,
    uint32_t n, uint32_t slotBase, uint4* __restrict__ streams, uint32_t* __restrict__ queueCount, uint32_t* __restrict__ generated)
{
    // With several samples in flight the grid is 2-D: blockIdx.y = group of `interleave` samples, x over that group's
    // `interleave * numOwned` entries -- the index arithmetic is shifts and masks (five 32-bit divisions were two
    // thirds of this kernel's instructions).
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = r, plane = 0, k = first + r;
    if (fp.planes > 1u) {
        // consecutive entries = `interleave` (a power of two) samples of ONE pixel, then the next pixel; after every
        // owned pixel, the next group of samples
        const uint32_t span = fp.numOwned << fp.interleaveShift;
        if (r >= span)
            return;
        i = blockIdx.y * span + r;
        plane = (blockIdx.y << fp.interleaveShift) + (r & ((1u << fp.interleaveShift) - 1u));
        k = first + (r >> fp.interleaveShift);
    }
    if (i == 0) { // the queue now holds the surviving rays [0, slotBase) plus n new ones
        *queueCount = slotBase + n;
        *generated += n;
    }
    if (i >= n)
        return;
    const uint32_t pixel = pixelList ? pixelList[k] : k;
    // pixel -> (x, y) without an integer division: float estimate, then exact correction (pixel < 2^31)
    uint32_t py = (uint32_t)((float)pixel * fp.invWidth);
    if (py * fp.width > pixel)
        py--;
    if ((py + 1u) * fp.width <= pixel)
        py++;
    uint32_t px = pixel - py * fp.width;
    if (fp.integrator == INTEGRATOR_COMPARE && px >= fp.width / 2u) // COMPARE_SHADING, kernel.cl:48-51
        px -= fp.width / 2u;
    Rng rng = fp.parity ? rngLfsrLoad(streams, i) : rngCounter(pixel, fp.sample + plane, fp.seed, 0u);
    V3 o, d;
    if (fp.parity)
        cameraRayPrecise(fp.cam, (int)px, (int)py, (float)fp.width, (float)fp.height, rng, &o, &d);
    else
        cameraRay(fp.cam, (int)px, (int)py, (float)fp.width, (float)fp.height, rng, &o, &d);
    if (fp.parity)
        rngLfsrStore(streams, i, rng);
    const uint32_t slot = slotBase + i;
    q.o[slot] = make_float4(o.x, o.y, o.z, asF(pixel));
    q.d[slot] = make_float4(d.x, d.y, d.z, asF(packState(FLAG_LASTSPECULAR, 0u, plane)));
    if (fp.parity) // production mode: a primary ray's throughput is 1 by definition, k_shade does not read it (16 B per ray less each way)
        q.thr[slot] = make_float4(1.f, 1.f, 1.f, 0.f);
}

struct ShadeArgs {
    SceneDev sc;
    FrameParams fp;
    RayQueue in;
    HitQueue hits;
    RayQueue out;
    ShadowQueue shadow;
    AccumView accum;
    const uint32_t* inCount;
    uint32_t* outCount;
    uint32_t* shadowCount;
    uint32_t* shadeHits;
    uint32_t* deposits; // accumulator updates made here (emissive hits seen through a specular chain, sky misses)
    uint4* streams; // parity mode
    // parity mode: un-compacted staging + active flags for the stable compaction pass
    uint32_t* activeFlag;
    uint32_t firstTile; // first 512-entry tile this launch is responsible for
    uint32_t derivedPrimaries; // pass 0 behind k_trace_multi (pinhole): the queue holds (direction, pixel) only -- origin = the eye, sample from the entry index
    // entries the out queue / the shadow queue hold (round 6: they may be smaller than the batch -- ptamd.hip sizes a batch so that what its first pass emits
    // fits, from the counts of earlier batches; the guard below and k_clamp_counts turn a wrong guess into a reported error instead of a write past the end)
    uint32_t outCap, shadowCap;
};

// shade, kernel.cl:190-301.  PARITY = reference queue semantics: every shaded hit is enqueued in both
// output queues, finished or not (kernel.cl:292-300), at the SAME index, in slot order.
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 512
#endif
#ifndef PT_SHADE_MIN_WAVES
#define PT_SHADE_MIN_WAVES 6 // waves per SIMD of the production instantiation: 3 workgroups of 512 per CU need <= 80 VGPRs -- 94 before the shadow
#endif                       // ray of an entry was parked in LDS (33 spilled at 80, 35.5 vs 27.1 ms); 80 with 2 spilled after: k_shade 24.9 -> 23.0 ms.
                             // The general (MIS / COMPARE_SHADING) and parity instantiations stay at 4 waves (102-110 VGPRs)
constexpr int kShadeBlock = PT_SHADE_BLOCK;
#ifndef PT_SHADE_PARK
#define PT_SHADE_PARK 1 // 1: the shadow ray of an entry waits in LDS (10 dwords per thread) instead of registers while the continuation is sampled
#endif
#ifndef PT_SHADE_NT
#define PT_SHADE_NT 1 // 1: queue entries are read / written with non-temporal accesses (streamed once: no reason to keep them in L2 next to the scene)
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ inline float4 ldQ(const float4* p)
{
#if PT_SHADE_NT
    const f4v v = __builtin_nontemporal_load((const f4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ inline void stQ(float4* p, float4 v)
{
#if PT_SHADE_NT
    const f4v w = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(w, (f4v*)p);
#else
    *p = v;
#endif
}

// GENERAL = false: the integrator the reference compiles in (neeIsShading, uniform light choice) -- the production kernel;
// GENERAL = true: integrator and light choice selected by a.fp at run time (MIS, COMPARE_SHADING, weighted lights).
// The launch does not know how many entries are live (the count is a device word: no read-back anywhere in the schedule) and
// covers the queue's capacity, while the queue of a later bounce holds 5-30 % of that: a workgroup whose tile lies beyond the
// live entries leaves at once, before any ballot, LDS traffic or barrier.  (A grid sized to the machine with every workgroup
// walking tiles in a loop was built too: the loop keeps the scene pointers live across iterations -- 128 instead of 96 VGPRs and
// spills; used only for the later passes of a batch, where 75-99 % of the workgroups leave at once -- 0.6 ms of dispatch per launch --
// it still loses: k_shade 26.8 instead of 25.75 ms per batch.)
// LOOP = true: a small grid whose workgroups walk tiles from a.firstTile on (tile += gridDim.x) -- the safety net behind a launch that
// covers only the head of the queue (ptamd.hip, launchShade).  Slower per entry than the one-tile kernel (128 VGPRs and spills: the
// loop keeps the scene pointers live), so it is never the main path.
// BINNED = true (opt-in, PT_FLAG_MATERIAL_BINS, for scenes whose surfaces are of more than one material type; MEASURED SLOWER than queue
// order even with five types per triangle at random -- 34.9 vs 31.1 ms per batch: the kernel is bound by its gathers and barriers, not by
// divergent BSDF code): the 512 entries of a tile are shaded in MATERIAL ORDER.  The
// reference dispatches on the material inside one kernel (shading.cl:387-601) and so does shadeHit: a wave whose 64 hits are of three
// types runs three BSDFs one after the other.  Here every thread first looks up the type of its entry's hit (4 bytes of the triangle's
// shading record), the workgroup counting-sorts its tile by (type, miss) through LDS -- ballot ranks inside a wave, one 64-lane scan over
// the (type, wave) counters -- and thread t then shades the t-th entry of that order: waves see one type except at the seams.  Nothing
// moves in memory; the compaction below writes the outputs wherever its counters say, as always.
template <bool PARITY, bool GENERAL = false, bool LOOP = false, bool BINNED = false>
__global__ void __launch_bounds__(kShadeBlock, (PARITY || GENERAL || LOOP) ? 4 : PT_SHADE_MIN_WAVES) k_shade(ShadeArgs a)
{
    static_assert(!(PARITY && LOOP), "the parity kernel stages at the input slot: one tile per workgroup");
    static_assert(!BINNED || (!PARITY && !LOOP), "material order: production queue semantics, one tile per workgroup");
    const uint32_t count = *a.inCount;
    uint32_t tile = a.firstTile + blockIdx.x;
    if (tile * kShadeBlock >= count) // uniform for the workgroup
        return;
    __shared__ uint32_t sCount[kShadeBlock / 64][4];
    __shared__ uint32_t sBase[4];
  do {
    uint32_t i = tile * kShadeBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (BINNED) {
        constexpr uint32_t kWaves = kShadeBlock / 64u, kKeys = 64u / kWaves; // (8 keys with the 512-thread workgroups of the build; six are in use)
        static_assert(kKeys * kWaves == 64u, "one lane per (key, wave) counter");
        __shared__ uint32_t sBin[kKeys * kWaves]; // [key][wave]
        __shared__ uint16_t sPerm[kShadeBlock];
        const uint32_t w = threadIdx.x >> 6;
        uint32_t key = kKeys - 1u; // beyond the live entries: sorts last
        if (i < count) {
            const int prim = (int)((const uint32_t*)&a.hits.h[i])[3];
            key = prim < 0 ? 5u : min(asU(a.sc.triFat[prim].mat.w), 4u); // material type (MAT_*), 5: the ray left the scene
        }
        uint32_t rank = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kKeys; k++) {
            const unsigned long long m = __ballot(key == k);
            if (key == k)
                rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0)
                sBin[k * kWaves + w] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < 64u) { // exclusive scan of the 64 counters, key-major
            const uint32_t v = sBin[threadIdx.x];
            uint32_t incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if ((int)lane >= o)
                    incl += up;
            }
            sBin[threadIdx.x] = incl - v;
        }
        __syncthreads();
        sPerm[sBin[key * kWaves + w] + rank] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = tile * kShadeBlock + sPerm[threadIdx.x];
    }
#if PT_SHADE_PARK
    __shared__ float sPark[PARITY ? 1 : 10][PARITY ? 1 : kShadeBlock];
#endif
    bool emitRay = false, emitShadow = false, shaded = false, deposited = false;
    ShadeResult r;
    uint32_t pixel = 0, bounce = 0, plane = 0;
    if (i < count) {
        const float4 rd = ldQ(&a.in.d[i]);
        float4 ro;
        uint32_t fb;
        if (!PARITY && a.derivedPrimaries) { // (uniform) camera rays of a pinhole as k_trace_multi queued them
            ro = a.fp.cam.eye;
            pixel = asU(rd.w);
            // the sample of the batch this entry is (k_gen's order, primaryEntry): group of `interleave` samples = i / span, sample inside it = the low bits
            const uint32_t span = a.fp.numOwned << a.fp.interleaveShift;
            uint32_t group = (uint32_t)((float)i * a.fp.invSpan);
            if (group * span > i)
                group--;
            if ((group + 1u) * span <= i)
                group++;
            fb = packState(FLAG_LASTSPECULAR, 0u, (group << a.fp.interleaveShift) + (i & ((1u << a.fp.interleaveShift) - 1u)));
        } else {
            ro = ldQ(&a.in.o[i]);
            fb = asU(rd.w);
            pixel = asU(ro.w);
        }
        bounce = (fb >> 8) & 0xFFu;
        plane = fb >> 16;
        if (!(fb & FLAG_FINISHED)) {
            const float4 h = ldQ(&a.hits.h[i]);
            float4 thr = make_float4(1.f, 1.f, 1.f, 0.f);
            if (PARITY || bounce != 0u) // primary rays: 1, not stored (k_gen)
                thr = ldQ(&a.in.thr[i]);
            const V3 o = xyz(ro), d = xyz(rd), throughput = xyz(thr);
            const int prim = (int)asU(h.w);
            if (prim >= 0) {
                shaded = true;
                const V3 X = o + h.x * d;
                Rng rng = PARITY ? rngLfsrLoad(a.streams, i) : rngCounter(pixel, a.fp.sample + plane, a.fp.seed, 1u + bounce);
                ShadeOpts opt;
                opt.mis = opt.weightedLights = false;
                if (GENERAL) {
                    opt.mis = a.fp.integrator == INTEGRATOR_MIS
                        || (a.fp.integrator == INTEGRATOR_COMPARE && (pixel % a.fp.width) < a.fp.width / 2u);
                    opt.weightedLights = a.fp.weightedLights != 0u;
                    opt.rayOrigin = o;
                    opt.inPdf = thr.w; // 0 for primary rays (they carry LASTSPECULAR: never consulted)
                }
#if PT_SHADE_PARK
                shadeHit<GENERAL>(a.sc, X, normalize(d), h.x, h.y, h.z, (uint32_t)prim, (uint32_t)a.hits.inst[i], throughput, fb & 0xFFu, rng, r, opt,
                    PARITY ? nullptr : &sPark[0][threadIdx.x], PARITY ? 0u : (uint32_t)kShadeBlock);
#else
                shadeHit<GENERAL>(a.sc, X, normalize(d), h.x, h.y, h.z, (uint32_t)prim, (uint32_t)a.hits.inst[i], throughput, fb & 0xFFu, rng, r, opt);
#endif
                if (PARITY)
                    rngLfsrStore(a.streams, i, rng);
                if (r.radiance.x != 0.f || r.radiance.y != 0.f || r.radiance.z != 0.f) {
                    float4* ap = a.accum.at(plane, pixel);
                    float4 px = *ap;
                    px.x += r.radiance.x, px.y += r.radiance.y, px.z += r.radiance.z;
                    *ap = px;
                    deposited = true;
                }
                bounce += 1;
                if (bounce >= a.fp.maxBounces) // kernel.cl:295-296
                    r.flags = FLAG_FINISHED;
                emitRay = PARITY ? true : !(r.flags & FLAG_FINISHED);
                emitShadow = PARITY ? true : !(r.shadowFlags & FLAG_FINISHED);
            } else { // miss: skydome (kernel.cl:285-289)
                // (the accumulator entry is fetched BEFORE the sky is looked up: one dependent round trip to memory less)
                float4* ap = a.accum.at(plane, pixel);
                float4 px = *ap;
                asm volatile("" : "+v"(px.x), "+v"(px.y), "+v"(px.z), "+v"(px.w)); // keeps the load where it is written
                const V3 c = throughput * readSkydome(a.sc, normalize(d));
                px.x += c.x, px.y += c.y, px.z += c.z;
                *ap = px;
                deposited = true;
            }
        }
    }
    if (PARITY) {
        const uint32_t nDep = (uint32_t)__popcll(__ballot(deposited));
        if (lane == 0 && nDep)
            atomicAdd(a.deposits, nDep);
        // stage at the input slot; k_compact_stable assigns output indices in slot order
        if (i < count)
            a.activeFlag[i] = shaded ? 1u : 0u;
        if (shaded) {
            a.out.o[i] = make_float4(r.origin.x, r.origin.y, r.origin.z, asF(pixel));
            a.out.d[i] = make_float4(r.direction.x, r.direction.y, r.direction.z, asF(packState(r.flags, bounce, plane)));
            a.out.thr[i] = make_float4(r.throughput.x, r.throughput.y, r.throughput.z, GENERAL ? r.pdf : 0.f);
            a.shadow.o[i] = make_float4(r.shadowOrigin.x, r.shadowOrigin.y, r.shadowOrigin.z, r.shadowLength);
            a.shadow.d[i] = make_float4(r.shadowDirection.x, r.shadowDirection.y, r.shadowDirection.z, asF(pixel));
            a.shadow.c[i] = make_float4(r.shadowContribution.x, r.shadowContribution.y, r.shadowContribution.z, asF(packState(r.shadowFlags, 0u, plane)));
        }
        return;
    }
    // Workgroup-aggregated compaction: ballots rank the lanes of a wave, the waves of the block exchange
    // their counts through LDS, and ONE lane per queue issues the atomicAdd for the whole block.  One atomic
    // per wave was not enough: a single device-scope word sustains ~90-100 atomics/us, so the 1 M waves of a
    // first-bounce launch would spend > 10 ms on it.  The block size trades that atomic rate (one per
    // kShadeBlock entries) against waves idling at the two barriers: 1024 / 768 / 640 / 512 / 384 / 256 threads ->
    // 17.8 / 20.1 / 22.3 / 14.3 / 17.4 / 16.3 ms of k_shade per 128-sample batch at 1080p (256-thread blocks sit
    // exactly on the atomic rate: 1.36 M blocks per word in 16 ms = 85 atomics/us).
    const unsigned long long mRay = __ballot(emitRay);
    const unsigned long long mSh = __ballot(emitShadow);
    const unsigned long long mHit = __ballot(shaded);
    const unsigned long long mDep = __ballot(deposited);
    const uint32_t wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    if (lane == 0) {
        sCount[wave][0] = (uint32_t)__popcll(mRay);
        sCount[wave][1] = (uint32_t)__popcll(mSh);
        sCount[wave][2] = (uint32_t)__popcll(mHit);
        sCount[wave][3] = (uint32_t)__popcll(mDep);
    }
    __syncthreads();
    if (threadIdx.x < 4) { // thread q: exclusive prefix of queue q over the waves + the block's atomic
        uint32_t sum = 0;
        for (uint32_t w = 0; w < nWaves; w++) {
            const uint32_t n = sCount[w][threadIdx.x];
            sCount[w][threadIdx.x] = sum;
            sum += n;
        }
        uint32_t* counter = threadIdx.x == 0 ? a.outCount : (threadIdx.x == 1 ? a.shadowCount : (threadIdx.x == 2 ? a.shadeHits : a.deposits + (blockIdx.x % kDepositSlots) * 16u));
        sBase[threadIdx.x] = sum ? atomicAdd(counter, sum) : 0u;
    }
    __syncthreads();
    const uint32_t baseRay = sBase[0] + sCount[wave][0], baseSh = sBase[1] + sCount[wave][1];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (emitRay && baseRay + (uint32_t)__popcll(mRay & below) < a.outCap) {
        const uint32_t idx = baseRay + (uint32_t)__popcll(mRay & below);
        stQ(&a.out.o[idx], make_float4(r.origin.x, r.origin.y, r.origin.z, asF(pixel)));
        stQ(&a.out.d[idx], make_float4(r.direction.x, r.direction.y, r.direction.z, asF(packState(r.flags, bounce, plane))));
        stQ(&a.out.thr[idx], make_float4(r.throughput.x, r.throughput.y, r.throughput.z, GENERAL ? r.pdf : 0.f));
    }
    if (emitShadow && baseSh + (uint32_t)__popcll(mSh & below) < a.shadowCap) {
        const uint32_t idx = baseSh + (uint32_t)__popcll(mSh & below);
#if PT_SHADE_PARK
        const float* pk = &sPark[0][PARITY ? 0 : threadIdx.x];
        constexpr uint32_t S = PARITY ? 0 : kShadeBlock;
        stQ(&a.shadow.o[idx], make_float4(pk[0 * S], pk[1 * S], pk[2 * S], pk[3 * S]));
        stQ(&a.shadow.d[idx], make_float4(pk[4 * S], pk[5 * S], pk[6 * S], asF(pixel)));
        stQ(&a.shadow.c[idx], make_float4(pk[7 * S], pk[8 * S], pk[9 * S], asF(packState(0u, 0u, plane))));
#else
        stQ(&a.shadow.o[idx], make_float4(r.shadowOrigin.x, r.shadowOrigin.y, r.shadowOrigin.z, r.shadowLength));
        stQ(&a.shadow.d[idx], make_float4(r.shadowDirection.x, r.shadowDirection.y, r.shadowDirection.z, asF(pixel)));
        stQ(&a.shadow.c[idx], make_float4(r.shadowContribution.x, r.shadowContribution.y, r.shadowContribution.z, asF(packState(0u, 0u, plane))));
#endif
    }
    if (!LOOP)
        break;
    __syncthreads(); // the counters in LDS are reused by the next tile
    tile += gridDim.x;
  } while (tile * kShadeBlock < count);
}

// Parity mode: stable (slot-ordered) compaction of the staged shade outputs -- the order the
// reference's atomic_inc compaction takes when work-items run in gid order (oracle/_ref).
// One workgroup walks the queue in 1024-entry tiles with an LDS scan.
struct CompactArgs {
    RayQueue staged, out;
    ShadowQueue stagedShadow, outShadow;
    const uint32_t* activeFlag;
    const uint32_t* inCount;
    uint32_t* outCount; // rays and shadow rays share the index (kernel.cl:292-300)
    uint32_t* shadowCount;
    uint32_t* shadeHits;
};
__global__ void __launch_bounds__(1024) k_compact_stable(CompactArgs a)
{
    __shared__ uint32_t scan[1024];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    const uint32_t count = *a.inCount;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t tile = 0; tile < count; tile += 1024) {
        const uint32_t i = tile + tid;
        const uint32_t f = (i < count) ? a.activeFlag[i] : 0u;
        scan[tid] = f;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
            uint32_t v = (tid >= off) ? scan[tid - off] : 0u;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const uint32_t base = carry;
        if (f) {
            const uint32_t idx = base + scan[tid] - 1u;
            a.out.o[idx] = a.staged.o[i];
            a.out.d[idx] = a.staged.d[i];
            a.out.thr[idx] = a.staged.thr[i];
            a.outShadow.o[idx] = a.stagedShadow.o[i];
            a.outShadow.d[idx] = a.stagedShadow.d[i];
            a.outShadow.c[idx] = a.stagedShadow.c[i];
        }
        __syncthreads();
        if (tid == 1023)
            carry = base + scan[1023];
        __syncthreads();
    }
    if (tid == 0) {
        *a.outCount = carry;
        *a.shadowCount = carry;
        *a.shadeHits = carry;
    }
}

// accumulate, accumulate.cl:6-34: mean -> exposure (exposure.cl:7-41) -> Reinhard (tonemapping.cl:7-10)
// -> sRGB (gamma.cl:4-13)
__global__ void __launch_bounds__(256) k_resolve(const float4* __restrict__ accum, float4* __restrict__ out, uint32_t n, float spp,
    float relativeAperture, float shutterTime, float ISO)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float EV100 = log2f(relativeAperture * relativeAperture / shutterTime * 100 / ISO);
    const float exposure = 1.0f / (1.2f * powf(2.0f, EV100));
    const float4 s = accum[i];
    float c[3] = { s.x / spp * exposure, s.y / spp * exposure, s.z / spp * exposure };
    for (int k = 0; k < 3; k++) {
        const float col = c[k] / (1.0f + c[k]);
        c[k] = (col <= 0.0031308f) ? col * 12.92f : (powf(fabsf(col), 1.0f / 2.4f) * 1.055f) - 0.055f;
    }
    out[i] = make_float4(c[0], c[1], c[2], 1.0f);
}

// fold the extra accumulator planes of a batch into the accumulator and clear them; only the pixels this
// context owns were written (pixels == nullptr: all of them), so a rank of an N-GPU job reads 1/N of the planes.
// Sixteen lanes share a pixel: its planes are adjacent in memory, so they read whole cache lines.
constexpr uint32_t kFoldLanes = 16;
// small launches (one sample in flight): the shadow rays deposit into an accumulator of their own (ptamd.hip, renderSampleFixed), added to the accumulator
// proper -- and cleared -- once per pt_render
__global__ void __launch_bounds__(256) k_merge_accum(float4* __restrict__ acc, float4* __restrict__ shadowAcc, uint32_t n, uint32_t planes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    // one plane per bounce, added in the order of the bounces whatever order the passes ran in (the serial and the overlapped schedule: the same sums)
    float4 a = acc[i];
    bool any = false;
    for (uint32_t p = 0; p < planes; p++) {
        const float4 b = shadowAcc[(size_t)p * n + i];
        if (b.x != 0.f || b.y != 0.f || b.z != 0.f) { // (adding zero changes nothing: planes no shadow ray of this pixel reached are not written)
            a.x += b.x, a.y += b.y, a.z += b.z;
            shadowAcc[(size_t)p * n + i] = make_float4(0.f, 0.f, 0.f, 0.f);
            any = true;
        }
    }
    if (any)
        acc[i] = a;
}

__global__ void __launch_bounds__(256) k_fold_planes(AccumView acc, uint32_t planes, const uint32_t* pixels, uint32_t numOwned)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = t / kFoldLanes, part = t % kFoldLanes;
    const bool live = k < numOwned; // numOwned * kFoldLanes need not fill the last wave; its lanes still take part in the shuffles
    const uint32_t i = live ? (pixels ? pixels[k] : k) : 0u;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const uint32_t slot = acc.ordinal ? k : i; // the ordinal of the k-th owned pixel is k
    if (live)
        for (uint32_t p = 1u + part; p < planes; p += kFoldLanes) {
            float4* e = acc.atOrdinal(p, slot);
            const float4 v = *e;
            sx += v.x, sy += v.y, sz += v.z;
            *e = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    for (int o = kFoldLanes / 2; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, kFoldLanes);
        sy += __shfl_xor(sy, o, kFoldLanes);
        sz += __shfl_xor(sz, o, kFoldLanes);
    }
    if (live && part == 0u) {
        float4 s = acc.plane0[i];
        s.x += sx, s.y += sy, s.z += sz;
        acc.plane0[i] = s;
    }
}

__global__ void k_set_word(uint32_t* p, uint32_t v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *p = v;
}

// end-of-sample bookkeeping: fold the per-pass counters into 64-bit totals and zero the control
// block for the next sample (replaces updateKernelData, kernel.cl:303-317, and the per-pass
// blocking KernelData read-back of raytracer.cpp:381-389).
// `countsOut` (may be null): pinned host memory that receives the pass counters -- next frame's launch sizes and kernel choices (a copy command on the
// stream was a blit kernel and two launch gaps on a 1-spp frame's critical path: ~20 us of ~750)
// Queues smaller than the batch (round 6): what pass `pass` emitted is cut at the queues' sizes -- the kernels behind read the counts, not the sizes -- and a cut is
// REPORTED (a sticky word in pinned host memory: pt_synchronize and every read of the image fail once it is set).  One thread.
__global__ void k_clamp_counts(Control* ctl, uint32_t pass, uint32_t outCap, uint32_t shadowCap, uint32_t* overflowPinned)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    bool cut = false;
    if (ctl->extCount[pass + 1] > outCap)
        ctl->extCount[pass + 1] = outCap, cut = true;
    if (ctl->shadowCount[pass] > shadowCap)
        ctl->shadowCount[pass] = shadowCap, cut = true;
    if (cut)
        *overflowPinned = 1u;
}

__global__ void k_end_sample(Control* ctl, Totals* tot, uint32_t passes, uint32_t* countsOut)
{
    // one wave: lane p folds and clears the counters of pass p, lane k the k-th deposit slot (17 passes, 64 slots: one round of loads instead of a
    // single thread's hundred dependent ones -- 13 us of a 1-spp frame's critical path were this kernel)
    static_assert(kMaxPasses < 64 && kDepositSlots <= 64u, "one lane per pass / per deposit slot");
    if (blockIdx.x != 0 || threadIdx.x >= 64u)
        return;
    const uint32_t lane = threadIdx.x;
    unsigned long long ext = 0, sh = 0, hits = 0, slots = 0;
    if (countsOut && lane <= (uint32_t)kMaxPasses) { // [0, kMaxPasses]: rays in the extension queue per pass; behind them the shadow rays per pass
        countsOut[lane] = lane <= passes ? ctl->extCount[lane] : 0u;
        countsOut[kMaxPasses + 1 + lane] = lane <= passes ? ctl->shadowCount[lane] : 0u;
    }
    if (lane <= passes && lane <= (uint32_t)kMaxPasses) {
        ext = ctl->extCount[lane];
        sh = ctl->shadowCount[lane];
        hits = ctl->shadeHits[lane];
        ctl->extCount[lane] = ctl->shadowCount[lane] = ctl->extCursor[lane] = ctl->shadowCursor[lane] = ctl->shadeHits[lane] = 0;
    }
    if (lane < kDepositSlots) {
        slots = ctl->depositSlots[lane][0];
        ctl->depositSlots[lane][0] = 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ext += __shfl_xor(ext, m);
        sh += __shfl_xor(sh, m);
        hits += __shfl_xor(hits, m);
        slots += __shfl_xor(slots, m);
    }
    if (lane == 0) {
        tot->raysExtension += ext;
        tot->raysShadow += sh;
        tot->shadeHits += hits;
        tot->raysGenerated += ctl->generated;
        tot->deposits += slots + ctl->depositsShade + ctl->depositsShadow;
        tot->depositsShadow += ctl->depositsShadow;
        ctl->generated = 0;
        ctl->depositsShade = ctl->depositsShadow = 0;
    }
}

} // namespace ptd
