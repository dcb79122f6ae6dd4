"""numpy mirrors of the C-ABI structs in include/ptamd.h (= the reference's device structs,
SURVEY.md section 2.3) and of the reference-only queue records used by the oracle."""
import numpy as np

VERTEX = np.dtype([("vertex", "<f4", 4), ("normal", "<f4", 4), ("texCoord", "<f4", 2), ("_pad", "<f4", 2)])
TRIANGLE = np.dtype([("indices", "<u4", 3), ("materialIndex", "<u4")])
# 48-byte tagged union (assets/cl/material.cl:3-51); overlapping views by explicit offsets
MATERIAL = np.dtype({
    "names": ["colour", "textureId", "smoothness", "f0NonMetal", "metallic", "refractiveIndexRough",
              "refractiveIndexBasic", "type", "raw"],
    "formats": [("<f4", 4), "<i4", "<f4", "<f4", "u1", "<f4", "<f4", "<i4", ("u1", 48)],
    "offsets": [0, 16, 16, 20, 24, 20, 16, 32, 0],
    "itemsize": 48,
})
EMISSIVE_TRIANGLE = np.dtype([("vertices", "<f4", (3, 4)), ("material", MATERIAL)])
SUB_BVH_NODE = np.dtype([("min", "<f4", 4), ("max", "<f4", 4), ("left", "<u4"), ("count", "<u4"), ("_pad", "<u4", 2)])
TOP_BVH_NODE = np.dtype([("min", "<f4", 4), ("max", "<f4", 4), ("invTransform", "<f4", 16), ("a", "<u4"), ("b", "<u4"),
                         ("isLeaf", "<u4"), ("_pad", "<u4")])
CAMERA = np.dtype([("eyePoint", "<f4", 4), ("screenPoint", "<f4", 4), ("u", "<f4", 4), ("v", "<f4", 4),
                   ("uNormalized", "<f4", 4), ("vNormalized", "<f4", 4), ("focalDistance", "<f4"),
                   ("apertureRadius", "<f4"), ("relativeAperture", "<f4"), ("shutterTime", "<f4"), ("ISO", "<f4"),
                   ("thinLensEnabled", "u1"), ("_pad", "u1", 11)])

# ---- reference-only records (oracle side) ---------------------------------------------------
# RayData, assets/cl/shading.cl:16-29 (80 B)
RAY_DATA = np.dtype({
    "names": ["origin", "direction", "multiplier", "outputPixel", "flags", "numBounces", "rayLength", "pdf", "t"],
    "formats": [("<f4", 4), ("<f4", 4), ("<f4", 4), "<u8", "<i4", "<i4", "<f4", "<f4", "<f4"],
    "offsets": [0, 16, 32, 48, 56, 60, 60, 64, 68],
    "itemsize": 80,
})
# ShadingData, assets/cl/kernel_data.cl:26-33 (32 B)
SHADING_DATA = np.dtype({
    "names": ["uv", "invTransform", "triangleIndex", "t", "hit"],
    "formats": [("<f4", 2), "<u8", "<i4", "<f4", "u1"],
    "offsets": [0, 8, 16, 20, 24],
    "itemsize": 32,
})
# KernelData, assets/cl/kernel_data.cl:4-24 (176 B)
KERNEL_DATA = np.dtype({
    "names": ["camera", "numEmissiveTriangles", "topLevelBvhRoot", "rayOffset", "scrWidth", "scrHeight",
              "numInRays", "numOutRays", "numShadowRays", "maxRays", "newRays"],
    "formats": [CAMERA] + ["<u4"] * 10,
    "offsets": [0] + [128 + 4 * i for i in range(10)],
    "itemsize": 176,
})
# clrngLfsr113HostStream, third_party/clRNG/include/clRNG/lfsr113.clh:73-78 (48 B)
LFSR113_STREAM = np.dtype([("current", "<u4", 4), ("initial", "<u4", 4), ("substream", "<u4", 4)])

SHADINGFLAGS_HASFINISHED = 1
SHADINGFLAGS_LASTSPECULAR = 2
MAT_DIFFUSE, MAT_PBR, MAT_REFRACTIVE, MAT_BASIC_REFRACTIVE, MAT_EMISSIVE = range(5)

for _dt, _sz in ((VERTEX, 48), (TRIANGLE, 16), (MATERIAL, 48), (EMISSIVE_TRIANGLE, 96), (SUB_BVH_NODE, 48),
                 (TOP_BVH_NODE, 112), (CAMERA, 128), (RAY_DATA, 80), (SHADING_DATA, 32), (KERNEL_DATA, 176),
                 (LFSR113_STREAM, 48)):
    assert _dt.itemsize == _sz, (_dt, _dt.itemsize, _sz)


def material_diffuse(colour, texture_id=-1):
    m = np.zeros((), MATERIAL)
    m["type"] = MAT_DIFFUSE
    m["colour"][:3] = colour
    m["textureId"] = texture_id
    return m


def material_pbr_metal(reflectance, smoothness):
    m = np.zeros((), MATERIAL)
    m["type"] = MAT_PBR
    m["colour"][:3] = reflectance
    m["smoothness"] = smoothness
    m["metallic"] = 1
    return m


def material_pbr_dielectric(base_colour, smoothness, f0=0.04):
    m = np.zeros((), MATERIAL)
    m["type"] = MAT_PBR
    m["colour"][:3] = base_colour
    m["smoothness"] = smoothness
    m["f0NonMetal"] = f0
    m["metallic"] = 0
    return m


def material_refractive(smoothness, ior, colour=(1, 1, 1), absorption_factor=0.0):
    m = np.zeros((), MATERIAL)
    m["type"] = MAT_REFRACTIVE
    m["colour"][:3] = (1.0 - np.asarray(colour, np.float32)) * np.float32(absorption_factor)
    m["smoothness"] = smoothness
    m["refractiveIndexRough"] = ior
    return m


def material_basic_refractive(ior, colour=(1, 1, 1), absorption_factor=0.0):
    m = np.zeros((), MATERIAL)
    m["type"] = MAT_BASIC_REFRACTIVE
    m["colour"][:3] = (1.0 - np.asarray(colour, np.float32)) * np.float32(absorption_factor)
    m["refractiveIndexBasic"] = ior
    return m


def material_emissive(colour, intensity=500.0):
    m = np.zeros((), MATERIAL)
    m["type"] = MAT_EMISSIVE
    m["colour"][:3] = np.asarray(colour, np.float32) * np.float32(intensity)
    return m
