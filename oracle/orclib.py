"""TEST INFRASTRUCTURE ONLY.  ctypes front-end to the two checkers:

  * liboracle.so        -- our CPU restatement (oracle/oracle.cpp)
  * _ref/libref_*.so    -- the reference's own OpenCL kernels compiled for the host (oracle/Makefile)

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package (opencl-path-tracer_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", "opencl-path-tracer_amd"))
from ptamd import layout as L  # noqa: E402  (layout definitions only: plain numpy dtypes)

ORACLE_SO = os.path.join(_HERE, "liboracle_san.so" if os.environ.get("PTAMD_SANITIZE", "") not in ("", "0") else "liboracle.so")  # _san: ASan + UBSan build (ptamd/build.py)
ORACLE_FAST_SO = os.path.join(_HERE, "liboracle_fast.so")
REF_KERNELS_SO = os.path.join(_HERE, "_ref", "libref_kernels.so")
REF_KERNELS_MIS_SO = os.path.join(_HERE, "_ref", "libref_kernels_mis.so")  # the same kernels built with -DCOMPARE_SHADING
REF_ACCUM_SO = os.path.join(_HERE, "_ref", "libref_accumulate.so")
REF_CLRNG_SO = os.path.join(_HERE, "_ref", "libref_clrng.so")

RNG_LFSR113, RNG_COUNTER = 0, 1


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("layers", C.c_int32), ("_pad", C.c_int32),
                ("rgba", C.c_void_p)]


class SceneStruct(C.Structure):  # OrcScene == RefScene
    _fields_ = [(n, C.c_void_p) for n in ("vertices", "triangles", "subBvh", "topBvh", "emissive", "materials",
                                          "materialTextures", "skydomeTextures")]


INTEGRATOR_IS, INTEGRATOR_MIS, INTEGRATOR_COMPARE = 0, 1, 2  # neeIsShading | neeMisShading | COMPARE_SHADING's half-and-half
INTEGRATOR_MIS_AS_COMPILED, INTEGRATOR_COMPARE_AS_COMPILED = 3, 4  # ... with the reference's uninitialised pdf read as 0 (oracle/_ref)
LIGHTS_UNIFORM, LIGHTS_SOLID_ANGLE = 0, 1  # randomPointOnLight | weightedRandomPointOnLight


class Params(C.Structure):
    _fields_ = [("rngMode", C.c_uint32), ("sample", C.c_uint32), ("seed", C.c_uint32), ("maxBounces", C.c_uint32),
                ("integrator", C.c_uint32), ("lightSampling", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("raysExtension", "raysShadow", "raysGenerated", "shadeHits", "deposits",
                                          "topVisits", "innerSteps", "triangleTests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(fast=False, ref=True):
    targets = ["liboracle.so"] + (["liboracle_fast.so"] if fast else []) + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def have_ref():
    return all(os.path.exists(p) for p in (REF_KERNELS_SO, REF_ACCUM_SO, REF_CLRNG_SO))


_cache = {}


def _load(path):
    if path not in _cache:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _cache[path] = C.CDLL(path)
    return _cache[path]


def oracle(fast=False):
    lib = _load(ORACLE_FAST_SO if fast else ORACLE_SO)
    lib.orc_lfsr113_u01.restype = C.c_float
    lib.orc_counter_u01.restype = C.c_float
    lib.orc_counter_u01.argtypes = [C.c_uint32] * 5
    return lib


def ref_kernels(compare_shading=False):
    """The reference's kernels for the host; compare_shading: its COMPARE_SHADING build (kernel.cl:6) -- neeMisShading for
    the left half of the image, neeIsShading for the right half, both halves showing the left half's view."""
    return _load(REF_KERNELS_MIS_SO if compare_shading else REF_KERNELS_SO)


def have_ref_mis():
    return have_ref() and os.path.exists(REF_KERNELS_MIS_SO)


def ref_accumulate():
    return _load(REF_ACCUM_SO)


def ref_clrng():
    lib = _load(REF_CLRNG_SO)
    lib.clrngLfsr113CreateStreams.restype = C.c_void_p
    lib.clrngLfsr113CreateStreams.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    lib.clrngLfsr113DestroyStreams.argtypes = [C.c_void_p]
    lib.clrngLfsr113RandomU01_cl_float.restype = C.c_float
    lib.clrngLfsr113RandomU01_cl_float.argtypes = [C.c_void_p]
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class BoundScene:
    """Keeps the numpy arrays alive and exposes the C scene struct both checkers accept."""

    def __init__(self, flat, sky=None, material_textures=None):
        self.flat = flat
        # one zeroed light past the end: RandomInteger(0, n-1) can return n
        self.lights = np.zeros(len(flat.lights) + 1, L.EMISSIVE_TRIANGLE)
        self.lights[:len(flat.lights)] = flat.lights
        self._imgs = []
        self.sky_img = self._image(sky if sky is not None else np.zeros((1, 1, 1, 4), np.float32))
        self.tex_img = self._image(material_textures if material_textures is not None
                                   else np.ones((1, 1, 1, 4), np.float32))
        self.struct = SceneStruct(_p(flat.vertices).value, _p(flat.triangles).value, _p(flat.sub_nodes).value,
                                  _p(flat.top_nodes).value, _p(self.lights).value, _p(flat.materials).value,
                                  C.addressof(self.tex_img), C.addressof(self.sky_img))

    def _image(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        assert arr.ndim == 4 and arr.shape[3] == 4, "textures are [layers][h][w][4] float32"
        self._imgs.append(arr)
        return Image(arr.shape[2], arr.shape[1], arr.shape[0], 0, _p(arr).value)

    def kernel_data(self, camera, width, height):
        kd = np.zeros((), L.KERNEL_DATA)
        kd["camera"] = camera
        kd["numEmissiveTriangles"] = len(self.flat.lights)
        kd["topLevelBvhRoot"] = self.flat.top_root
        kd["scrWidth"], kd["scrHeight"] = width, height
        return kd

    def top_leaf_of_matrix(self, ptrs):
        """Map ShadingData.invTransform pointers back to top-level node indices."""
        base = self.flat.top_nodes.ctypes.data + L.TOP_BVH_NODE.fields["invTransform"][1]
        idx = (np.asarray(ptrs, np.int64) - base) // L.TOP_BVH_NODE.itemsize
        return idx


def create_streams(count, use_ref=False):
    """clrngLfsr113CreateStreams(NULL, count): stream k+1 = jump-ahead of stream k."""
    out = np.zeros(count, L.LFSR113_STREAM)
    if use_ref:
        lib = ref_clrng()
        # the default stream creator is process-global state that every CreateStreams call advances;
        # rewind it so that each call starts at stream 0 like a fresh RayTracer (raytracer.cpp:739-751)
        lib.clrngLfsr113RewindStreamCreator(None)
        size, err = C.c_size_t(0), C.c_int(0)
        ptr = lib.clrngLfsr113CreateStreams(None, count, C.byref(size), C.byref(err))
        assert err.value == 0 and size.value == count * 48
        C.memmove(_p(out), ptr, size.value)
        lib.clrngLfsr113DestroyStreams(ptr)
    else:
        oracle().orc_lfsr113_create_streams(C.c_uint32(count), _p(out))
    return out


class QueueState:
    """Buffers of one RayTracer instance in the reference's layouts (raytracer.cpp:623-788)."""

    def __init__(self, width, height, max_rays):
        assert max_rays % 64 == 0, "queue capacity must be a multiple of the work-group size 64"
        self.width, self.height, self.max_rays = width, height, max_rays
        self.rays = [np.zeros(max_rays, L.RAY_DATA), np.zeros(max_rays, L.RAY_DATA)]
        self.shadow = np.zeros(max_rays, L.RAY_DATA)
        self.shading = np.zeros(max_rays, L.SHADING_DATA)
        self.stack = np.zeros(max_rays * 32, np.uint32)
        self.accum = np.zeros((height * width, 4), np.float32)


def trace_rays(which, scene, camera, state, streams, params=None, max_passes=256, counters=None):
    """One sample per pixel through the whole queue loop.  which: 'ref', 'ref_compare_shading' (the reference built with
    -DCOMPARE_SHADING) or 'oracle' (integrator chosen by params).
    Returns the per-pass trace [(numInRays, newRays, rayOffset, numOutRays)]."""
    kd = scene.kernel_data(camera, state.width, state.height)
    trace = np.zeros((max_passes, 4), np.uint32)
    if which in ("ref", "ref_compare_shading"):
        n = ref_kernels(which == "ref_compare_shading").ref_trace_rays(_p(kd), C.c_uint32(state.max_rays), _p(state.rays[0]), _p(state.rays[1]),
                                         _p(state.shadow), _p(state.shading), _p(state.stack), _p(streams),
                                         _p(state.accum), C.byref(scene.struct), _p(trace), max_passes)
    else:
        prm = params or Params(RNG_LFSR113, 0, 0, 0, INTEGRATOR_IS, LIGHTS_UNIFORM)
        n = oracle().orc_trace_rays(_p(kd), C.c_uint32(state.max_rays), _p(state.rays[0]), _p(state.rays[1]),
                                    _p(state.shadow), _p(state.shading), _p(streams), _p(state.accum),
                                    C.byref(scene.struct), C.byref(prm), _p(trace), max_passes,
                                    C.byref(counters) if counters is not None else None)
    return trace[:min(n, max_passes)].copy(), kd


def weighted_light(which, scene, x, stream):
    """weightedRandomPointOnLight (shading_helper.cl:216-259) at shading point x with one LFSR113 stream (advanced in place):
    the reference's compiled function ('ref') or the oracle's restatement.  Returns (point, normal, colour, area)."""
    x = np.ascontiguousarray(x, np.float32)
    p, n, c = (np.zeros(3, np.float32) for _ in range(3))
    area = C.c_float(0)
    fn = ref_kernels().ref_test_weighted_light if which == "ref" else oracle().orc_weighted_light
    fn(C.byref(scene.struct), C.c_uint32(len(scene.flat.lights)), _p(x), _p(stream), _p(p), _p(n), _p(c), C.byref(area))
    return p, n, c, float(area.value)


def intersect_batch(scene, o, d, tmax=None, any_hit=False, threads=1, fast=False, counters=None):
    """oracle traceRay over SoA rays. o, d: (n,3) float32. Returns dict(t,u,v,prim,inst)."""
    lib = oracle(fast)
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    n = len(o)
    cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
    tm = np.ascontiguousarray(tmax, np.float32) if tmax is not None else np.full(n, np.inf, np.float32)
    t, u, v = (np.zeros(n, np.float32) for _ in range(3))
    prim, inst = np.zeros(n, np.int32), np.zeros(n, np.int32)
    lib.orc_intersect_batch(C.byref(scene.struct), C.c_uint32(scene.flat.top_root), C.c_uint32(n),
                            *[_p(c) for c in cols], _p(tm), int(any_hit), _p(t), _p(u), _p(v), _p(prim), _p(inst),
                            int(threads), C.byref(counters) if counters is not None else None)
    return dict(t=t, u=u, v=v, prim=prim, inst=inst)


def render(scene, camera, width, height, spp, seed=1, first_sample=0, max_bounces=0, pixels=None, threads=1,
           fast=False, accum=None, integrator=INTEGRATOR_IS, light_sampling=LIGHTS_UNIFORM):
    """Production-mode (counter PRNG) path-by-path render on `threads` host threads."""
    lib = oracle(fast)
    kd = scene.kernel_data(camera, width, height)
    if accum is None:
        accum = np.zeros((width * height, 4), np.float32)
    cnt = Counters()
    px = None if pixels is None else np.ascontiguousarray(pixels, np.uint32)
    lib.orc_render_ex(_p(kd), C.byref(scene.struct), C.c_uint32(first_sample), C.c_uint32(spp), C.c_uint32(seed),
                      C.c_uint32(max_bounces), C.c_uint32(integrator), C.c_uint32(light_sampling), _p(px),
                      C.c_uint32(0 if px is None else len(px)), _p(accum), int(threads), C.byref(cnt))
    return accum, cnt.as_dict()


def accumulate(which, accum, kd, width, height, n):
    out = np.zeros((height, width, 4), np.float32)
    if which == "ref":
        img = Image(width, height, 1, 0, _p(out).value)
        ref_accumulate().ref_accumulate(C.c_uint32(width), C.c_uint32(height), C.byref(img), _p(accum), _p(kd),
                                        C.c_uint32(n))
    else:
        oracle().orc_accumulate(C.c_uint32(width), C.c_uint32(height), _p(out), _p(accum), _p(kd), C.c_uint32(n))
    return out
