"""ctypes binding of the HIP C-ABI (include/ptamd.h).  There is no CPU fallback: if libptamd.so
is missing or no HIP device is present, construction fails loudly."""
import ctypes as C
import os
import numpy as np
from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
DEVICE_LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(_HERE, "..", "csrc", "libptamd.so")  # PTAMD_LIB: tuning builds

RNG_COUNTER, RNG_LFSR113_PARITY = 0, 1
TEX_RGBA32F, TEX_BGRA8_UNORM = 0, 1
FLAG_ROWMAJOR_PIXELS = 1
FLAG_NO_BAKED_INSTANCES = 2  # every instance stays two-level
FLAG_TWO_LEVEL_ONLY = 4  # only single-leaf instances are copied to world space
FLAG_NO_PACKETS = 8  # primary rays through the per-ray kernel too
FLAG_PACKET_INTERSECT = 16  # the pt_intersect hook uses the packet kernel where the scene allows it
FLAG_INTEGRATOR_MIS = 32  # neeMisShading instead of neeIsShading
FLAG_COMPARE_SHADING = 64  # the reference's COMPARE_SHADING build: MIS on the left half of the image, IS on the right, same view
FLAG_SOLID_ANGLE_LIGHTS = 128  # NEE picks lights by weightedRandomPointOnLight
FLAG_MATERIAL_BINS = 512  # k_shade walks its tiles in material order (scenes with several material types; measured slower: opt-in)
FLAG_PARKED_INSTANCES = 4096  # entered instances always take the general (parked) route; default: translation + uniform scale is walked through entry nodes
FLAG_TEAM_INTERSECT = 8192  # the pt_intersect hook runs the team kernel (four lanes per ray) where the scene allows it
FLAG_QUEUE_PRIMARY_RAYS = 256  # k_gen writes the primary rays even where the packet kernel could regenerate them


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_active_rays", C.c_uint32),
                ("max_bounces", C.c_uint32), ("rng_mode", C.c_uint32), ("seed", C.c_uint32), ("device", C.c_int32),
                ("flags", C.c_uint32), ("samples_in_flight", C.c_uint32), ("ext_queue_fraction", C.c_float), ("shadow_queue_fraction", C.c_float)]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_extension", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_generated", C.c_uint64),
                ("shade_hits", C.c_uint64), ("deposits", C.c_uint64), ("samples", C.c_uint64),
                ("ms_last_render", C.c_double), ("ms_intersect", C.c_double), ("ms_shade", C.c_double),
                ("ms_shadow", C.c_double), ("ms_gen", C.c_double), ("packet_launches", C.c_uint64), ("ms_packet", C.c_double),
                ("deposits_shadow", C.c_uint64), ("gen_launches", C.c_uint64), ("bundle_launches", C.c_uint64),
                ("stack_need", C.c_uint32), ("folded_instances", C.c_uint32), ("team_launches", C.c_uint64),
                ("entered_instances", C.c_uint32), ("batch_samples", C.c_uint32), ("first_pass_ext_ratio", C.c_float), ("first_pass_shadow_ratio", C.c_float), ("probe_batches", C.c_uint32), ("general_route", C.c_uint32)]


class RaysSoA(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "tmax")]


class HitsSoA(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t", "u", "v", "prim", "inst")]


class ShadeBatchIO(C.Structure):
    _fields_ = ([("n", C.c_uint32)]
                + [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "thr_r", "thr_g", "thr_b", "pixel",
                                             "flags", "bounce", "t", "u", "v", "prim", "inst")]
                + [("sample", C.c_uint32)]
                + [(n, C.c_void_p) for n in ("radiance", "out_alive", "nox", "noy", "noz", "ndx", "ndy", "ndz",
                                             "nthr_r", "nthr_g", "nthr_b", "nflags", "shadow_alive", "sox", "soy",
                                             "soz", "sdx", "sdy", "sdz", "slen", "sc_r", "sc_g", "sc_b")])


EXPORTS = ["pt_create", "pt_destroy", "pt_last_error", "pt_set_stream", "pt_upload_static", "pt_upload_static_async", "pt_upload_dynamic",
           "pt_upload_dynamic_async", "pt_frame_tick", "pt_update_geometry", "pt_refit_vertices",
           "pt_upload_texture_array", "pt_set_camera", "pt_set_tiles", "pt_set_accum_buffer", "pt_clear", "pt_render",
           "pt_synchronize", "pt_resolve", "pt_resolve_device", "pt_resolve_device_ptr", "pt_read_accum", "pt_write_accum", "pt_accum_device_ptr",
           "pt_samples_per_pixel", "pt_stats_get", "pt_stats_reset", "pt_profile_kernels", "pt_reduce_accum",
           "pt_intersect", "pt_gen_rays", "pt_primary_pass", "pt_shade_batch", "pt_debug_quantise_node", "pt_debug_copy_bandwidth", "pt_version"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(DEVICE_LIB_PATH):
            raise RuntimeError(f"{DEVICE_LIB_PATH} missing: the HIP extension was not built "
                               "(run __graft_entry__.build()); there is no CPU fallback")
        try:
            # torch ships its own libamdhip64; two HIP runtimes in one process do not both see the GPU.
            # Importing torch first makes libptamd.so bind to the runtime torch has already loaded.
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(DEVICE_LIB_PATH)
        l.pt_last_error.restype = C.c_char_p
        l.pt_last_error.argtypes = [C.c_void_p]
        l.pt_version.restype = C.c_char_p
        l.pt_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        l.pt_destroy.argtypes = [C.c_void_p]
        l.pt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_upload_static.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                       C.c_uint32, C.c_void_p, C.c_uint32]
        l.pt_upload_static_async.argtypes = l.pt_upload_static.argtypes
        l.pt_upload_dynamic.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        l.pt_upload_dynamic_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        l.pt_frame_tick.argtypes = [C.c_void_p]
        l.pt_update_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        l.pt_refit_vertices.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        l.pt_upload_texture_array.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        l.pt_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_set_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        l.pt_set_accum_buffer.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_clear.argtypes = [C.c_void_p]
        l.pt_render.argtypes = [C.c_void_p, C.c_uint32]
        l.pt_synchronize.argtypes = [C.c_void_p]
        l.pt_resolve.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_resolve_device.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_resolve_device_ptr.restype = C.c_void_p
        l.pt_resolve_device_ptr.argtypes = [C.c_void_p]
        l.pt_read_accum.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_write_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        l.pt_accum_device_ptr.restype = C.c_void_p
        l.pt_accum_device_ptr.argtypes = [C.c_void_p]
        l.pt_samples_per_pixel.restype = C.c_uint32
        l.pt_samples_per_pixel.argtypes = [C.c_void_p]
        l.pt_stats_get.argtypes = [C.c_void_p, C.POINTER(Stats)]
        l.pt_stats_reset.argtypes = [C.c_void_p]
        l.pt_profile_kernels.argtypes = [C.c_void_p, C.c_int]
        l.pt_reduce_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        l.pt_intersect.argtypes = [C.c_void_p, C.POINTER(RaysSoA), C.c_uint32, C.c_int, C.POINTER(HitsSoA), C.c_uint32,
                                   C.POINTER(C.c_float)]
        l.pt_gen_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 7
        l.pt_primary_pass.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8
        l.pt_shade_batch.argtypes = [C.c_void_p, C.POINTER(ShadeBatchIO)]
        _lib = l
    return _lib


class PtError(RuntimeError):
    pass


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One render context on one GPU (mirrors what RayTracer owns, reference src/raytracer.h:54-106)."""

    def __init__(self, width, height, max_active_rays=0, max_bounces=0, rng_mode=RNG_COUNTER, seed=1, device=0,
                 flags=0, samples_in_flight=0, ext_queue_fraction=0.0, shadow_queue_fraction=0.0):
        self._h = C.c_void_p()
        self.width, self.height = width, height
        cfg = Config(width, height, max_active_rays, max_bounces, rng_mode, seed, device, flags, samples_in_flight, ext_queue_fraction, shadow_queue_fraction)
        rc = lib().pt_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise PtError(f"pt_create failed ({rc}): {lib().pt_last_error(None).decode()}")

    def _chk(self, rc, what):
        if rc != 0:
            raise PtError(f"{what} failed ({rc}): {lib().pt_last_error(self._h).decode()}")

    def close(self):
        if self._h:
            lib().pt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def upload_scene(self, flat, sky=None, material_textures=None):
        self._chk(lib().pt_upload_static(self._h, _p(flat.vertices), len(flat.vertices), _p(flat.triangles),
                                         len(flat.triangles), _p(flat.materials), len(flat.materials),
                                         _p(flat.sub_nodes), len(flat.sub_nodes)), "pt_upload_static")
        self.upload_dynamic(flat)
        if sky is not None:
            self.upload_texture(1, sky)
        if material_textures is not None:
            self.upload_texture(0, material_textures)

    def upload_static_async(self, flat):
        """A rebuilt scene, converted and copied beside the one that is rendering (the render stream is not synchronised); follow with
        upload_dynamic_async(flat) and frame_tick(), which adopts both."""
        self._chk(lib().pt_upload_static_async(self._h, _p(flat.vertices), len(flat.vertices), _p(flat.triangles), len(flat.triangles),
                                               _p(flat.materials), len(flat.materials), _p(flat.sub_nodes), len(flat.sub_nodes)), "pt_upload_static_async")

    def upload_dynamic(self, flat):
        self._chk(lib().pt_upload_dynamic(self._h, _p(flat.lights), len(flat.lights), _p(flat.top_nodes),
                                          len(flat.top_nodes), flat.top_root), "pt_upload_dynamic")

    def upload_dynamic_async(self, flat):
        """Convert the next dynamic state and start copying it into the inactive device buffers; frame_tick() adopts it."""
        self._chk(lib().pt_upload_dynamic_async(self._h, _p(flat.lights), len(flat.lights), _p(flat.top_nodes),
                                                len(flat.top_nodes), flat.top_root), "pt_upload_dynamic_async")

    def frame_tick(self):
        self._chk(lib().pt_frame_tick(self._h), "pt_frame_tick")

    def update_geometry(self, flat):
        """Refitted geometry (same topology): new vertices and sub-BVH boxes; follow with upload_dynamic(_async) / frame_tick."""
        self._chk(lib().pt_update_geometry(self._h, _p(flat.vertices), len(flat.vertices), _p(flat.sub_nodes), len(flat.sub_nodes)),
                  "pt_update_geometry")

    def refit_vertices(self, first_vertex, vertices):
        """A deformed mesh, refitted on the device: its vertex records (L.VERTEX) replace [first_vertex, ...) of the uploaded array; every box
        of the device's trees is recomputed there.  Follow with upload_dynamic(_async) / frame_tick."""
        v = np.ascontiguousarray(vertices, L.VERTEX)
        self._chk(lib().pt_refit_vertices(self._h, first_vertex, _p(v), len(v)), "pt_refit_vertices")

    def upload_texture(self, kind, arr):
        """[layers][h][w][4]: float32 r g b a (PT_TEX_RGBA32F) or uint8 b g r a (PT_TEX_BGRA8_UNORM, the reference's material
        array format); rows bottom-up."""
        arr = np.asarray(arr)
        fmt = TEX_BGRA8_UNORM if arr.dtype == np.uint8 else TEX_RGBA32F
        arr = np.ascontiguousarray(arr, np.uint8 if fmt == TEX_BGRA8_UNORM else np.float32)
        assert arr.ndim == 4 and arr.shape[3] == 4
        self._chk(lib().pt_upload_texture_array(self._h, kind, arr.shape[2], arr.shape[1], arr.shape[0], fmt, _p(arr)),
                  "pt_upload_texture_array")

    def set_camera(self, camera):
        cam = np.ascontiguousarray(np.asarray(camera, L.CAMERA).reshape(1))
        self._chk(lib().pt_set_camera(self._h, _p(cam)), "pt_set_camera")

    def set_tiles(self, rects):
        arr = (Rect * max(len(rects), 1))(*[Rect(*r) for r in rects])
        self._chk(lib().pt_set_tiles(self._h, arr if rects else None, len(rects)), "pt_set_tiles")

    def set_stream(self, stream_handle):
        """Enqueue everything on a caller-owned HIP stream.  A handle of 0 is the legacy default stream -- what
        torch.cuda.current_stream().cuda_stream is unless a torch.cuda.Stream is current -- which the C-ABI cannot tell from
        "no stream" (NULL = the context's own stream): refuse it instead of silently losing the ordering."""
        if not stream_handle:
            raise PtError("set_stream(0): the legacy default stream cannot be selected; make a torch.cuda.Stream current "
                          "and pass its handle (own_stream() returns to the context's own stream)")
        self._chk(lib().pt_set_stream(self._h, C.c_void_p(stream_handle)), "pt_set_stream")

    def own_stream(self):
        self._chk(lib().pt_set_stream(self._h, None), "pt_set_stream")

    def set_accum_buffer(self, device_ptr):
        self._chk(lib().pt_set_accum_buffer(self._h, C.c_void_p(device_ptr)), "pt_set_accum_buffer")

    # ---- render
    def clear(self):
        self._chk(lib().pt_clear(self._h), "pt_clear")

    def render(self, spp=1, sync=True):
        self._chk(lib().pt_render(self._h, spp), "pt_render")
        if sync:
            self.synchronize()

    def synchronize(self):
        self._chk(lib().pt_synchronize(self._h), "pt_synchronize")

    def read_accum(self):
        out = np.zeros((self.height * self.width, 4), np.float32)
        self._chk(lib().pt_read_accum(self._h, _p(out)), "pt_read_accum")
        return out

    def write_accum(self, accum, spp):
        a = np.ascontiguousarray(accum, np.float32)
        self._chk(lib().pt_write_accum(self._h, _p(a), spp), "pt_write_accum")

    def resolve(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._chk(lib().pt_resolve(self._h, _p(out)), "pt_resolve")
        return out

    def resolve_device(self, device_ptr=None):
        """accumulate kernel into device memory (None: a buffer owned by the context); asynchronous."""
        self._chk(lib().pt_resolve_device(self._h, C.c_void_p(device_ptr) if device_ptr else None), "pt_resolve_device")

    @property
    def resolve_device_ptr(self):
        """Device address of the image resolve_device(None) writes (0 before the first such call)."""
        return lib().pt_resolve_device_ptr(self._h) or 0

    @property
    def samples_per_pixel(self):
        return int(lib().pt_samples_per_pixel(self._h))

    @property
    def accum_device_ptr(self):
        return lib().pt_accum_device_ptr(self._h)

    def stats(self):
        s = Stats()
        self._chk(lib().pt_stats_get(self._h, C.byref(s)), "pt_stats_get")
        return {n: getattr(s, n) for n, _ in s._fields_}

    def copy_bandwidth(self, nbytes=1 << 30, repeat=5):
        """GB/s (read + written) of a float4 grid-stride device copy: the achievable-HBM yardstick of this box."""
        lib().pt_debug_copy_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_float)]
        g = C.c_float(0)
        self._chk(lib().pt_debug_copy_bandwidth(self._h, nbytes, repeat, C.byref(g)), "pt_debug_copy_bandwidth")
        return float(g.value)

    def reset_stats(self):
        self._chk(lib().pt_stats_reset(self._h), "pt_stats_reset")

    def profile_kernels(self, enable=True):
        self._chk(lib().pt_profile_kernels(self._h, int(enable)), "pt_profile_kernels")

    # ---- kernel-granular hooks
    def intersect(self, o, d, tmax=None, any_hit=False, repeat=1):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = len(o)
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        tm = np.ascontiguousarray(tmax, np.float32) if tmax is not None else np.full(n, np.inf, np.float32)
        rays = RaysSoA(*[_p(c).value for c in cols], _p(tm).value)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        prim, inst = np.zeros(n, np.int32), np.zeros(n, np.int32)
        hits = HitsSoA(_p(t).value, _p(u).value, _p(v).value, _p(prim).value, _p(inst).value)
        ms = C.c_float(0)
        self._chk(lib().pt_intersect(self._h, C.byref(rays), n, int(any_hit), C.byref(hits), repeat, C.byref(ms)),
                  "pt_intersect")
        return dict(t=t, u=u, v=v, prim=prim, inst=inst, ms=ms.value)

    def gen_rays(self, sample, n):
        arrs = [np.zeros(n, np.float32) for _ in range(6)]
        pixel = np.zeros(n, np.uint32)
        self._chk(lib().pt_gen_rays(self._h, sample, n, *[_p(a) for a in arrs], _p(pixel)), "pt_gen_rays")
        return np.stack(arrs[:3], 1), np.stack(arrs[3:], 1), pixel

    def primary_pass(self, sample, batch, n):
        """first pass of one batch as pt_render issues it: (o, d, pixel, hits) in queue order"""
        arrs = [np.zeros(n, np.float32) for _ in range(6)]
        pixel = np.zeros(n, np.uint32)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        prim, inst = np.zeros(n, np.int32), np.zeros(n, np.int32)
        hits = HitsSoA(_p(t).value, _p(u).value, _p(v).value, _p(prim).value, _p(inst).value)
        self._chk(lib().pt_primary_pass(self._h, sample, batch, n, *[_p(a) for a in arrs], _p(pixel), C.byref(hits)), "pt_primary_pass")
        return np.stack(arrs[:3], 1), np.stack(arrs[3:], 1), pixel, dict(t=t, u=u, v=v, prim=prim, inst=inst)

    def shade_batch(self, o, d, thr, pixel, flags, bounce, t, u, v, prim, inst, sample=0):
        n = len(o)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u32 = lambda a: np.ascontiguousarray(a, np.uint32)
        o, d, thr = f32(o), f32(d), f32(thr)
        ins = [f32(o[:, 0]), f32(o[:, 1]), f32(o[:, 2]), f32(d[:, 0]), f32(d[:, 1]), f32(d[:, 2]), f32(thr[:, 0]),
               f32(thr[:, 1]), f32(thr[:, 2]), u32(pixel), u32(flags), u32(bounce), f32(t), f32(u), f32(v),
               np.ascontiguousarray(prim, np.int32), np.ascontiguousarray(inst, np.int32)]
        out = dict(radiance=np.zeros((n, 3), np.float32), out_alive=np.zeros(n, np.uint32))
        for k in ("nox", "noy", "noz", "ndx", "ndy", "ndz", "nthr_r", "nthr_g", "nthr_b"):
            out[k] = np.zeros(n, np.float32)
        out["nflags"] = np.zeros(n, np.uint32)
        out["shadow_alive"] = np.zeros(n, np.uint32)
        for k in ("sox", "soy", "soz", "sdx", "sdy", "sdz", "slen", "sc_r", "sc_g", "sc_b"):
            out[k] = np.zeros(n, np.float32)
        order = ["radiance", "out_alive", "nox", "noy", "noz", "ndx", "ndy", "ndz", "nthr_r", "nthr_g", "nthr_b",
                 "nflags", "shadow_alive", "sox", "soy", "soz", "sdx", "sdy", "sdz", "slen", "sc_r", "sc_g", "sc_b"]
        io = ShadeBatchIO(n, *[_p(a).value for a in ins], sample, *[_p(out[k]).value for k in order])
        self._chk(lib().pt_shade_batch(self._h, C.byref(io)), "pt_shade_batch")
        return out
