"""Generates tests/golden/golden_v1.npz by RUNNING THE REFERENCE'S OWN KERNELS (oracle/_ref: the
reference's OpenCL C compiled for the host CPU, see oracle/Makefile) on small seeded inputs.  Only
inputs and expected outputs are stored -- no reference source.  Run in the build container (needs
/root/reference for `make -C oracle ref`):   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C  # noqa: E402
import orclib as O  # noqa: E402
from ptamd import host as H, layout as L, scenes  # noqa: E402


def scene_mixed(width, height):
    """Cornell room, one PBR dielectric box, one rough-glass box."""
    mats = [L.material_pbr_dielectric((0.2, 0.5, 0.8), 0.6), L.material_refractive(0.9, 1.5, (0.6, 1.0, 0.6), 5.0)]
    return scenes.cornell_box(width, height, box_materials=mats)


def scene_plain(width, height):
    """Cornell room (diffuse) with a PBR metal and a PBR dielectric box: no refraction."""
    mats = [L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), L.material_pbr_dielectric((0.2, 0.5, 0.8), 0.6)]
    return scenes.cornell_box(width, height, box_materials=mats)


def scene_instances(width, height):
    """Textured (alpha cut-out) floor room + two scaled/translated instances of a low-poly blob (metal)
    + a basic-refractive blob; gradient sky."""
    mats = scenes._room_materials()
    mats.append(L.material_diffuse((0, 0, 0), texture_id=0))
    mb = scenes._MeshBuilder()
    scenes._room(mb, mats)
    mb.m[0] = mb.m[1] = len(mats) - 1
    sc = H.Scene()
    sc.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    metal = scenes.blob_mesh(L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), level=1, builder=H.BVH_SPATIAL_SPLIT)
    glass = scenes.blob_mesh(L.material_basic_refractive(1.5, (1.0, 0.6, 0.6), 5.0), level=1, seed=3, builder=H.BVH_BINNED_FAST)
    smooth = scenes.blob_mesh(L.material_pbr_dielectric((0.8, 0.3, 0.2), 0.96), level=1, seed=5, builder=H.BVH_BINNED_SAH)
    sc.add_node(metal, location=(-0.45, 0.5, 0.2), scale=(0.9, 0.9, 0.9))
    sc.add_node(metal, location=(0.5, 1.2, 0.4), scale=(0.5, 0.5, 0.5))
    sc.add_node(glass, location=(0.35, 0.45, -0.35), scale=(0.8, 0.8, 0.8))
    sc.add_node(smooth, location=(-0.3, 1.4, -0.2), scale=(0.45, 0.45, 0.45))
    cam = scenes._camera(width, height, (0.0, 1.0, -3.9), (0.0, 1.0, 0.0), 46.0)
    b = scenes.SceneBundle(sc, cam, width, height, name="instances")
    b.material_textures = scenes.checker_texture(16, holes=True)[None]
    yy = np.linspace(0, 1, 8, dtype=np.float32)[:, None, None]
    sky = np.ones((1, 8, 16, 4), np.float32)
    sky[0, :, :, :3] = 0.2 + 0.8 * yy * np.array([0.6, 0.8, 1.0], np.float32)
    b.sky = sky
    return b


def flat_arrays(prefix, b):
    f = b.flat
    out = {prefix + k: getattr(f, k) for k in ("vertices", "triangles", "materials", "sub_nodes", "lights", "top_nodes")}
    out[prefix + "top_root"] = np.uint32(f.top_root)
    out[prefix + "camera"] = b.camera
    out[prefix + "sky"] = b.sky if b.sky is not None else np.full((1, 2, 2, 4), 0.5, np.float32)
    out[prefix + "tex"] = b.material_textures if b.material_textures is not None else np.ones((1, 1, 1, 4), np.float32)
    return out


def bound(b):
    sky = b.sky if b.sky is not None else np.full((1, 2, 2, 4), 0.5, np.float32)
    return O.BoundScene(b.flat, sky=sky, material_textures=b.material_textures)


def main():
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref could not be built (needs /root/reference)"
    G = {}
    refk = O.ref_kernels()

    # ---- (1) LFSR113 known answers from the reference's clRNG host library -----------------------
    streams = O.create_streams(8, use_ref=True)
    G["lfsr_streams8"] = streams
    st = streams[:1].copy()
    rc = O.ref_clrng()
    G["lfsr_stream0_first16"] = np.array([rc.clrngLfsr113RandomU01_cl_float(st.ctypes.data_as(C.c_void_p)) for _ in range(16)], np.float32)

    # ---- (2) generatePrimaryRays: pinhole + thin lens at 16x9 ------------------------------------
    for name, thin in (("pinhole", False), ("thinlens", True)):
        b = scene_mixed(16, 9)
        cam = scenes._camera(16, 9, (0.0, 1.0, -3.9), (0.1, 0.9, 0.0), 50.0, thin_lens=thin, focal_length_mm=50.0, aperture_fstops=2.0)
        sc = bound(b)
        kd = sc.kernel_data(cam, 16, 9)
        kd["maxRays"] = 192
        rays = np.zeros(192, L.RAY_DATA)
        s = O.create_streams(16 * 9, use_ref=True)
        refk.ref_generatePrimaryRays(C.c_size_t(160), O._p(rays), O._p(kd), O._p(s))
        G[f"gen_{name}_camera"] = cam
        G[f"gen_{name}_origin"] = rays["origin"][:144, :3].copy()
        G[f"gen_{name}_direction"] = rays["direction"][:144, :3].copy()
        G[f"gen_{name}_pixel"] = rays["outputPixel"][:144].astype(np.uint32)
        G[f"gen_{name}_streams_after"] = s["current"].copy()
        G[f"gen_{name}_newRays"] = np.uint32(kd["newRays"])

    # ---- (3) intersectWalk / intersectShadows on two scenes -----------------------------------------
    for sname, b in (("mixed", scene_mixed(64, 36)), ("inst", scene_instances(64, 36))):
        G.update(flat_arrays(f"scene_{sname}_", b))
        sc = bound(b)
        rng = np.random.default_rng(11)
        n = 4096
        o = rng.uniform((-1, 0, -1), (1, 2, 1), (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        # edge cases: axis-parallel rays, origins on box faces / exactly 0, origin inside a box
        d[:64] = 0
        d[np.arange(64), np.arange(64) % 3] = np.where(np.arange(64) % 2 == 0, 1.0, -1.0)
        o[64:96, 1] = 0.0          # on the floor plane
        o[96:128, 0] = -1.0        # on the left wall
        o[128:160] = 0.0           # exactly the world origin components
        o[128:160, 1] = 1.0
        rays = np.zeros(n, L.RAY_DATA)
        rays["origin"][:, :3], rays["direction"][:, :3] = o, d
        kd = sc.kernel_data(b.camera, 64, 36)
        kd["numInRays"] = n
        shading = np.zeros(n, L.SHADING_DATA)
        stack = np.zeros(n * 32, np.uint32)
        refk.ref_intersectWalk(C.c_size_t(n), O._p(shading), O._p(rays), O._p(stack), O._p(kd), C.byref(sc.struct))
        hit = shading["hit"] != 0
        G[f"isect_{sname}_o"], G[f"isect_{sname}_d"] = o, d
        G[f"isect_{sname}_hit"] = hit
        G[f"isect_{sname}_t"] = np.where(hit, shading["t"], np.inf).astype(np.float32)
        G[f"isect_{sname}_uv"] = np.where(hit[:, None], shading["uv"], 0).astype(np.float32)
        G[f"isect_{sname}_prim"] = np.where(hit, shading["triangleIndex"], -1).astype(np.int32)
        G[f"isect_{sname}_inst"] = np.where(hit, sc.top_leaf_of_matrix(shading["invTransform"]), -1).astype(np.int32)
        # shadow segments: towards random targets, length = distance; half of them shortened to just before the first hit
        sh = np.zeros(n, L.RAY_DATA)
        sh["origin"][:, :3], sh["direction"][:, :3] = o, d
        length = rng.uniform(0.05, 3.0, n).astype(np.float32)
        tt = G[f"isect_{sname}_t"]
        just = hit & (np.arange(n) % 2 == 0)
        length[just] = tt[just] * np.where(np.arange(n)[just] % 4 == 0, np.float32(0.999), np.float32(1.001))
        sh["rayLength"] = length
        sh["multiplier"][:, :3] = 1.0
        sh["outputPixel"] = np.arange(n) % (64 * 36)
        kd["numOutRays"] = n
        acc = np.zeros((64 * 36, 4), np.float32)
        # one ray per launch slot may share a pixel: run them one by one to read the verdict per ray
        occl = np.zeros(n, np.uint8)
        for i in range(n):
            one = sh[i:i + 1].copy()
            one["outputPixel"] = 0
            a1 = np.zeros((1, 4), np.float32)
            kd1 = kd.copy()
            kd1["numOutRays"] = 1
            refk.ref_intersectShadows(C.c_size_t(1), O._p(a1), O._p(one), O._p(stack), O._p(kd1), C.byref(sc.struct))
            occl[i] = 0 if a1[0, 0] > 0 else 1
        G[f"shadow_{sname}_len"] = length
        G[f"shadow_{sname}_occluded"] = occl

    # ---- (4) shade: one pass over camera rays + a second pass over the survivors ------------------------
    for sname, b in (("mixed", scene_mixed(64, 36)), ("inst", scene_instances(64, 36))):
        sc = bound(b)
        W, Hh, N = 64, 36, 64 * 36
        st = O.QueueState(W, Hh, N)
        s = O.create_streams(W * Hh, use_ref=True)
        kd = sc.kernel_data(b.camera, W, Hh)
        kd["maxRays"] = N
        refk.ref_generatePrimaryRays(C.c_size_t(N), O._p(st.rays[0]), O._p(kd), O._p(s))
        cur, nxt = 0, 1
        for p in range(2):
            refk.ref_intersectWalk(C.c_size_t(N), O._p(st.shading), O._p(st.rays[cur]), O._p(st.stack), O._p(kd), C.byref(sc.struct))
            G[f"shade_{sname}_p{p}_in_rays"] = st.rays[cur].copy()
            G[f"shade_{sname}_p{p}_in_hit"] = (st.shading["hit"] != 0)
            G[f"shade_{sname}_p{p}_in_t"] = st.shading["t"].copy()
            G[f"shade_{sname}_p{p}_in_uv"] = st.shading["uv"].copy()
            G[f"shade_{sname}_p{p}_in_prim"] = st.shading["triangleIndex"].copy()
            G[f"shade_{sname}_p{p}_in_inst"] = np.where(st.shading["hit"] != 0, sc.top_leaf_of_matrix(st.shading["invTransform"]), -1).astype(np.int32)
            G[f"shade_{sname}_p{p}_streams_before"] = s["current"].copy()
            G[f"shade_{sname}_p{p}_count_in"] = np.uint32(kd["numInRays"] + kd["newRays"])
            shade_acc = np.zeros_like(st.accum)  # shade's own deposits (emissive / sky), exact
            refk.ref_shade(C.c_size_t(N), O._p(shade_acc), O._p(st.rays[nxt]), O._p(st.shadow), O._p(st.rays[cur]), O._p(st.shading),
                           O._p(kd), C.byref(sc.struct), O._p(s))
            n_out = int(kd["numOutRays"])
            G[f"shade_{sname}_p{p}_radiance"] = shade_acc[:, :3].copy()
            st.accum += shade_acc
            G[f"shade_{sname}_p{p}_out_rays"] = st.rays[nxt][:n_out].copy()
            G[f"shade_{sname}_p{p}_out_shadow"] = st.shadow[:n_out].copy()
            G[f"shade_{sname}_p{p}_streams_after"] = s["current"].copy()
            refk.ref_intersectShadows(C.c_size_t((n_out + 63) // 64 * 64), O._p(st.accum), O._p(st.shadow), O._p(st.stack), O._p(kd), C.byref(sc.struct))
            refk.ref_updateKernelData(O._p(kd))
            cur, nxt = nxt, cur

    # ---- (5) queue semantics with refill: 32x18, maxRays 256 ------------------------------------------
    b = scene_instances(32, 18)
    sc = bound(b)
    st = O.QueueState(32, 18, 256)
    s = O.create_streams(32 * 18, use_ref=True)
    trace, _ = O.trace_rays("ref", sc, b.camera, st, s)
    G["queue_trace_32x18_cap256"] = trace
    G["queue_accum_32x18_cap256"] = st.accum[:, :3].copy()
    G["queue_camera"] = b.camera

    # ---- (6) accumulated images (LFSR113, gid-ordered compaction): 256 spp at 64x36 ------------------
    for sname, b in (("mixed", scene_mixed(64, 36)), ("inst", scene_instances(64, 36))):
        sc = bound(b)
        st = O.QueueState(64, 36, 64 * 36)
        s = O.create_streams(64 * 36, use_ref=True)
        for spp in range(256):
            _, kd = O.trace_rays("ref", sc, b.camera, st, s)
            if spp + 1 in (16, 256):
                G[f"image_{sname}_accum_{spp + 1}spp"] = st.accum[:, :3].copy()
        # (7) accumulate kernel on the 256-spp sums
        G[f"image_{sname}_resolved_256spp"] = O.accumulate("ref", st.accum, kd, 64, 36, 256)

    # ---- (8) a scene without refractive materials, where fp32 round-off flips decisions so rarely that a
    # GPU render with the same streams tracks these sums pixel by pixel for the first ~100k paths
    b = scene_plain(64, 36)
    G.update(flat_arrays("scene_plain_", b))
    sc = bound(b)
    st = O.QueueState(64, 36, 64 * 36)
    s = O.create_streams(64 * 36, use_ref=True)
    for spp in range(32):
        O.trace_rays("ref", sc, b.camera, st, s)
        if spp + 1 in (4, 32):
            G[f"image_plain_accum_{spp + 1}spp"] = st.accum[:, :3].copy()

    # structured arrays (C-ABI records with overlapping union fields) are stored as raw bytes:
    # key "name@DTYPE" -> uint8[..., itemsize]; tests/golden_io.py turns them back into records
    names = {id(getattr(L, n)): n for n in dir(L) if isinstance(getattr(L, n), np.dtype)}
    packed = {}
    for k, v in G.items():
        v = np.asarray(v)
        if v.dtype.names:
            dt = [n for n in dir(L) if isinstance(getattr(L, n), np.dtype) and getattr(L, n) == v.dtype][0]
            a = np.ascontiguousarray(v).reshape(-1)
            packed[f"{k}@{dt}"] = np.frombuffer(a.tobytes(), np.uint8).reshape(len(a), v.dtype.itemsize)
        else:
            packed[k] = v
    out = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(out, **packed)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB,", len(G), "arrays")


if __name__ == "__main__":
    main()
