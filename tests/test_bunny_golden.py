"""BASELINE configs 2 / 3 on the REAL Stanford bunny.  The reference's bun_zipper.ply does not travel to the GPU box; what travels is
tests/golden/bunny_flat.npz -- the typed arrays the host library flattens it to (binned-SAH tree / diffuse: config 2; SBVH / rough glass: config 3; in the
five-wall room) and what the oracle finds on them (tests/golden/make_bunny_golden.py, build container only).  CPU: the oracle reproduces the fixture from
the stored arrays (and, where /root/reference exists, the host library still flattens the PLY to the very same arrays).  GPU: the HIP path uploads the
arrays through the C-ABI and is held against the stored hits, occlusion verdicts and 256-spp image."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "opencl-path-tracer_amd"))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import gpu_util as U  # noqa: E402
import orclib as O  # noqa: E402
from ptamd import layout as L  # noqa: E402
from ptamd.host import FlatScene  # noqa: E402

W, Hh, SPP = 64, 36, 256


def _load():
    z = np.load(os.path.join(HERE, "golden", "bunny_flat.npz"))
    out = {}
    for k in z.files:
        if "@" in k:
            name, dt = k.split("@")
            out[name] = np.frombuffer(z[k].tobytes(), getattr(L, dt)).copy()
        else:
            out[k] = z[k]
    return out


def _flat(g, name):
    p = name + "_"
    return FlatScene(g[p + "vertices"], g[p + "triangles"], g[p + "materials"], g[p + "sub_nodes"], g[p + "lights"], g[p + "top_nodes"], int(g[p + "top_root"]),
                     int((g[p + "top_nodes"]["isLeaf"] != 0).sum()))


@pytest.mark.parametrize("name", ["binned", "sbvh"])
def test_the_oracle_reproduces_the_bunny_fixture(name):
    g = _load()
    flat, cam = _flat(g, name), g[name + "_camera"][0]
    assert int(g[name + "_stats"][0]) == 69451 and len(flat.vertices) == 35947 + 24  # the bunny + the room's six quads
    sc = O.BoundScene(flat)
    o, d, tmax = g[name + "_ray_o"], g[name + "_ray_d"], g[name + "_ray_tmax"]
    h = O.intersect_batch(sc, o, d, threads=4)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(h[k], g[f"{name}_hit_{k}"]), k
    assert (h["inst"] == 1).mean() > 0.3  # rays that meet the bunny (instance 1; instance 0 is the room)
    occ = O.intersect_batch(sc, o, d, tmax=tmax, any_hit=True, threads=4)["prim"]
    assert np.array_equal(occ.astype(np.uint8), g[name + "_occluded"])
    ref, cnt = O.render(sc, cam, W, Hh, 16, seed=6, threads=4)  # (the first 16 of the fixture's 256 samples: the sums are of the same paths)
    assert cnt["raysGenerated"] == W * Hh * 16 and np.isfinite(ref).all()
    assert abs(ref[:, :3].mean() / 16 - g[name + "_accum"].mean() / SPP) < 0.05 * g[name + "_accum"].mean() / SPP


@pytest.mark.skipif(not os.path.exists("/root/reference/assets/3dmodels/stanford/bunny/bun_zipper.ply"), reason="the reference's assets are only in the build container")
def test_the_host_library_still_flattens_the_ply_to_the_fixture():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_bunny_golden as M
    from ptamd import host as H
    g = _load()
    for name, builder, material in (("binned", H.BVH_BINNED_SAH, L.material_diffuse((0.8, 0.8, 0.8))), ("sbvh", H.BVH_SPATIAL_SPLIT, L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0))):
        b, _ = M.bunny_room(builder, material)
        for k in ("vertices", "triangles", "sub_nodes", "top_nodes"):
            assert getattr(b.flat, k).tobytes() == g[f"{name}_{k}"].tobytes(), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("flags_name", ["copied", "entered"])
@pytest.mark.parametrize("name", ["binned", "sbvh"])
def test_the_real_bunny_on_the_device(gpu, name, flags_name):
    g = _load()
    flat, cam = _flat(g, name), g[name + "_camera"][0]
    flags = gpu.FLAG_NO_BAKED_INSTANCES if flags_name == "entered" else 0
    ctx = U.make_ctx(gpu, flat, W, Hh, camera=cam, seed=6, samples_in_flight=64, flags=flags)
    o, d, tmax = g[name + "_ray_o"], g[name + "_ray_d"], g[name + "_ray_tmax"]
    want = {k: g[f"{name}_hit_{k}"] for k in ("t", "u", "v", "prim", "inst")}
    info = U.compare_hits(flat, ctx.intersect(o, d), want, edge_flip_frac=0.0 if flags_name == "entered" else 5e-4)
    assert info["n"] > 3000 and info["flips"] == 0
    occ = ctx.intersect(o, d, tmax=tmax, any_hit=True)["prim"]
    assert (occ.astype(np.uint8) != g[name + "_occluded"]).sum() <= 2 + info["edge_flips"]
    ctx.render(SPP)
    a, st = ctx.read_accum()[:, :3], ctx.stats()
    ctx.close()
    cnt = g[name + "_counts"]
    assert st["rays_generated"] == int(cnt[0]) == W * Hh * SPP
    for k, v in (("rays_extension", cnt[1]), ("rays_shadow", cnt[2]), ("shade_hits", cnt[3])):
        assert abs(st[k] - int(v)) <= 1e-3 * int(v) + 2, (k, st[k], int(v))
    U.image_margins(f"the real Stanford bunny, {name}, {flags_name}: 64x36, 256 spp path by path", a, g[name + "_accum"], SPP, cam, 1e-3, 1e-3)
