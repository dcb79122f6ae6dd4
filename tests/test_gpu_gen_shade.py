"""gen_camera_rays and shade through the kernel-granular C-ABI hooks against the oracle (counter PRNG,
identical keying on both sides) -- every material type, per entry."""
import ctypes as C

import numpy as np
import pytest

import golden_io
import gpu_util as U
import orclib as O
from ptamd import layout as L, scenes

pytestmark = pytest.mark.gpu
F = L.SHADINGFLAGS_HASFINISHED


@pytest.mark.parametrize("thin", [False, True])
def test_gen_rays_match_oracle(gpu, thin):
    b = scenes.instanced_grid(48, 27, level=1, sky_size=(8, 4), thin_lens=thin)
    ctx = U.make_ctx(gpu, b, 48, 27, seed=9, flags=gpu.FLAG_ROWMAJOR_PIXELS)
    o, d, pixel = ctx.gen_rays(3, 48 * 27)
    assert np.array_equal(pixel, np.arange(48 * 27))
    sc = U.oracle_scene(b)
    kd = sc.kernel_data(b.camera, 48, 27)
    kd["maxRays"] = 48 * 27
    rays = np.zeros(48 * 27, L.RAY_DATA)
    O.oracle().orc_generatePrimaryRays(C.c_size_t(48 * 27), O._p(rays), O._p(kd), None, C.byref(O.Params(O.RNG_COUNTER, 3, 9, 0)))
    assert np.allclose(o, rays["origin"][:, :3], rtol=1e-6, atol=1e-7)
    assert np.allclose(d, rays["direction"][:, :3], rtol=1e-5, atol=1e-6)
    ctx.close()


def test_default_pixel_order_is_8x8_blocks(gpu):
    b = scenes.cornell_box(32, 16)
    ctx = U.make_ctx(gpu, b, 32, 16)
    _, _, pixel = ctx.gen_rays(0, 32 * 16)
    assert sorted(pixel.tolist()) == list(range(512))
    first = pixel[:64]
    assert set((first % 32).tolist()) == set(range(8)) and set((first // 32).tolist()) == set(range(8))
    ctx.close()


@pytest.mark.parametrize("name", ["mixed", "inst"])
@pytest.mark.parametrize("p", [0, 1])
def test_shade_batch_matches_oracle(gpu, golden, name, p):
    """Inputs: the golden shade passes (camera rays and first-bounce rays with their hit records from the
    reference kernels: all five material types, textured + alpha-0 texels, sky misses).  Both sides shade
    them with the counter PRNG."""
    flat, cam, sky, tex = golden_io.scene_inputs(golden, name)
    n_in = int(golden[f"shade_{name}_p{p}_count_in"])
    rays = golden[f"shade_{name}_p{p}_in_rays"][:n_in]
    live = (rays["flags"] & F) == 0
    idx = np.flatnonzero(live)[:700]
    rays = rays[idx]
    hit = golden[f"shade_{name}_p{p}_in_hit"][:n_in][idx]
    t, uv = golden[f"shade_{name}_p{p}_in_t"][:n_in][idx], golden[f"shade_{name}_p{p}_in_uv"][:n_in][idx]
    prim = np.where(hit, golden[f"shade_{name}_p{p}_in_prim"][:n_in][idx], -1)
    inst = golden[f"shade_{name}_p{p}_in_inst"][:n_in][idx]
    n = len(idx)
    ctx = U.make_ctx(gpu, flat, 64, 36, camera=cam, sky=sky, tex=tex, seed=4)
    got = ctx.shade_batch(rays["origin"][:, :3], rays["direction"][:, :3], rays["multiplier"][:, :3], rays["outputPixel"],
                          rays["flags"], rays["numBounces"], t, uv[:, 0], uv[:, 1], prim, inst, sample=2)
    # oracle: same entries as a queue of n slots
    sc = U.oracle_scene(flat, sky=sky, tex=tex)
    N = (n + 63) // 64 * 64
    qr = np.zeros(N, L.RAY_DATA)
    qr[:n] = rays
    sd = np.zeros(N, L.SHADING_DATA)
    sd["hit"][:n] = hit
    sd["t"][:n], sd["uv"][:n], sd["triangleIndex"][:n] = t, uv, np.maximum(prim, 0)
    base = flat.top_nodes.ctypes.data + L.TOP_BVH_NODE.fields["invTransform"][1]
    sd["invTransform"][:n] = np.where(hit, base + inst.astype(np.int64) * L.TOP_BVH_NODE.itemsize, 0).astype(np.uint64)
    kd = sc.kernel_data(cam, 64, 36)
    kd["numInRays"], kd["maxRays"] = n, N
    out_r, out_s = np.zeros(N, L.RAY_DATA), np.zeros(N, L.RAY_DATA)
    acc = np.zeros((64 * 36, 4), np.float32)
    O.oracle().orc_shade(C.c_size_t(N), O._p(acc), O._p(out_r), O._p(out_s), O._p(qr), O._p(sd), O._p(kd), C.byref(sc.struct), None,
                         C.byref(O.Params(O.RNG_COUNTER, 2, 4, 0)), None)
    # oracle enqueues every shaded hit (reference semantics); map back to input entries in order
    shaded = np.flatnonzero(hit)
    assert int(kd["numOutRays"]) == len(shaded)
    o_alive = np.zeros(n, bool)
    s_alive = np.zeros(n, bool)
    o_alive[shaded] = (out_r["flags"][:len(shaded)] & F) == 0
    s_alive[shaded] = (out_s["flags"][:len(shaded)] & F) == 0
    # a continuation/shadow decision can flip only when a random draw lands within fp32 round-off of a
    # threshold; allow 0.5 % of entries
    agree_o = got["out_alive"].astype(bool) == o_alive
    agree_s = got["shadow_alive"].astype(bool) == s_alive
    assert agree_o.mean() > 0.995 and agree_s.mean() > 0.995
    slot = np.full(n, -1)
    slot[shaded] = np.arange(len(shaded))
    both = o_alive & got["out_alive"].astype(bool)
    k = slot[both]
    for a, f, c in (("nox", "origin", 0), ("noy", "origin", 1), ("noz", "origin", 2), ("ndx", "direction", 0), ("ndy", "direction", 1),
                    ("ndz", "direction", 2), ("nthr_r", "multiplier", 0), ("nthr_g", "multiplier", 1), ("nthr_b", "multiplier", 2)):
        want = out_r[f][k, c]
        close = np.isclose(got[a][both], want, rtol=2e-3, atol=2e-4)
        assert close.mean() > 0.99, (a, close.mean())
    assert np.array_equal(got["nflags"][both] & 2, out_r["flags"][k] & 2)
    sb = s_alive & got["shadow_alive"].astype(bool)
    k = slot[sb]
    for a, f, c in (("sox", "origin", 0), ("sdx", "direction", 0), ("sdy", "direction", 1), ("sc_r", "multiplier", 0), ("sc_g", "multiplier", 1),
                    ("sc_b", "multiplier", 2)):
        close = np.isclose(got[a][sb], out_s[f][k, c], rtol=2e-3, atol=2e-4)
        assert close.mean() > 0.99, (a, close.mean())
    assert np.isclose(got["slen"][sb], out_s["rayLength"][k], rtol=1e-4, atol=1e-5).mean() > 0.99
    # radiance deposited by shade itself (emissive hits, sky misses): per entry on the GPU, per pixel in the oracle
    px = rays["outputPixel"].astype(np.int64)
    want_rad = np.zeros((64 * 36, 3), np.float64)
    np.add.at(want_rad, px, got["radiance"].astype(np.float64))
    assert np.allclose(want_rad, acc[:, :3], rtol=2e-3, atol=1e-4)
    ctx.close()
