"""Properties of the oracle that the GPU tests rely on: the path-by-path production render equals the
queue loop with the counter PRNG, is deterministic and thread-count independent; traversal counters."""
import numpy as np

import orclib as O
from ptamd import scenes


def _scene():
    b = scenes.instanced_grid(40, 24, level=2, sky_size=(32, 16))
    return b, O.BoundScene(b.flat, sky=b.sky)


def test_pathwise_render_equals_queue_loop_counter_mode():
    b, sc = _scene()
    W, Hh = 40, 24
    st = O.QueueState(W, Hh, (W * Hh + 63) // 64 * 64)
    cnt = O.Counters()
    for s in range(4):
        O.trace_rays("oracle", sc, b.camera, st, None, params=O.Params(O.RNG_COUNTER, s, 7, 0), counters=cnt)
    acc, c2 = O.render(sc, b.camera, W, Hh, 4, seed=7, threads=1)
    assert np.array_equal(acc[:, :3], st.accum[:, :3])
    q = cnt.as_dict()
    for k in ("raysExtension", "raysShadow", "raysGenerated", "shadeHits"):
        assert q[k] == c2[k], k
    assert c2["raysGenerated"] == W * Hh * 4 and c2["raysExtension"] >= c2["raysGenerated"]


def test_render_deterministic_and_thread_independent():
    b, sc = _scene()
    a1, _ = O.render(sc, b.camera, 40, 24, 3, seed=1, threads=1)
    a4, _ = O.render(sc, b.camera, 40, 24, 3, seed=1, threads=4)
    assert np.array_equal(a1, a4)
    other, _ = O.render(sc, b.camera, 40, 24, 3, seed=2, threads=4)
    assert not np.array_equal(a1, other)
    # sample ranges compose: 2 + 1 samples == 3 samples
    part, _ = O.render(sc, b.camera, 40, 24, 2, seed=1, threads=2)
    part, _ = O.render(sc, b.camera, 40, 24, 1, seed=1, first_sample=2, threads=2, accum=part)
    assert np.allclose(part, a1, rtol=1e-6, atol=1e-6)


def test_pixel_subset_matches_full_render():
    b, sc = _scene()
    full, _ = O.render(sc, b.camera, 40, 24, 2, seed=1, threads=2)
    px = np.arange(0, 40 * 24, 3, dtype=np.uint32)
    sub, _ = O.render(sc, b.camera, 40, 24, 2, seed=1, pixels=px, threads=2)
    assert np.array_equal(sub[px], full[px]) and not sub[np.setdiff1d(np.arange(960), px)].any()


def test_counter_prng_is_uniform_and_keyed():
    lib = O.oracle()
    u = np.array([lib.orc_counter_u01(p, s, 1, 0, 5) for p in range(64) for s in range(64)])
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert lib.orc_counter_u01(3, 4, 1, 2, 5) != lib.orc_counter_u01(3, 4, 1, 3, 5) != lib.orc_counter_u01(4, 3, 1, 2, 5)


def test_traversal_counters():
    b, sc = _scene()
    rng = np.random.default_rng(0)
    o = rng.uniform(-3, 3, (2000, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 0.1
    d = rng.normal(size=(2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c = O.Counters()
    r = O.intersect_batch(sc, o, d, threads=2, counters=c)
    c = c.as_dict()
    assert c["raysExtension"] == 2000 and c["topVisits"] >= 2000 and c["innerSteps"] > 0 and c["triangleTests"] > 0
    assert (r["prim"] >= 0).any() and (r["prim"] < 0).any()
