"""Queues smaller than a batch (pt_config.ext_queue_fraction / shadow_queue_fraction, round 6; the reference keeps one slot per entry in every queue,
src/raytracer.cpp:760-787): the second extension queue and the shadow queue only hold what a batch's first pass emits; the library measures that (a probe batch,
then the counters of every batch) and cuts batches to what fits.  The image never depends on it; a wrong guess is reported, not rendered."""
import numpy as np
import pytest

import gpu_util as U
from ptamd import scenes

pytestmark = pytest.mark.gpu


def _render(gpu, b, W, Hh, spp, in_flight, seed=9, **kw):
    ctx = U.make_ctx(gpu, b, W, Hh, seed=seed, samples_in_flight=in_flight, **kw)
    ctx.render(spp)
    a, st = ctx.read_accum()[:, :3].copy(), ctx.stats()
    ctx.close()
    return a, st


@pytest.mark.parametrize("scene", ["open_sky", "closed_room", "thin_lens"])
def test_smaller_queues_render_the_same_image(gpu, scene):
    """open_sky: config 4's kind of scene (a quarter of the paths go on, fewer than half spawn a shadow ray): the fractions fit, batches stay whole after the
    probe.  closed_room: nearly every path goes on -- the same fractions force SMALLER batches, nothing else.  thin_lens: the first queue keeps all its planes."""
    W, Hh, n, spp = 160, 90, 64, 16 + 3 * 64  # (the probe batch of 16, then whole batches: the statistics name the LAST batch)
    if scene == "closed_room":
        b = scenes.cornell_box(W, Hh)
    else:
        b = scenes.instanced_grid(W, Hh, level=3, sky_size=(32, 16), thin_lens=scene == "thin_lens")
    want, st0 = _render(gpu, b, W, Hh, spp, n)
    got, st1 = _render(gpu, b, W, Hh, spp, n, ext_queue_fraction=0.45, shadow_queue_fraction=0.6)
    assert st0["probe_batches"] == 0 and st0["first_pass_ext_ratio"] == 0
    assert st1["probe_batches"] == 1 and 0 < st1["first_pass_ext_ratio"] <= 1 and 0 < st1["first_pass_shadow_ratio"] <= 1
    for k in ("rays_generated", "rays_extension", "rays_shadow", "shade_hits", "deposits"):
        assert st0[k] == st1[k], (k, st0[k], st1[k])  # the same paths, whatever the batches
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * want.max())  # (planes are folded in another order: sums differ by round-off)
    if scene == "closed_room":
        assert st1["first_pass_ext_ratio"] > 0.6 and st1["batch_samples"] < n, st1  # what goes on does not fit 0.45 of a whole batch: smaller batches
    else:
        assert st1["first_pass_ext_ratio"] < 0.42 and st1["first_pass_shadow_ratio"] < 0.57 and st1["batch_samples"] == n, st1


def test_a_new_camera_or_scene_state_is_probed_again(gpu):
    W, Hh, n = 160, 90, 64
    b = scenes.instanced_grid(W, Hh, level=3, sky_size=(32, 16))
    ctx = U.make_ctx(gpu, b, W, Hh, seed=2, samples_in_flight=n, ext_queue_fraction=0.5, shadow_queue_fraction=0.6)
    ctx.render(2 * n)
    r_sky = ctx.stats()["first_pass_ext_ratio"]
    assert ctx.stats()["probe_batches"] == 1
    ctx.render(n)
    assert ctx.stats()["probe_batches"] == 1  # same epoch: the counters of the earlier batches serve
    down = scenes._camera(W, Hh, (0.0, 6.0, 0.1), (0.0, 0.0, 0.0), 40.0)  # straight down at the meshes and the ground
    ctx.set_camera(down)
    ctx.clear()
    ctx.render(2 * n)
    st = ctx.stats()
    assert st["probe_batches"] == 2 and abs(st["first_pass_ext_ratio"] - r_sky) > 0.1, (st, r_sky)  # another view, another ratio: measured again, not inherited
    want, _ = _render(gpu, scenes.SceneBundle(b.scene, down, W, Hh, sky=b.sky), W, Hh, 2 * n, n, seed=2)
    assert np.allclose(ctx.read_accum()[:, :3], want, rtol=2e-5, atol=2e-5 * want.max())
    ctx.upload_dynamic(b.flat)  # a frame tick (the same state again): a new epoch all the same
    ctx.render(n)
    assert ctx.stats()["probe_batches"] == 3
    ctx.close()


def test_a_batch_that_outgrows_its_queues_is_reported_not_rendered(gpu, monkeypatch):
    """The guard behind the guess: PTAMD_DEBUG_BATCH_SCALE makes the library cut its batches three times too large for the closed room; the rays beyond the queues'
    ends are dropped ON THE DEVICE (no write past the end), the batch reports it, pt_synchronize and the image reads fail until pt_clear."""
    W, Hh, n = 160, 90, 64
    b = scenes.cornell_box(W, Hh)
    monkeypatch.setenv("PTAMD_DEBUG_BATCH_SCALE", "3.0")
    ctx = U.make_ctx(gpu, b, W, Hh, seed=3, samples_in_flight=n, ext_queue_fraction=0.3, shadow_queue_fraction=0.3)
    ctx.render(4 * n, sync=False)
    with pytest.raises(gpu.PtError, match="more rays than its queues hold"):
        ctx.synchronize()
    with pytest.raises(gpu.PtError, match="more rays than its queues hold"):
        ctx.read_accum()
    with pytest.raises(gpu.PtError, match="more rays than its queues hold"):
        ctx.render(n)
    monkeypatch.delenv("PTAMD_DEBUG_BATCH_SCALE")
    ctx.clear()
    ctx.render(2 * n)  # batches that fit again: the context is as good as new
    want, _ = _render(gpu, b, W, Hh, 2 * n, n, seed=3)
    assert np.allclose(ctx.read_accum()[:, :3], want, rtol=2e-5, atol=2e-5 * want.max())
    ctx.close()


def test_few_samples_in_flight_and_short_renders_with_smaller_queues(gpu):
    """The corners of the batch sizing: 32 samples in flight (the fractions are floored at what the 16-sample probe batch may emit), renders of fewer than 16
    samples (no bundles: whole camera rays are queued -- they fit the floor), a render that ends in a short batch."""
    W, Hh = 160, 90
    b = scenes.instanced_grid(W, Hh, level=3, sky_size=(32, 16))
    for n, spps in ((32, (32, 64)), (64, (5, 16 + 64 + 7)), (16, (48,))):
        want_ctx = U.make_ctx(gpu, b, W, Hh, seed=5, samples_in_flight=n)
        ctx = U.make_ctx(gpu, b, W, Hh, seed=5, samples_in_flight=n, ext_queue_fraction=0.3, shadow_queue_fraction=0.62)
        for spp in spps:
            ctx.render(spp)
            want_ctx.render(spp)
        a, w = ctx.read_accum()[:, :3], want_ctx.read_accum()[:, :3]
        assert np.allclose(a, w, rtol=2e-5, atol=2e-5 * w.max()), n
        for k in ("rays_generated", "rays_extension", "rays_shadow"):
            assert ctx.stats()[k] == want_ctx.stats()[k], (n, k)
        ctx.close()
        want_ctx.close()
