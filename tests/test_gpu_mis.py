"""The reference's other integrator and light sampler on the GPU: neeMisShading (shading.cl:35-349), its
COMPARE_SHADING arrangement (kernel.cl:48-51,248-265; raytracer.cpp:464-495) and weightedRandomPointOnLight
(shading_helper.cl:216-259) -- HIP kernels against the oracle path by path, and the reference's own estimator-agreement check.
The oracle's restatement of these is pinned bit for bit to the reference's compiled code in tests/test_oracle_vs_ref.py."""
import numpy as np
import pytest

import gpu_util as U
import orclib as O
from ptamd import host as H, layout as L, scenes

pytestmark = pytest.mark.gpu

# fractions of pixels / queue entries within tolerance of the oracle as measured on the MI355X (profiles/round6/parity_margins.json): gpu_util.fraction_gate holds
# every such comparison against 0.98 x its entry here (and never below the round-number gate of rounds 1-5)
MEASURED = {
    'COMPARE_SHADING arrangement vs the oracle: pixels within 1e-3': 1.0000,
    'MIS integrator vs the oracle, cornell: pixels within 1e-3 of the oracle': 1.0000,
    'MIS integrator vs the oracle, glass: pixels within 1e-3 of the oracle': 1.0000,
    'MIS integrator vs the oracle, instanced: pixels within 1e-3 of the oracle': 0.9967,
    'MIS integrator vs the oracle, pbr: pixels within 1e-3 of the oracle': 1.0000,
    'MIS integrator vs the oracle, textured: pixels within 1e-3 of the oracle': 1.0000,
    'solid-angle light choice vs the oracle, integrator 0: pixels within 1e-3 of the oracle': 1.0000,
    'solid-angle light choice vs the oracle, integrator 1: pixels within 1e-3 of the oracle': 1.0000,
}
W, Hh = 96, 54

CASES = {
    "cornell": lambda: scenes.cornell_box(W, Hh),
    "pbr": lambda: scenes.cornell_box(W, Hh, box_materials=[L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), L.material_pbr_dielectric((0.8, 0.3, 0.2), 0.96)]),
    "glass": lambda: scenes.cornell_box(W, Hh, box_materials=[L.material_basic_refractive(1.5, (1, .6, .6), 5.0), L.material_refractive(0.9, 1.5, (.6, 1, .6), 5.0)]),
    "textured": lambda: scenes.blob_room(W, Hh, level=3, textured_floor=True),
    "instanced": lambda: scenes.instanced_grid(W, Hh, level=3, sky_size=(64, 32)),
}


def _gates(a, ref, st, cnt, spp, cam, close_frac=0.97, name=""):
    assert st["rays_generated"] == cnt["raysGenerated"] == W * Hh * spp
    for k, ck in (("rays_extension", "raysExtension"), ("rays_shadow", "raysShadow"), ("shade_hits", "shadeHits")):
        assert abs(st[k] - cnt[ck]) <= 1e-3 * cnt[ck] + 2, (k, st[k], cnt[ck])
    assert abs(a.mean() - ref.mean()) / ref.mean() < 1e-3
    close = np.isclose(a, ref, rtol=1e-3, atol=1e-3 * ref.max()).all(axis=1)
    U.fraction_gate(f"{name}: pixels within 1e-3 of the oracle", close, MEASURED, legacy=close_frac)
    assert U.rmse(U.tonemap(a, spp, cam), U.tonemap(ref, spp, cam)) < 5e-3


@pytest.mark.parametrize("case", sorted(CASES))
def test_mis_render_matches_oracle(gpu, case):
    """PT_FLAG_INTEGRATOR_MIS: same gates as the production integrator's test (test_gpu_render.py), same counter-PRNG keys on
    both sides -- MIS draws once more per PBR light sample and carries the sampling density of every bounce in the queue."""
    b, spp = CASES[case](), 32
    ctx = U.make_ctx(gpu, b, W, Hh, seed=3, samples_in_flight=1, flags=gpu.FLAG_INTEGRATOR_MIS)
    ctx.render(spp)
    a, st = ctx.read_accum()[:, :3], ctx.stats()
    ctx.close()
    ref, cnt = O.render(U.oracle_scene(b), b.camera, W, Hh, spp, seed=3, threads=8, integrator=O.INTEGRATOR_MIS)
    _gates(a, ref[:, :3], st, cnt, spp, b.camera, name=f"MIS integrator vs the oracle, {case}")
    plain, _ = O.render(U.oracle_scene(b), b.camera, W, Hh, spp, seed=3, threads=8)
    if case != "glass" or True:
        assert not np.array_equal(plain[:, :3], ref[:, :3]), "MIS and IS are different estimators"


def test_compare_shading_halves_agree_in_the_mean(gpu):
    """The reference's own check, on the GPU: PT_FLAG_COMPARE_SHADING renders the left half's view twice -- neeMisShading into
    the left half of the image, neeIsShading into the right -- and calculateAverageGrayscale (raytracer.cpp:464-495) compares
    the mean luminance of the halves.  Two estimators of one image: 64x36, 4 096 spp, means within 1 % (measured noise ~0.2 %)."""
    w, h, spp = 64, 36, 4096
    b = scenes.cornell_box(w, h)
    ctx = U.make_ctx(gpu, b, w, h, seed=5, flags=gpu.FLAG_COMPARE_SHADING)
    ctx.render(spp)
    a = ctx.read_accum()[:, :3] / spp
    ctx.close()
    lum = (a @ np.float32([0.2126, 0.7152, 0.0722])).reshape(h, w)
    left, right = float(lum[:, :w // 2].mean()), float(lum[:, w // 2:].mean())
    assert abs(left - right) / right < 0.01, (left, right)
    # same view in both halves: the images agree pixel for pixel up to noise, not just in the mean
    assert np.corrcoef(lum[:, :w // 2].ravel(), lum[:, w // 2:].ravel())[0, 1] > 0.99
    # and path by path against the oracle in the same arrangement
    ctx = U.make_ctx(gpu, b, w, h, seed=5, samples_in_flight=1, flags=gpu.FLAG_COMPARE_SHADING)
    ctx.render(32)
    got = ctx.read_accum()[:, :3]
    ctx.close()
    ref, _ = O.render(U.oracle_scene(b), b.camera, w, h, 32, seed=5, threads=8, integrator=O.INTEGRATOR_COMPARE)
    close = np.isclose(got, ref[:, :3], rtol=1e-3, atol=1e-3 * ref.max()).all(axis=1)
    U.fraction_gate("COMPARE_SHADING arrangement vs the oracle: pixels within 1e-3", close, MEASURED, legacy=0.97)
    assert abs(got.mean() - ref[:, :3].mean()) / ref[:, :3].mean() < 1e-3


def test_weighted_light_choice_matches_oracle(gpu):
    """PT_FLAG_SOLID_ANGLE_LIGHTS = weightedRandomPointOnLight as the reference wrote it (no kernel of the reference calls it):
    a room with five emissive quads of different size, distance and orientation, path by path against the oracle, for both
    integrators."""
    mats = scenes._room_materials()
    mb = scenes._MeshBuilder()
    scenes._room(mb, mats)
    rng = np.random.default_rng(3)
    for k in range(4):  # extra lights on the walls and the floor, facing inwards
        c, s = rng.uniform(-0.6, 0.6, 2), rng.uniform(0.05, 0.3)
        if k == 0:
            mb.add_quad((-0.999, 1 + c[0] - s, c[1] - s), (-0.999, 1 + c[0] + s, c[1] - s), (-0.999, 1 + c[0] + s, c[1] + s), (-0.999, 1 + c[0] - s, c[1] + s), 3)
        elif k == 1:
            mb.add_quad((0.999, 1 + c[0] - s, c[1] - s), (0.999, 1 + c[0] - s, c[1] + s), (0.999, 1 + c[0] + s, c[1] + s), (0.999, 1 + c[0] + s, c[1] - s), 3)
        elif k == 2:
            mb.add_quad((c[0] - s, 1 + c[1] - s, 0.999), (c[0] - s, 1 + c[1] + s, 0.999), (c[0] + s, 1 + c[1] + s, 0.999), (c[0] + s, 1 + c[1] - s, 0.999), 3)
        else:
            mb.add_quad((c[0] - s, 0.001, c[1] - s), (c[0] - s, 0.001, c[1] + s), (c[0] + s, 0.001, c[1] + s), (c[0] + s, 0.001, c[1] - s), 3)
    scene = H.Scene()
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    b = scenes.SceneBundle(scene, scenes.cornell_box(W, Hh).camera, W, Hh, name="five lights")
    assert len(b.flat.lights) == 10
    for flags, integrator in ((gpu.FLAG_SOLID_ANGLE_LIGHTS, O.INTEGRATOR_IS), (gpu.FLAG_SOLID_ANGLE_LIGHTS | gpu.FLAG_INTEGRATOR_MIS, O.INTEGRATOR_MIS)):
        ctx = U.make_ctx(gpu, b, W, Hh, seed=7, samples_in_flight=1, flags=flags)
        ctx.render(16)
        a, st = ctx.read_accum()[:, :3], ctx.stats()
        ctx.close()
        ref, cnt = O.render(U.oracle_scene(b), b.camera, W, Hh, 16, seed=7, threads=8, integrator=integrator, light_sampling=O.LIGHTS_SOLID_ANGLE)
        # the walk over the light weights ends on a float comparison per light: a few more round-off flips than elsewhere
        _gates(a, ref[:, :3], st, cnt, 16, b.camera, close_frac=0.95, name=f"solid-angle light choice vs the oracle, integrator {integrator}")


@pytest.mark.parametrize("flag", ["FLAG_INTEGRATOR_MIS", "FLAG_COMPARE_SHADING", "FLAG_SOLID_ANGLE_LIGHTS"])
def test_batches_through_the_bundle_kernel_shade_like_queued_camera_rays(gpu, flag):
    """64 samples in flight: the camera rays go through k_trace_multi, which queues (direction, pixel) only -- the general shading kernel (MIS,
    COMPARE_SHADING's remapped pixels, the weighted light sampler) then takes the eye as origin and the sample from the queue index (round 5).
    Same paths as with k_gen's full ray records and the per-ray kernel: the same ray counts, images equal but for the order of a pixel's sums."""
    b = scenes.instanced_grid(W, Hh, level=3, sky_size=(64, 32))
    fl = getattr(gpu, flag)
    imgs, stats = [], []
    for extra in (0, gpu.FLAG_QUEUE_PRIMARY_RAYS | gpu.FLAG_NO_PACKETS):
        ctx = U.make_ctx(gpu, b, W, Hh, seed=11, samples_in_flight=64, flags=fl | extra)
        ctx.render(128)
        imgs.append(ctx.read_accum()[:, :3].copy())
        stats.append(ctx.stats())
        ctx.close()
    assert stats[0]["bundle_launches"] > 0 and stats[1]["packet_launches"] == 0
    for k in ("rays_generated", "rays_extension", "rays_shadow", "shade_hits"):
        assert abs(stats[0][k] - stats[1][k]) <= 2e-4 * stats[1][k], (k, stats[0][k], stats[1][k])  # (exact-t ties between the two camera-ray kernels)
    close = np.isclose(imgs[0], imgs[1], rtol=1e-4, atol=1e-4 * imgs[1].max()).all(axis=1)
    assert close.mean() > 0.999, close.mean()
