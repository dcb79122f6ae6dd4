"""Live comparison of the oracle with the reference's own kernels (oracle/_ref), beyond the committed
fixtures: whole queue loops on every material / instancing / camera variant.  Runs where oracle/_ref
was built (this container: /root/reference present); bit-exact agreement is required."""
import numpy as np
import pytest

import orclib as O
from ptamd import host as H, layout as L, scenes

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (reference tree absent)")


def _both(bundle, W, Hh, spp, max_rays=None):
    N = max_rays or (W * Hh + 63) // 64 * 64
    sky = bundle.sky if bundle.sky is not None else np.full((1, 4, 8, 4), 0.7, np.float32)
    sc = O.BoundScene(bundle.flat, sky=sky, material_textures=bundle.material_textures)
    out = {}
    for which in ("ref", "oracle"):
        st = O.QueueState(W, Hh, N)
        streams = O.create_streams(W * Hh, use_ref=(which == "ref"))
        traces = []
        for _ in range(spp):
            tr, _ = O.trace_rays(which, sc, bundle.camera, st, streams)
            traces.append(tr)
        out[which] = (st.accum[:, :3].copy(), streams.copy(), np.concatenate(traces))
    return out


CASES = {
    "pbr": lambda: scenes.cornell_box(48, 27, box_materials=[L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), L.material_pbr_dielectric((0.2, 0.5, 0.8), 0.6)]),
    "smooth_pbr": lambda: scenes.cornell_box(48, 27, box_materials=[L.material_pbr_metal((0.9, 0.9, 0.9), 0.97), L.material_pbr_dielectric((0.8, 0.3, 0.2), 0.96)]),
    "glass": lambda: scenes.cornell_box(48, 27, box_materials=[L.material_basic_refractive(1.5, (1, 0.6, 0.6), 5.0), L.material_refractive(0.9, 1.5, (0.6, 1, 0.6), 5.0)]),
    "sbvh_glass_textured": lambda: scenes.blob_room(48, 27, material=L.material_refractive(0.9, 1.5, (1, 0.6, 0.6), 5.0), level=3, builder=H.BVH_SPATIAL_SPLIT, textured_floor=True),
    "instanced_sky": lambda: scenes.instanced_grid(48, 27, level=2, sky_size=(64, 32)),
    "instanced_thin_lens": lambda: scenes.instanced_grid(48, 27, level=2, sky_size=(64, 32), thin_lens=True),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_full_queue_loop_bit_exact(case):
    r = _both(CASES[case](), 48, 27, spp=6)
    assert np.array_equal(r["ref"][2], r["oracle"][2]), "per-pass queue counters differ"
    assert np.array_equal(r["ref"][1], r["oracle"][1]), "random streams diverged (different draw counts)"
    assert np.array_equal(r["ref"][0], r["oracle"][0]), "accumulators differ"
    assert np.isfinite(r["ref"][0]).all() and r["ref"][0].max() > 0


def test_refill_smaller_queue_than_image():
    r = _both(CASES["instanced_sky"](), 48, 27, spp=3, max_rays=192)
    assert np.array_equal(r["ref"][2], r["oracle"][2]) and len(r["ref"][2]) > 3 * 7
    assert np.array_equal(r["ref"][0], r["oracle"][0])


def test_stream_creation_matches_clrng_host_library():
    assert np.array_equal(O.create_streams(300, use_ref=True), O.create_streams(300))


# ---- the reference's MIS integrator (neeMisShading, shading.cl:35-349), reachable only in its COMPARE_SHADING build ------------
def _compare_shading(bundle, W, Hh, spp, integrator):
    N = (W * Hh + 63) // 64 * 64
    sky = bundle.sky if bundle.sky is not None else np.full((1, 4, 8, 4), 0.7, np.float32)
    sc = O.BoundScene(bundle.flat, sky=sky, material_textures=bundle.material_textures)
    out = {}
    for which in ("ref_compare_shading", "oracle"):
        st = O.QueueState(W, Hh, N)
        streams = O.create_streams(W * Hh, use_ref=(which != "oracle"))
        traces = []
        for _ in range(spp):
            tr, _ = O.trace_rays(which, sc, bundle.camera, st, streams, params=O.Params(O.RNG_LFSR113, 0, 0, 0, integrator, O.LIGHTS_UNIFORM))
            traces.append(tr)
        out[which] = (st.accum[:, :3].copy(), streams.copy(), np.concatenate(traces))
    return out["ref_compare_shading"], out["oracle"]


MIS_CASES = dict(CASES, cornell=lambda: scenes.cornell_box(48, 27), textured=lambda: scenes.blob_room(48, 27, level=3, textured_floor=True))


@pytest.mark.skipif(not O.have_ref_mis(), reason="oracle/_ref COMPARE_SHADING build missing")
@pytest.mark.parametrize("case", sorted(MIS_CASES))
def test_compare_shading_build_bit_exact(case):
    """kernel.cl built with -DCOMPARE_SHADING: neeMisShading on the left half of the image, neeIsShading on the right, both
    halves showing the left half's view (kernel.cl:48-51,248-265).  The oracle's restatement, with the reference's
    uninitialised read evaluating to 0 as it does in this build, reproduces whole queue loops bit for bit -- counters, stream
    states (= draw counts: MIS draws once more per PBR light sample) and sums of both halves."""
    (ra, rs, rt), (oa, os_, ot) = _compare_shading(MIS_CASES[case](), 48, 27, 5, O.INTEGRATOR_COMPARE_AS_COMPILED)
    assert np.array_equal(rt, ot), "per-pass queue counters differ"
    assert np.array_equal(rs, os_), "random streams diverged (different draw counts)"
    assert np.array_equal(ra, oa), "accumulators differ"
    left, right = ra.reshape(27, 48, 3)[:, :24], ra.reshape(27, 48, 3)[:, 24:]
    assert left.max() > 0 and right.max() > 0 and not np.array_equal(left, right), "the two halves run different integrators"


@pytest.mark.skipif(not O.have_ref_mis(), reason="oracle/_ref COMPARE_SHADING build missing")
def test_the_fixed_mis_density_only_changes_light_hits_after_diffuse_bounces():
    """What the fix changes (oracle.cpp neeShading, 'FIXED'): with the density of a diffuse continuation taken from the sampled
    direction instead of read as 0, the same queue loop keeps its counters and draw counts -- the fix touches no decision --
    and only gains radiance: lights found by a diffuse bounce now contribute with their MIS weight."""
    b = scenes.cornell_box(48, 27)
    (ra, rs, rt), (oa, os_, ot) = _compare_shading(b, 48, 27, 8, O.INTEGRATOR_COMPARE)
    assert np.array_equal(rt, ot) and np.array_equal(rs, os_)
    right_ref, right_orc = ra.reshape(27, 48, 3)[:, 24:], oa.reshape(27, 48, 3)[:, 24:]
    assert np.array_equal(right_ref, right_orc), "the IS half is untouched"
    left_ref, left_orc = ra.reshape(27, 48, 3)[:, :24], oa.reshape(27, 48, 3)[:, :24]
    assert (left_orc >= left_ref - 1e-6).all() and left_orc.sum() > left_ref.sum()


def test_mis_and_is_estimators_agree_in_the_mean():
    """The reference's own check (COMPARE_SHADING + calculateAverageGrayscale, raytracer.cpp:464-495): both integrators are
    estimators of the same image, so the mean luminance of the two halves must agree within Monte-Carlo noise.  With the
    uninitialised read fixed they do (diffuse Cornell box, 64x36 halves of the same view, 512 spp, counter PRNG: noise of the
    mean ~0.3 %); as compiled, MIS is darker by the share of light that diffuse bounces would have found."""
    W, Hh, spp = 64, 36, 512
    b = scenes.cornell_box(W, Hh)
    sc = O.BoundScene(b.flat)

    def halves(integrator):
        acc, _ = O.render(sc, b.camera, W, Hh, spp, seed=3, threads=8, integrator=integrator)
        lum = (acc[:, :3] / spp) @ np.float32([0.2126, 0.7152, 0.0722])  # the grayscale of raytracer.cpp:481
        img = lum.reshape(Hh, W)
        return float(img[:, :W // 2].mean()), float(img[:, W // 2:].mean())
    mis, nee = halves(O.INTEGRATOR_COMPARE)
    assert abs(mis - nee) / nee < 0.01, (mis, nee)
    mis_c, nee_c = halves(O.INTEGRATOR_COMPARE_AS_COMPILED)
    assert nee_c == nee and mis_c < mis, "as compiled, the MIS half loses the light found by diffuse bounces"


def test_weighted_light_sampling_matches_the_compiled_reference_function():
    """weightedRandomPointOnLight (shading_helper.cl:216-259) is called by no kernel; the reference's compiled function is
    driven directly (oracle/ref_build/ref_driver.cpp) on a scene with several lights at different distances and
    orientations: chosen point, normal, colour scale, area and the stream state after the call, bit for bit."""
    b = scenes.cornell_box(32, 18)
    flat = b.flat
    lights = np.zeros(5, L.EMISSIVE_TRIANGLE)
    rng = np.random.default_rng(2)
    for k in range(5):
        c = rng.uniform(-0.8, 0.8, 3) + np.array([0, 1.0, 0])
        lights[k]["vertices"][:, :3] = c + rng.normal(scale=0.25, size=(3, 3))
        lights[k]["material"] = L.material_emissive(rng.uniform(0.2, 1.0, 3), 10.0)
    flat = H.FlatScene(flat.vertices, flat.triangles, flat.materials, flat.sub_nodes, lights, flat.top_nodes, flat.top_root, 1)
    sc = O.BoundScene(flat)
    picks = set()
    for trial in range(200):
        x = rng.uniform(-0.9, 0.9, 3).astype(np.float32) + np.float32([0, 1, 0])
        s_ref = O.create_streams(trial + 1, use_ref=True)[trial:trial + 1].copy()
        s_orc = s_ref.copy()
        r = O.weighted_light("ref", sc, x, s_ref)
        o = O.weighted_light("oracle", sc, x, s_orc)
        for a, bb in zip(r, o):
            # (a back-facing light enters with a negative weight, so the walk can fall off the end of the list: both then read the
            # zeroed pad record and return NaN normals -- equal_nan)
            assert np.array_equal(np.asarray(a), np.asarray(bb), equal_nan=True), (trial, r, o)
        assert np.array_equal(s_ref, s_orc), "3 draws each"
        picks.add(round(r[3], 6))
    assert len(picks) >= 3, "several different lights get chosen"
