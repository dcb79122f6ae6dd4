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
