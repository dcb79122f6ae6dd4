"""BASELINE.json's full-size configuration (1080p, ~1M instanced triangles) through size-independent
properties: ray-count conservation, determinism, sample additivity, tile-sum == frame, sampled-pixel
agreement with the oracle."""
import numpy as np
import pytest

import gpu_util as U
import orclib as O
from ptamd import scenes

pytestmark = pytest.mark.gpu

# fractions of pixels / queue entries within tolerance of the oracle as measured on the MI355X (profiles/round6/parity_margins.json): gpu_util.fraction_gate holds
# every such comparison against 0.98 x its entry here (and never below the round-number gate of rounds 1-5)
MEASURED = {
    'blob_room config2 diffuse, binned SAH, first 8 spp: sampled pixels within 1e-3 of the oracle': 1.0000,
    'blob_room config3 rough glass, SBVH, first 8 spp: sampled pixels within 1e-3 of the oracle': 1.0000,
    'config4 1080p, 8 spp: sampled pixels within 1e-3 of the oracle': 0.9993,
    'config4 as a 5x3 grid, 8 spp, copied: sampled pixels within 1e-3 of the oracle': 0.9992,
    'config4 as a 5x3 grid, 8 spp, entered: sampled pixels within 1e-3 of the oracle': 0.9990,
    'config4 as timed, copied: sampled pixels within 2e-3 of the oracle': 0.9998,
    'config4 as timed, entered: sampled pixels within 2e-3 of the oracle': 0.9998,
    'config4 as timed, thin_lens: sampled pixels within 2e-3 of the oracle': 1.0000,
    'config5 4K thin lens, 2 spp: sampled pixels within 1e-3 of the oracle': 1.0000,
}
W, H = 1920, 1080


@pytest.fixture(scope="module")
def bundle():
    return scenes.instanced_grid(W, H, level=6)


def test_full_size_properties(gpu, bundle):
    assert 0.95e6 < bundle.flat.instanced_triangles < 1.3e6
    ctx = U.make_ctx(gpu, bundle, W, H, seed=1)
    ctx.render(4)
    a = ctx.read_accum()[:, :3]
    st = ctx.stats()
    assert st["rays_generated"] == W * H * 4
    assert st["rays_generated"] <= st["rays_extension"] <= 4 * st["rays_generated"]  # <= 4 bounces
    assert st["shade_hits"] <= st["rays_extension"] and st["rays_shadow"] <= st["shade_hits"]
    assert np.isfinite(a).all() and a.min() >= 0 and a.mean() > 0
    # determinism + additivity: 4 more samples on top == 8 samples in one go (up to summation order)
    ctx.render(4)
    a8 = ctx.read_accum()[:, :3]
    ctx.close()
    ctx2 = U.make_ctx(gpu, bundle, W, H, seed=1)
    ctx2.render(8)
    b8 = ctx2.read_accum()[:, :3]
    assert abs(ctx2.stats()["rays_extension"] - 2 * st["rays_extension"]) < 0.01 * st["rays_extension"]
    assert np.allclose(a8, b8, rtol=1e-5, atol=1e-5 * b8.max())
    ctx2.close()
    # sampled pixels against the oracle (path by path, same random numbers)
    px = np.random.default_rng(0).choice(W * H, 6000, replace=False).astype(np.uint32)
    ref, _ = O.render(U.oracle_scene(bundle), bundle.camera, W, H, 8, seed=1, pixels=px, threads=8)
    got, want = b8[px], ref[px, :3]
    U.image_margins("config4 1080p, 8 spp, 6000 pixels", got, want, 8, bundle.camera, 1e-3, 1e-3)
    close = np.isclose(got, want, rtol=1e-3, atol=1e-3 * want.max()).all(axis=1)
    U.fraction_gate("config4 1080p, 8 spp: sampled pixels within 1e-3 of the oracle", close, MEASURED, legacy=0.97)


def test_config4_as_a_5x3_grid_against_the_oracle(gpu):
    """SURVEY 8(d) specifies config 4 as 5 x 3 = 15 instances; bench.py's headline keeps round 1's 4 x 3 (985 012 instanced triangles with the substitute
    meshes) and reports the 5 x 3 grid (1 231 264: over, not under, 1 M) as the `grid_5x3` object.  Its oracle gate: 8 spp at 1080p, 6 000 sampled pixels path
    by path, instances copied and entered (the per-ray kernels then walk all 17 instances without parking)."""
    b = scenes.instanced_grid(W, H, nx=5, nz=3, level=6)
    assert b.flat.instanced_triangles > 1.2e6 and b.flat.num_instances == 17
    px = np.random.default_rng(5).choice(W * H, 6000, replace=False).astype(np.uint32)
    ref, _ = O.render(U.oracle_scene(b), b.camera, W, H, 8, seed=1, pixels=px, threads=16)
    for name, flags, folded in (("copied", 0, 0), ("entered", gpu.FLAG_NO_BAKED_INSTANCES, 17)):
        ctx = U.make_ctx(gpu, b, W, H, seed=1, flags=flags)
        ctx.render(8)
        st = ctx.stats()
        assert st["rays_generated"] == W * H * 8 and st["folded_instances"] == folded
        got, want = ctx.read_accum()[:, :3][px], ref[px, :3]
        ctx.close()
        U.image_margins(f"config4 as a 5x3 grid, 8 spp, {name}", got, want, 8, b.camera, 1e-3, 1e-3)
        close = np.isclose(got, want, rtol=1e-3, atol=1e-3 * want.max()).all(axis=1)
        U.fraction_gate(f"config4 as a 5x3 grid, 8 spp, {name}: sampled pixels within 1e-3 of the oracle", close, MEASURED, legacy=0.97)


@pytest.mark.parametrize("flags_name", ["copied", "entered", "thin_lens"])
def test_the_timed_configuration_against_the_oracle_directly(gpu, bundle, flags_name):
    """What bench.py times -- config 4 with bench.IN_FLIGHT (512 since round 4: ~200 GB of queues and planes) samples in flight: ONE batch whose primary
    rays are generated and traced by the bundle kernel (k_trace_multi: beam test, four rays per lane, no k_gen launch), 1.06 G queue entries -- held against
    the oracle directly: 4 096 sampled pixels at the full sample count, at north_star's gate (mean bias < 1e-3, tone-mapped RMSE < 1e-3; measured 2e-5 /
    2e-6 at 256 in flight, profiles/round4/parity_margins.json).  `entered`: the same with every instance entered at traversal
    (PT_FLAG_NO_BAKED_INSTANCES), the `two_level` object of the bench line; `thin_lens`: config 5's camera (f/2 focused on the grid centre) -- the packets
    are converging bundles walked around their waist on the focal plane."""
    import bench
    import torch
    # bench.IN_FLIGHT where the card has the ~200 GB free that it wants; otherwise the largest multiple of 256 that fits, as bench.py itself falls
    # back (the note goes on the record with the margins: a box with another tenant must not turn a PARITY test red)
    free_b, _ = torch.cuda.mem_get_info(0)
    n = bench.fit_in_flight(bench.IN_FLIGHT, W * H, free_b)
    assert n >= 256, f"{free_b / 1e9:.0f} GB free: not even 256 samples in flight fit"
    flags = gpu.FLAG_NO_BAKED_INSTANCES if flags_name == "entered" else 0
    if flags_name == "thin_lens":
        bundle = scenes.instanced_grid(W, H, level=6, thin_lens=True)
        assert bundle.camera["thinLensEnabled"]
    ctx = U.make_ctx(gpu, bundle, W, H, seed=1, samples_in_flight=n, flags=flags)
    ctx.render(n)
    st = ctx.stats()
    assert st["packet_launches"] == 1 and st["gen_launches"] == 0, "the batch must take the path the benchmark times"
    assert st["bundle_launches"] == 1  # bundles of 4 x 64 (k_trace_multi): around the eye of a pinhole, around the waist of a thin lens's converging bundle (round 6)
    assert st["rays_generated"] == W * H * n and ctx.samples_per_pixel == n
    a = ctx.read_accum()[:, :3]
    ctx.close()
    px = np.random.default_rng(4).choice(W * H, 4096, replace=False).astype(np.uint32)
    ref, _ = O.render(U.oracle_scene(bundle), bundle.camera, W, H, n, seed=1, pixels=px, threads=16)
    got, want = a[px], ref[px, :3]
    U.image_margins(f"config4 as timed ({n} in flight), {n} spp, {flags_name}", got, want, n, bundle.camera, 1e-3, 1e-3,
                    in_flight_wanted=bench.IN_FLIGHT, device_free_gb=round(free_b / 1e9, 1),
                    note="" if n == bench.IN_FLIGHT else f"fell back from {bench.IN_FLIGHT} to {n} samples in flight: {free_b / 1e9:.0f} GB of device memory free")
    # path by path most pixels agree to round-off (a pixel holds hundreds of paths here; one fp32 decision flip per pixel is common)
    close = np.isclose(got, want, rtol=2e-3, atol=2e-3 * want.max()).all(axis=1)
    U.fraction_gate(f"config4 as timed, {flags_name}: sampled pixels within 2e-3 of the oracle", close, MEASURED, legacy=0.85)


def test_config5_4k_thin_lens_properties(gpu):
    """BASELINE.json config 5 (same scene at 3840x2160, thin lens f/2 focused on the grid centre): ray-count
    conservation, tile sharding over 8 ranks' worth of tile lists == the whole frame, sampled pixels against the
    oracle path by path (the thin lens leaves primary directions un-normalised)."""
    W4, H4 = 3840, 2160
    b = scenes.instanced_grid(W4, H4, level=6, thin_lens=True)
    assert b.camera["thinLensEnabled"]
    ctx = U.make_ctx(gpu, b, W4, H4, seed=3)
    ctx.render(2)
    st = ctx.stats()
    assert st["rays_generated"] == W4 * H4 * 2
    assert st["rays_generated"] <= st["rays_extension"] <= 4 * st["rays_generated"]
    whole = ctx.read_accum()[:, :3]
    assert np.isfinite(whole).all() and whole.min() >= 0 and whole.mean() > 0
    ctx.close()
    # two of eight interleaved tile sets, each on its own context, land exactly on the whole-frame values
    import bench
    for rank in (0, 5):
        part = U.make_ctx(gpu, b, W4, H4, seed=3)
        rects = bench.tile_rects(W4, H4, rank, 8)
        part.set_tiles(rects)
        part.render(2)
        a = part.read_accum()[:, :3].reshape(H4, W4, 3)
        mask = np.zeros((H4, W4), bool)
        for x0, y0, x1, y1 in rects:
            mask[y0:y1, x0:x1] = True
        assert np.array_equal(a[mask], whole.reshape(H4, W4, 3)[mask]) and not a[~mask].any()
        part.close()
    px = np.random.default_rng(1).choice(W4 * H4, 4000, replace=False).astype(np.uint32)
    ref, _ = O.render(U.oracle_scene(b), b.camera, W4, H4, 2, seed=3, pixels=px, threads=8)
    got, want = whole[px], ref[px, :3]
    U.image_margins("config5 4K thin lens, 2 spp path by path", got, want, 2, b.camera, 1e-3, 1e-3)
    close = np.isclose(got, want, rtol=1e-3, atol=1e-3 * want.max()).all(axis=1)
    U.fraction_gate("config5 4K thin lens, 2 spp: sampled pixels within 1e-3 of the oracle", close, MEASURED, legacy=0.97)
    # the configuration's statistics at a sample count that means something: 64 spp in ONE batch (64 in flight at 4K = the 531 M queue
    # entries of the 1080p benchmark), 4 000 sampled pixels against the oracle
    ctx = U.make_ctx(gpu, b, W4, H4, seed=3, samples_in_flight=64)
    ctx.render(64)
    a64 = ctx.read_accum()[:, :3]
    assert ctx.stats()["rays_generated"] == W4 * H4 * 64
    ctx.close()
    ref64, _ = O.render(U.oracle_scene(b), b.camera, W4, H4, 64, seed=3, pixels=px, threads=16)
    U.image_margins("config5 4K thin lens, 64 spp", a64[px], ref64[px, :3], 64, b.camera, 1e-3, 1e-3)


def _config_room_test(gpu, bundle, spp_total, seed, n_pixels=4096, first=8, label=""):
    """A BASELINE room configuration at its full workload (1080p, its sample count): (1) the first `first` samples
    path by path against the oracle on `n_pixels` sampled pixels at the production gates; (2) the full sample count:
    ray-count conservation, additivity, determinism (a second context lands on the same bits) and the sampled pixels
    against the oracle's full-spp values (every pixel now holds hundreds of paths, so one fp32 round-off flip per pixel
    is the norm: the gates there are the image statistics of SURVEY 8(d) -- mean bias and tone-mapped RMSE)."""
    sc = U.oracle_scene(bundle)
    px = np.random.default_rng(seed).choice(W * H, n_pixels, replace=False).astype(np.uint32)
    ctx = U.make_ctx(gpu, bundle, W, H, seed=seed)
    ctx.render(first)
    a = ctx.read_accum()[:, :3]
    st = ctx.stats()
    assert st["rays_generated"] == W * H * first
    assert st["rays_generated"] <= st["rays_extension"] <= 4 * st["rays_generated"]
    assert st["shade_hits"] <= st["rays_extension"] and st["rays_shadow"] <= st["shade_hits"]
    assert 0 < st["deposits_shadow"] <= st["rays_shadow"] and st["deposits"] <= st["rays_shadow"] + st["rays_extension"]
    ref, cnt = O.render(sc, bundle.camera, W, H, first, seed=seed, pixels=px, threads=8)
    got, want = a[px], ref[px, :3]
    U.image_margins(f"{bundle.name} {label}, first {first} spp", got, want, first, bundle.camera, 1e-3, 1e-3)
    close = np.isclose(got, want, rtol=1e-3, atol=1e-3 * want.max()).all(axis=1)
    U.fraction_gate(f"{bundle.name} {label}, first {first} spp: sampled pixels within 1e-3 of the oracle", close, MEASURED, legacy=0.97)
    # the rest of the configuration's samples on top
    ctx.render(spp_total - first)
    assert ctx.samples_per_pixel == spp_total
    full = ctx.read_accum()[:, :3]
    st = ctx.stats()
    assert st["rays_generated"] == W * H * spp_total
    assert np.isfinite(full).all() and full.min() >= 0
    ctx.close()
    again = U.make_ctx(gpu, bundle, W, H, seed=seed)
    again.render(first)
    again.render(spp_total - first)
    assert np.array_equal(again.read_accum()[:, :3], full), "same seed, same batching: same bits"
    again.close()
    ref, _ = O.render(sc, bundle.camera, W, H, spp_total, seed=seed, pixels=px, threads=8)
    got, want = full[px], ref[px, :3]
    U.image_margins(f"{bundle.name} {label}, {spp_total} spp", got, want, spp_total, bundle.camera, 1e-3, 1e-3)
    return st


def test_config2_diffuse_mesh_binned_sah_1080p_256spp(gpu):
    """BASELINE.json configs[1]: the ~70k-triangle mesh (81 920 here: the bunny PLY does not travel to the GPU box,
    SURVEY 8d allows the seeded substitute), 3-axis binned-SAH BVH, diffuse 0.8, five-wall room + area light, 1080p, 256 spp."""
    b = scenes.blob_room(W, H, level=6, builder=gpu_host().BVH_BINNED_SAH)
    assert 70000 < len(b.flat.triangles) < 100000
    st = _config_room_test(gpu, b, 256, seed=21, label="config2 diffuse, binned SAH")
    # 8 samples first (one batch of 8: fewer than 16 samples of a pixel side by side -- k_gen and the per-ray kernel), then 248 in batches of the
    # context's 16 samples in flight, whose camera rays the bundle kernel generates and walks itself
    assert st["gen_launches"] >= 1 and st["bundle_launches"] >= 240 // 16, (st["gen_launches"], st["bundle_launches"], st["packet_launches"])


def test_config3_glass_mesh_sbvh_1080p_1024spp(gpu):
    """BASELINE.json configs[2]: the same room with Material::Refractive(smoothness 0.9, ior 1.5, colour (1,0.6,0.6),
    absorption 5) on the mesh (material.h:112-120) and an SBVH bottom level, 1080p, 1024 spp."""
    from ptamd import layout as L
    b = scenes.blob_room(W, H, level=6, builder=gpu_host().BVH_SPATIAL_SPLIT, material=L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0))
    st = _config_room_test(gpu, b, 1024, seed=22, label="config3 rough glass, SBVH")
    assert st["rays_extension"] > 1.5 * st["rays_generated"], "paths continue through the glass"


def gpu_host():
    from ptamd import host
    return host
