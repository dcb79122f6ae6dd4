"""HIP traversal kernels (through pt_intersect) against the oracle and against the golden vectors made
by the reference's own kernels.  Integer results (hit, prim, instance, occlusion) are exact; t,u,v are
fp32 with FMA contraction on the GPU: tolerance 2e-4 relative (stated in gpu_util.compare_hits)."""
import numpy as np
import pytest

import golden_io
import gpu_util as U
import orclib as O
from ptamd import host as H, layout as L, scenes

pytestmark = pytest.mark.gpu


MODES = ["two_level", "baked", "packet", "two_level_packet", "unbaked", "unbaked_packet", "two_level_parked", "unbaked_parked", "team"]


def _flags(gpu, mode):
    """two_level: instances are entered like the reference does (meshes that are a single leaf are still copied); unbaked: every
    instance is entered; baked: copied to world space (the default); *packet: pt_intersect runs the packet traversal kernel
    (one wave walks the tree once for 64 rays) -- on the world-space tree, or entering instances as a wave."""
    return {"two_level": gpu.FLAG_TWO_LEVEL_ONLY, "baked": 0, "packet": gpu.FLAG_PACKET_INTERSECT,
            "two_level_packet": gpu.FLAG_TWO_LEVEL_ONLY | gpu.FLAG_PACKET_INTERSECT, "unbaked": gpu.FLAG_NO_BAKED_INSTANCES,
            "unbaked_packet": gpu.FLAG_NO_BAKED_INSTANCES | gpu.FLAG_PACKET_INTERSECT,
            # *parked: the general route into an instance for every instance (rounds 2-4); without it (round 5) the per-ray kernels walk instances that are a
            # translation + uniform scale through entry nodes, the ray taken into the instance's space on the fly
            "two_level_parked": gpu.FLAG_TWO_LEVEL_ONLY | gpu.FLAG_PARKED_INSTANCES, "unbaked_parked": gpu.FLAG_NO_BAKED_INSTANCES | gpu.FLAG_PARKED_INSTANCES,
            # team: four lanes per ray on the world-space tree (pt_team.h: the kernel of launches that do not fill the machine)
            "team": gpu.FLAG_TEAM_INTERSECT}[mode]


def _entered(mode):
    return mode.startswith(("two_level", "unbaked"))


def _check_kernel_used(ctx, mode, folded=None, general=None):
    assert (ctx.stats()["packet_launches"] > 0) == mode.endswith("packet"), "wrong traversal kernel ran"
    assert (ctx.stats()["team_launches"] > 0) == (mode == "team"), "the team kernel did not run where it should (or ran where it should not)"
    if folded is not None:  # how many instances of the scene the per-ray kernels walk through entry nodes
        assert ctx.stats()["folded_instances"] == folded, (ctx.stats()["folded_instances"], folded)
    if general is not None:  # instances of any transform entered as leaf-kind steps (pt_trace.h, LEVELS 2)
        assert ctx.stats()["general_route"] == (1 if general else 0), (ctx.stats()["general_route"], general)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["mixed", "inst"])
def test_golden_closest_and_any_hit(gpu, golden, name, mode):
    """two_level: instances are entered like the reference does (ray transformed into instance space) -> hit/miss
    identical to the reference kernels.  Default: instances baked to world space while the byte budget lasts; a
    triangle edge is then rounded in world space, so a ray within round-off of an edge may change sides -- at most
    5e-4 of the rays, each verified to graze an edge.  The packet kernel tests, per ray, the same boxes and triangles
    with the same arithmetic as the per-ray kernel on the baked tree."""
    two_level = _entered(mode)
    flat, cam, sky, tex = golden_io.scene_inputs(golden, name)
    ctx = U.make_ctx(gpu, flat, 64, 36, camera=cam, sky=sky, tex=tex, flags=_flags(gpu, mode))
    o, d = golden[f"isect_{name}_o"], golden[f"isect_{name}_d"]
    got = ctx.intersect(o, d)
    want = dict(t=golden[f"isect_{name}_t"], u=golden[f"isect_{name}_uv"][:, 0], v=golden[f"isect_{name}_uv"][:, 1],
                prim=golden[f"isect_{name}_prim"], inst=golden[f"isect_{name}_inst"])
    info = U.compare_hits(flat, got, want, edge_flip_frac=0.0 if two_level else 5e-4)
    assert info["flips"] == 0
    occ = ctx.intersect(o, d, tmax=golden[f"shadow_{name}_len"], any_hit=True)["prim"]
    g = golden[f"shadow_{name}_occluded"]
    # segments cut 0.1 % before / after the first hit are decided by the same fp32 t on both sides
    assert (occ != g).sum() <= (2 if two_level else 2 + info["edge_flips"]), f"{(occ != g).sum()} occlusion verdicts differ"
    _check_kernel_used(ctx, mode)
    ctx.close()


@pytest.mark.parametrize("builder", [H.BVH_BINNED_SAH, H.BVH_BINNED_FAST, H.BVH_SPATIAL_SPLIT])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("rotate", [False, True])
def test_random_rays_two_level(gpu, builder, mode, rotate):
    if rotate and builder != H.BVH_SPATIAL_SPLIT:
        pytest.skip("rotated instances: one builder is enough")
    two_level = _entered(mode)
    b = scenes.instanced_grid(64, 36, level=4, builder=builder, sky_size=(16, 8), rotate=rotate)
    ctx = U.make_ctx(gpu, b, 64, 36, flags=_flags(gpu, mode))
    sc = U.oracle_scene(b)
    o, d = U.random_rays(60000, 5, (-4, 0.05, -4), (4, 3, 4))
    got, want = ctx.intersect(o, d), O.intersect_batch(sc, o, d, threads=8)
    info = U.compare_hits(b.flat, got, want, edge_flip_frac=0.0 if two_level else 5e-4)
    assert info["n"] > 10000 and info["flips"] == 0
    tmax = np.random.default_rng(1).uniform(0.05, 4, len(o)).astype(np.float32)
    occ = ctx.intersect(o, d, tmax=tmax, any_hit=True)["prim"]
    ref = O.intersect_batch(sc, o, d, tmax=tmax, any_hit=True, threads=8)["prim"]
    assert (occ != ref).sum() <= 3 + info["edge_flips"]
    # translated + uniformly scaled instances that are entered are folded (12 meshes; with nothing copied also the two quads); a scene with a rotated
    # instance takes the general route for every entered instance (round 6: leaf-kind entry steps, nothing parked), the *parked modes park (rounds 2-4)
    quads = 2 if mode.startswith("unbaked") else 0  # the ground and the light: single-leaf meshes, copied to world space unless nothing is
    want_folded = 0 if (mode.endswith("parked") or not two_level or rotate) else quads + 12
    _check_kernel_used(ctx, mode, want_folded, general=two_level and rotate and not mode.endswith("parked"))
    ctx.close()


@pytest.mark.parametrize("kind", ["baked", "two_level"])
def test_a_launch_of_between_one_and_two_rays_per_lane(gpu, kind):
    """A launch of 1-2 rays per lane: every wave of the per-ray kernels takes its static packet of 64, the rest comes from the shared cursor in spans
    (pt_trace.h, requestPacket).  600 000 rays in one launch (459 k lanes) must find what the same rays find in two launches of 300 000 (static
    packets only), closest hit and any hit, ray for ray; and so must the ragged sizes around the boundaries of the static part."""
    b = scenes.instanced_grid(64, 36, level=4, sky_size=(16, 8))
    ctx = U.make_ctx(gpu, b, 64, 36, flags=gpu.FLAG_TWO_LEVEL_ONLY if kind == "two_level" else 0)
    lanes = 256 * 4 * 7 * 64  # the persistent grid of the per-ray kernels on an MI355X
    for n in (600_000, lanes + 1, 2 * lanes, 2 * lanes + 64 + 5):
        o, d = U.random_rays(n, 11, (-4, 0.05, -4), (4, 3, 4))
        tmax = np.random.default_rng(2).uniform(0.05, 4, n).astype(np.float32)
        h = n // 2
        for any_hit in (False, True):
            kw = dict(any_hit=True) if any_hit else {}
            whole = ctx.intersect(o, d, tmax=tmax if any_hit else None, **kw)
            parts = [ctx.intersect(o[s], d[s], tmax=tmax[s] if any_hit else None, **kw) for s in (slice(0, h), slice(h, n))]
            for k in ("prim", "inst") if not any_hit else ("prim",):
                assert np.array_equal(whole[k], np.concatenate([q[k] for q in parts])), (n, any_hit, k)
            if not any_hit:
                assert np.array_equal(whole["t"].view(np.uint32), np.concatenate([q["t"] for q in parts]).view(np.uint32))
                assert 0.2 < (whole["prim"] >= 0).mean() < 1.0
    ctx.close()


@pytest.mark.parametrize("mode", ["baked", "packet"])
def test_edge_cases(gpu, mode):
    """empty queue tail / ragged sizes, axis-parallel rays (zero components), origins exactly on box faces
    and at 0, rays starting inside a box, rays that miss everything, degenerate zero-length shadow rays."""
    b = scenes.cornell_box(32, 32)
    ctx = U.make_ctx(gpu, b, 32, 32, flags=_flags(gpu, mode))
    sc = U.oracle_scene(b)
    for n in (1, 63, 64, 65, 1000):  # ragged wave fills
        o, d = U.random_rays(n, n, (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9))
        U.compare_hits(b.flat, ctx.intersect(o, d), O.intersect_batch(sc, o, d))
    o = np.zeros((12, 3), np.float32)
    o[:, 1] = 1.0
    d = np.zeros((12, 3), np.float32)
    for k in range(6):
        d[k, k % 3] = 1.0 if k < 3 else -1.0
        d[6 + k, k % 3] = 1.0 if k < 3 else -1.0
    o[6:, 0], o[6:, 2] = -1.0, 1.0  # on the left wall plane and the back wall plane
    got, want = ctx.intersect(o, d), O.intersect_batch(sc, o, d)
    gh, wh = got["prim"] >= 0, want["prim"] >= 0
    assert np.array_equal(gh[:6], wh[:6])
    # rays 6..11 lie IN two box faces with a zero direction component: the reference's slab test evaluates
    # 0 * inf there, so whether a box is entered is an accident of the tree; a verdict may only differ where the
    # ray grazes a triangle's boundary (corner / edge hit)
    for k in np.nonzero(gh != wh)[0]:
        h = got if gh[k] else want
        u, v = float(h["u"][k]), float(h["v"][k])
        assert min(abs(u), abs(v), abs(1.0 - u - v)) < 1e-5, (k, u, v)
    hit = gh & wh
    assert np.allclose(got["t"][hit], want["t"][hit], rtol=1e-5)
    far = np.array([[0, 50, 0]], np.float32)
    up = np.array([[0, 1, 0]], np.float32)
    r = ctx.intersect(far, up)
    assert r["prim"][0] == -1 and np.isinf(r["t"][0]) and r["inst"][0] == -1
    z = ctx.intersect(o[:4], d[:4], tmax=np.zeros(4, np.float32), any_hit=True)["prim"]
    assert not z.any()
    ctx.close()


@pytest.mark.parametrize("kind", ["baked", "two_level", "unbaked_rotated"])
@pytest.mark.parametrize("thin", [False, True])
def test_beam_packets_find_the_hits_of_the_per_ray_kernel(gpu, thin, kind):
    """Closest-hit packets whose 64 rays point into one octant are walked with ONE conservative beam test per node
    (pt_packet.h) instead of 64 ray tests.  The beam may enter boxes no ray enters, never skip one a ray enters, and
    the triangle tests are the per-ray kernel's: hits must be identical (same triangle, same t / u / v bits) except
    at exact-t ties (duplicated SBVH references).  Three packet shapes: the samples of one pixel (a thin beam), 64 neighbouring pixels (a wide
    one), and packets that mix rays of distant pixels (many straddle an octant boundary: the per-lane fallback)."""
    W, Hh = 256, 144
    # two_level / unbaked_rotated: the packet enters instances as a wave (ray transformed per lane, beam rebuilt in the instance's
    # space); with rotated instances some packets point into more than one octant after the transform and start over per lane
    b = scenes.instanced_grid(W, Hh, level=4, thin_lens=thin, sky_size=(16, 8), rotate=kind == "unbaked_rotated")
    base = {"baked": 0, "two_level": gpu.FLAG_TWO_LEVEL_ONLY, "unbaked_rotated": gpu.FLAG_NO_BAKED_INSTANCES}[kind]
    packet = U.make_ctx(gpu, b, W, Hh, flags=base | gpu.FLAG_PACKET_INTERSECT)
    per_ray = U.make_ctx(gpu, b, W, Hh, flags=base)
    rows = []
    for sample in range(4):
        o, d, _ = per_ray.gen_rays(sample, W * Hh)
        rows.append((o, d))
    o_n, d_n = rows[0]  # consecutive entries = neighbouring pixels of a row
    # the samples of one pixel next to each other: entry (pixel, sample) for 64 'samples' made of 4 real samples x 16 jitters
    rng = np.random.default_rng(3)
    pix = rng.choice(W * Hh, 600, replace=False)
    o_s = np.concatenate([np.repeat(rows[k][0][pix], 16, axis=0).reshape(len(pix), 16, 3) for k in range(4)], axis=1).reshape(-1, 3)
    d_s = np.concatenate([np.repeat(rows[k][1][pix], 16, axis=0).reshape(len(pix), 16, 3) for k in range(4)], axis=1).reshape(-1, 3)
    d_s = (d_s + rng.normal(0, 2e-4, d_s.shape)).astype(np.float32)
    perm = rng.permutation(len(o_n))[:20000]
    for name, o, d in (("neighbours", o_n, d_n), ("one pixel", o_s.astype(np.float32), d_s), ("shuffled", o_n[perm], d_n[perm])):
        got, want = packet.intersect(o, d), per_ray.intersect(o, d)
        same = (got["prim"] == want["prim"]) & (got["inst"] == want["inst"])
        # a differing record is a tie -- both kernels found a hit at the same distance, the traversal order picked the winner: an
        # SBVH mesh holds the triangles it split once per leaf that references them, the same triangle under two indices
        diff = ~same
        assert diff.mean() < 1e-3, (name, diff.sum())
        assert np.allclose(got["t"][diff], want["t"][diff], rtol=1e-6), name
        for k in ("t", "u", "v"):
            assert np.array_equal(got[k][same], want[k][same]), (name, k)
        assert (got["prim"] >= 0).mean() > 0.3, name
    assert packet.stats()["packet_launches"] > 0 and per_ray.stats()["packet_launches"] == 0
    packet.close()
    per_ray.close()


def _rays_leaving_surfaces(ctx, W, Hh, per_pixel, seed, towards=None):
    """Packets of rays that start within one pixel's footprint of each other, the way k_shade queues them: for `per_pixel` jittered camera
    rays of every pixel the hit point, lifted off the surface, and a direction -- random (the first bounce's extension rays) or
    towards a small area (its shadow rays, with their lengths)."""
    rng = np.random.default_rng(seed)
    o0, d0, _ = ctx.gen_rays(0, W * Hh)
    o = np.repeat(o0, per_pixel, axis=0)
    d = np.repeat(d0, per_pixel, axis=0) + rng.normal(0, 3e-4, (len(o0) * per_pixel, 3)).astype(np.float32)
    h = ctx.intersect(o, d.astype(np.float32))
    keep = h["prim"] >= 0  # (compaction: the rays that left the scene spawn nothing)
    p = (o + d * h["t"][:, None])[keep]
    if towards is None:
        nd = rng.normal(size=p.shape)
        nd /= np.linalg.norm(nd, axis=1, keepdims=True)
        p = p + 1e-3 * nd
        return p.astype(np.float32), nd.astype(np.float32), None
    target = np.asarray(towards[0], np.float32) + rng.uniform(-0.5, 0.5, p.shape).astype(np.float32) * np.asarray(towards[1], np.float32)
    nd = target - p
    length = np.linalg.norm(nd, axis=1)
    nd /= length[:, None]
    p = p + 1e-3 * nd
    return p.astype(np.float32), nd.astype(np.float32), (length - 2e-3).astype(np.float32)


@pytest.mark.parametrize("kind", ["baked", "two_level", "unbaked_rotated", "baked_thin_lens", "two_level_thin_lens", "unbaked_rotated_thin_lens"])
@pytest.mark.parametrize("spp", [64, 16])
def test_first_pass_of_a_batch_finds_the_hits_of_the_per_ray_kernel(gpu, kind, spp):
    """pt_primary_pass runs the first pass of a batch the way pt_render does: the camera rays are generated inside the traversal kernel
    and bundles of several packets (the samples of one pixel, or of a few neighbouring ones) walk the tree as ONE bundle (pt_packet_multi.h) --
    a pinhole's around their common origin, a thin lens's (*thin_lens, round 6) as a converging bundle around its waist on the focal plane, every
    lane keeping four origins.  The rays it queues for the shading kernel must be the bits k_gen writes, and every
    hit record the per-ray kernel's for that ray: same triangle and same t / u / v bits, except at exact-t ties."""
    W, Hh = 96, 54
    # two_level / unbaked_rotated: the bundle enters instances as a wave; behind a rotation some bundles point into more than one octant
    # and start over sub-packet by sub-packet
    b = scenes.instanced_grid(W, Hh, level=4, thin_lens=kind.endswith("thin_lens"), sky_size=(16, 8), rotate=kind.startswith("unbaked_rotated"))
    base = {"two_level": gpu.FLAG_TWO_LEVEL_ONLY, "unbaked_rotated": gpu.FLAG_NO_BAKED_INSTANCES}.get(kind.replace("_thin_lens", ""), 0)
    first = U.make_ctx(gpu, b, W, Hh, flags=base, samples_in_flight=spp)
    queued = U.make_ctx(gpu, b, W, Hh, flags=base | gpu.FLAG_QUEUE_PRIMARY_RAYS | gpu.FLAG_NO_PACKETS, samples_in_flight=spp)
    n = W * Hh * spp
    for sample in (0, spp):
        o, d, pixel, got = first.primary_pass(sample, spp, n)
        o2, d2, pixel2, want = queued.primary_pass(sample, spp, n)  # k_gen + the per-ray kernel
        assert np.array_equal(pixel, pixel2)
        assert np.array_equal(o.view(np.uint32), o2.view(np.uint32)) and np.array_equal(d.view(np.uint32), d2.view(np.uint32))
        again = queued.intersect(o, d)  # the hook, rays handed in
        for k in ("t", "u", "v", "prim", "inst"):
            assert np.array_equal(again[k], want[k]), k
        same = (got["prim"] == want["prim"]) & (got["inst"] == want["inst"])
        diff = ~same
        assert diff.mean() < 1e-3, diff.sum()
        assert np.allclose(got["t"][diff], want["t"][diff], rtol=1e-6)
        for k in ("t", "u", "v"):
            assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
        assert 0.3 < (got["prim"] >= 0).mean() < 1.0
    assert first.stats()["packet_launches"] == 2 and queued.stats()["packet_launches"] == 0
    assert first.stats()["bundle_launches"] == 2  # (rounds 3-5: a thin lens went through packets of 64)
    first.close()
    queued.close()


@pytest.mark.parametrize("spp", [16, 48])
def test_first_pass_with_a_ragged_tail_and_a_pixel_list(gpu, spp):
    """The bundles of the first pass where the queue does not end on a bundle boundary and the context owns only some tiles of the frame
    (pixel = pixelList[k]): the last bundle is walked sub-packet by sub-packet, every entry is written exactly once."""
    W, Hh = 49, 31
    b = scenes.instanced_grid(W, Hh, level=3, sky_size=(16, 8))
    tiles = [(0, 0, 49, 7), (8, 11, 31, 20), (40, 25, 49, 31)]  # x0, y0, x1, y1
    owned = sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles)
    n = owned * spp
    assert n % 256 != 0
    # (a context rounds its samples in flight down to a power of two -- what pt_render's batches are cut to --; the hook takes any batch up to that)
    in_flight = 1 << (spp - 1).bit_length()
    first = U.make_ctx(gpu, b, W, Hh, samples_in_flight=in_flight)
    queued = U.make_ctx(gpu, b, W, Hh, flags=gpu.FLAG_QUEUE_PRIMARY_RAYS | gpu.FLAG_NO_PACKETS, samples_in_flight=in_flight)
    for ctx in (first, queued):
        ctx.set_tiles(tiles)
    o, d, pixel, got = first.primary_pass(3, spp, n)
    o2, d2, pixel2, want = queued.primary_pass(3, spp, n)
    assert np.array_equal(pixel, pixel2) and len(np.unique(pixel)) == owned
    assert np.array_equal(o.view(np.uint32), o2.view(np.uint32)) and np.array_equal(d.view(np.uint32), d2.view(np.uint32))
    same = (got["prim"] == want["prim"]) & (got["inst"] == want["inst"])
    assert (~same).mean() < 1e-3
    assert np.allclose(got["t"][~same], want["t"][~same], rtol=1e-6)
    for k in ("t", "u", "v"):
        assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
    assert first.stats()["bundle_launches"] == 1
    first.close()
    queued.close()


@pytest.mark.parametrize("mode", ["baked", "unbaked", "unbaked_packet"])
def test_thousand_instances_of_a_small_mesh(gpu, mode):
    """1 000 rotated, scaled instances of a 1 280-triangle mesh (1.28 M instanced triangles): here the TOP level is the deep tree
    (a ray crosses many instance boxes).  Camera-like and random rays against the oracle, closest hit and any hit."""
    W, Hh = 256, 144
    b = scenes.instance_field(W, Hh, n=1000, level=3)
    ctx = U.make_ctx(gpu, b, W, Hh, flags=_flags(gpu, mode))
    sc = U.oracle_scene(b)
    o1, d1, _ = ctx.gen_rays(0, W * Hh)
    side = 0.45 * 1000 ** 0.5
    o2, d2 = U.random_rays(40000, 11, (-side, 0.05, -side), (side, 4, side))
    for o, d in ((o1, d1), (o2, d2)):
        got, want = ctx.intersect(o, d), O.intersect_batch(sc, o, d, threads=8)
        # world-space copies of ROTATED instances: a ray through the gap round-off opens at a shared edge hits what lies behind
        # (the copy's edges are rounded in another space than the reference's): a handful of rays, t within 1 %
        # (2 cm triangles up to 40 units away: u and v themselves are only good to a few percent, the hit point they encode to 2e-4)
        # and a ray that grazes one of them may miss it on one side (the ray is taken into the instance with FMAs here, without in the
        # oracle) and report the triangle behind: a handful of rays, each checked to graze an edge
        info = U.compare_hits(b.flat, got, want, edge_flip_frac=5e-4, t_outlier_frac=5e-4, uv_atol=5e-2, t_atol=2e-5)  # coordinates up to 20
        assert info["n"] > 10000 and info["flips"] == 0
        slack = int(5e-4 * len(o))
        tmax = np.random.default_rng(2).uniform(0.05, 12, len(o)).astype(np.float32)
        occ = ctx.intersect(o, d, tmax=tmax, any_hit=True)["prim"]
        ref = O.intersect_batch(sc, o, d, tmax=tmax, any_hit=True, threads=8)["prim"]
        assert (occ != ref).sum() <= 3 + info["edge_flips"] + slack
    _check_kernel_used(ctx, mode)
    ctx.close()


@pytest.mark.parametrize("n", [60, 200])
def test_translated_and_scaled_instances_around_the_fold_table(gpu, n):
    """The per-ray kernels walk translated + uniformly scaled instances without parking while the scene's instances fit the LDS table of their transforms
    (95 entries, pt_trace.h); a scene with more takes the general route (round 6: leaf-kind entry steps; rounds 2-5: the parked route for all of them).  Both sides of that limit, every instance entered
    (PT_FLAG_NO_BAKED_INSTANCES): hits equal the oracle's, occlusion verdicts too, with mixed scales and a rotated instance in between (the general
    route and the folded one in one traversal)."""
    mat = L.material_pbr_dielectric((0.7, 0.3, 0.2), 0.6)
    mesh = scenes.blob_mesh(mat, level=2, seed=3, builder=H.BVH_BINNED_SAH)
    scene = H.Scene()
    mb = scenes._MeshBuilder()
    ext = 0.5 * n ** 0.5 + 2
    mb.add_quad((-ext, 0, -ext), (-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext), 0)
    scene.add_node(mb.build([L.material_diffuse((0.6, 0.6, 0.6))], H.BVH_BINNED_SAH))
    rng = np.random.default_rng(n)
    side = 0.45 * n ** 0.5
    for k in range(n - 1):
        s = float(rng.uniform(0.4, 1.3))
        q = (1, 0, 0, 0) if k != 7 else (float(np.cos(0.4)), 0.0, float(np.sin(0.4)), 0.0)  # one rotated instance: the general route
        scene.add_node(mesh, location=(float(rng.uniform(-side, side)), float(rng.uniform(0.4, 2.0)), float(rng.uniform(-side, side))), orientation_wxyz=q, scale=(s, s, s))
    flat = scene.flatten()
    assert flat.num_instances == n
    ctx = gpu.Context(64, 36, flags=gpu.FLAG_NO_BAKED_INSTANCES)
    ctx.upload_scene(flat)
    # 60: the kernels' LDS table holds the transforms, and one turned instance in 60 is few enough for the folded route to stay (that one parks); 200
    # (round 6; rounds 2-5 parked them all): the general route, every instance entered as a leaf-kind step (pt_trace.h, LEVELS 2)
    st = ctx.stats()
    assert (st["folded_instances"], st["general_route"]) == ((n - 1, 0) if n + 1 <= 96 else (0, 1)), st
    sc = O.BoundScene(flat)
    o, d = U.random_rays(40000, n, (-side, 0.05, -side), (side, 3, side))
    got, want = ctx.intersect(o, d), O.intersect_batch(sc, o, d, threads=8)
    info = U.compare_hits(flat, got, want, uv_atol=2e-2, t_atol=1e-5)
    assert info["n"] > 8000 and info["flips"] == 0
    tmax = np.random.default_rng(1).uniform(0.05, 6, len(o)).astype(np.float32)
    occ = ctx.intersect(o, d, tmax=tmax, any_hit=True)["prim"]
    ref = O.intersect_batch(sc, o, d, tmax=tmax, any_hit=True, threads=8)["prim"]
    assert (occ != ref).sum() <= 3
    ctx.close()


def test_a_scene_beyond_the_32_bit_offsets_is_refused(gpu, monkeypatch):
    """The traversal kernels address nodes and triangle records by base + 32-bit byte offset: a dynamic state whose node or triangle array (world-space
    copies included) would pass 4 GB is refused with a message that names the way out, not traversed with wrapped offsets.  (The limit is lowered
    through PTAMD_OFFSET_LIMIT here; entering the instances instead of copying them fits again.)"""
    b = scenes.instanced_grid(64, 36, level=4, sky_size=(16, 8))
    monkeypatch.setenv("PTAMD_OFFSET_LIMIT", str(2 << 20))  # 2 MB: the 12 world-space copies of the 5 120-triangle meshes need ~3 MB of triangle records
    with pytest.raises(RuntimeError, match="more than 4 GB|PT_FLAG_NO_BAKED_INSTANCES"):
        U.make_ctx(gpu, b, 64, 36)
    ctx = U.make_ctx(gpu, b, 64, 36, flags=gpu.FLAG_NO_BAKED_INSTANCES)  # nothing copied: 2 x 5 120 triangles
    o, d = U.random_rays(2000, 3, (-4, 0.05, -4), (4, 3, 4))
    assert (ctx.intersect(o, d)["prim"] >= 0).mean() > 0.2
    ctx.close()


def test_invalid_scenes_are_rejected_not_traversed(gpu):
    b = scenes.cornell_box(16, 16)
    f = b.flat
    ctx = gpu.Context(16, 16)
    bad = f.triangles.copy()
    bad["indices"][3, 1] = len(f.vertices) + 5
    with pytest.raises(gpu.PtError, match="vertex index"):
        ctx._chk(gpu.lib().pt_upload_static(ctx._h, f.vertices.ctypes.data, len(f.vertices), bad.ctypes.data, len(bad),
                                            f.materials.ctypes.data, len(f.materials), f.sub_nodes.ctypes.data, len(f.sub_nodes)), "upload")
    nodes = f.sub_nodes.copy()
    leaf = np.flatnonzero(nodes["count"] != 0)[0]
    nodes["count"][leaf] = 10 ** 6
    with pytest.raises(gpu.PtError, match="out of bounds"):
        ctx._chk(gpu.lib().pt_upload_static(ctx._h, f.vertices.ctypes.data, len(f.vertices), f.triangles.ctypes.data, len(f.triangles),
                                            f.materials.ctypes.data, len(f.materials), nodes.ctypes.data, len(nodes)), "upload")
    with pytest.raises(gpu.PtError, match="scene"):
        ctx.render(1)
    ctx.close()


def test_large_leaves_are_split_at_upload(gpu):
    """100 coincident triangles make one 100-triangle leaf (no SAH gain); device leaves hold <= 31."""
    pos = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (100, 1))
    pos[:, 2] = np.repeat(np.arange(100) * 1e-4, 3)
    mesh = H.Mesh(pos, np.arange(300, dtype=np.uint32).reshape(100, 3), [L.material_diffuse((1, 1, 1))], builder=H.BVH_BINNED_SAH)
    sc = H.Scene()
    sc.add_node(mesh)
    flat = sc.flatten()
    ctx = gpu.Context(8, 8)
    ctx.upload_scene(flat)
    o = np.array([[0.2, 0.2, -1.0], [0.2, 0.2, 1.0], [2, 2, -1]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 1]], np.float32)
    got, want = ctx.intersect(o, d), O.intersect_batch(O.BoundScene(flat), o, d)
    assert np.array_equal(got["prim"], want["prim"]) and np.allclose(got["t"][:2], want["t"][:2])
    ctx.close()


def _chain_scene(n):
    """n parallel triangles (z = 0.1 k) under a hand-made degenerate BVH: node k -> (leaf k, rest)."""
    pos = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (n, 1))
    pos[:, 2] = np.repeat(np.arange(n, dtype=np.float32) * 0.1, 3)
    mesh = H.Mesh(pos, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), [L.material_diffuse((1, 1, 1))], builder=H.BVH_BINNED_SAH)
    sc = H.Scene()
    sc.add_node(mesh)
    flat = sc.flatten()
    z = flat.vertices["vertex"][flat.triangles["indices"][:, 0], 2]  # triangles were reordered by the builder
    order = np.argsort(z, kind="stable")
    flat.triangles = flat.triangles[order]
    z = z[order]
    nodes = np.zeros(2 * n, L.SUB_BVH_NODE)  # node 0 root, node 1 the reference allocator's dummy, then pairs
    for k in range(n - 1):
        me = 0 if k == 0 else 2 * k + 1
        nodes[me]["min"][:3], nodes[me]["max"][:3] = (0, 0, z[k]), (1, 1, z[n - 1])
        nodes[me]["left"], nodes[me]["count"] = 2 * k + 2, 0
        leaf = 2 * k + 2
        nodes[leaf]["min"][:3], nodes[leaf]["max"][:3] = (0, 0, z[k]), (1, 1, z[k])
        nodes[leaf]["left"], nodes[leaf]["count"] = k, 1
    last = 2 * (n - 2) + 3
    nodes[last]["min"][:3], nodes[last]["max"][:3] = (0, 0, z[n - 1]), (1, 1, z[n - 1])
    nodes[last]["left"], nodes[last]["count"] = n - 1, 1
    nodes[1] = nodes[last]  # never referenced
    flat.sub_nodes = nodes
    return flat, z


def test_packet_kernel_stack_limit_and_fallback(gpu):
    """The packet kernel keeps at most 64 pending entries (one per lane of its stack registers): a 40-level chain
    fits, a 100-level chain does not and the per-ray kernel runs instead -- same answers either way."""
    for n, fits in ((40, True), (100, False)):
        flat, z = _chain_scene(n)
        ctx = gpu.Context(8, 8, flags=gpu.FLAG_PACKET_INTERSECT)
        ctx.upload_scene(flat)
        m = 1000
        rng = np.random.default_rng(n)
        xy = rng.uniform(0.05, 0.45, (m, 2)).astype(np.float32)
        o = np.concatenate([np.c_[xy, np.full(m, 100.0, np.float32)], np.c_[xy, np.full(m, -100.0, np.float32)]]).astype(np.float32)
        d = np.concatenate([np.tile([0, 0, -1], (m, 1)), np.tile([0, 0, 1], (m, 1))]).astype(np.float32)
        got = ctx.intersect(o, d)
        assert (got["prim"][:m] == n - 1).all() and (got["prim"][m:] == 0).all()
        assert np.allclose(got["t"][:m], 100.0 - z[n - 1], rtol=1e-5) and np.allclose(got["t"][m:], 100.0 + z[0], rtol=1e-5)
        occ = ctx.intersect(o[:m], d[:m], tmax=np.full(m, 100.0 - z[n - 1] - 0.05, np.float32), any_hit=True)["prim"]
        assert not occ.any()
        occ = ctx.intersect(o[:m], d[:m], tmax=np.full(m, 100.0 - z[n - 1] + 0.05, np.float32), any_hit=True)["prim"]
        assert occ.all()
        assert (ctx.stats()["packet_launches"] > 0) == fits
        ctx.close()


def test_deep_trees_use_the_spilled_stack_and_too_deep_ones_are_rejected(gpu):
    """A 100-level chain needs ~100 pending entries when entered from the far side: 16 live in LDS, the rest in
    the global spill region.  A 130-level chain exceeds LDS + spill and must be refused at upload."""
    n = 100
    flat, z = _chain_scene(n)
    ctx = gpu.Context(8, 8)
    ctx.upload_scene(flat)
    m = 4096
    rng = np.random.default_rng(5)
    xy = rng.uniform(0.05, 0.45, (m, 2)).astype(np.float32)
    o = np.concatenate([np.c_[xy, np.full(m, 100.0, np.float32)], np.c_[xy, np.full(m, -100.0, np.float32)]]).astype(np.float32)
    d = np.concatenate([np.tile([0, 0, -1], (m, 1)), np.tile([0, 0, 1], (m, 1))]).astype(np.float32)
    got = ctx.intersect(o, d)
    assert (got["prim"][:m] == n - 1).all() and (got["prim"][m:] == 0).all()
    assert np.allclose(got["t"][:m], 100.0 - z[n - 1], rtol=1e-5) and np.allclose(got["t"][m:], 100.0 + z[0], rtol=1e-5)
    # slanted rays: brute force over the n planes
    o2 = np.c_[rng.uniform(0.05, 0.3, (m, 2)), np.full(m, 50.0)].astype(np.float32)
    d2 = np.c_[rng.uniform(-0.002, 0.002, (m, 2)), np.full(m, -1.0)].astype(np.float32)
    got2 = ctx.intersect(o2, d2)
    t_all = (z[None, :] - o2[:, 2:3]) / d2[:, 2:3]
    p = o2[:, None, :2] + t_all[..., None] * d2[:, None, :2]
    inside = (p[..., 0] >= 0) & (p[..., 1] >= 0) & (p[..., 0] + p[..., 1] <= 1)
    t_all = np.where(inside, t_all, np.inf)
    want = np.where(np.isfinite(t_all.min(1)), t_all.argmin(1), -1)
    assert (got2["prim"] == want).mean() > 0.999  # edge-grazing rays may differ by round-off
    occ = ctx.intersect(o[:m], d[:m], tmax=np.full(m, 85.0, np.float32), any_hit=True)["prim"]  # stops short of every triangle
    assert not occ.any()
    occ = ctx.intersect(o[:m], d[:m], tmax=np.full(m, 95.0, np.float32), any_hit=True)["prim"]  # reaches half of them
    assert occ.all()
    ctx.close()
    flat2, _ = _chain_scene(130)
    ctx = gpu.Context(8, 8)
    with pytest.raises(RuntimeError, match="stack"):
        ctx.upload_scene(flat2)
    ctx.close()


def test_stack_bound_of_a_root_that_shares_its_subtree_with_an_earlier_root(gpu):
    """Two roots of the caller's sub-BVH array may share a subtree (nothing in the boundary's contract says the array is a forest): the
    second one is packed as a run of its own whose children lie in the FIRST root's run.  Its worst-case stack is then its own siblings
    plus the shared subtree's -- taking 0 for children packed earlier (round 3) would let a 100-level chain pass as shallow, and the packet
    kernel's 64-entry lane stack would wrap.  Node 1 (the reference allocator's unused dummy) becomes a second root over the chain from
    node 4 on; the top-level leaf names it."""
    n = 100
    flat, z = _chain_scene(n)
    flat.sub_nodes[1] = flat.sub_nodes[3]  # an inner node with the children of node 3: (leaf 1, the rest of the chain)
    assert flat.sub_nodes[1]["count"] == 0 and flat.sub_nodes[1]["left"] == 4
    leaves = np.flatnonzero(flat.top_nodes["isLeaf"] != 0)
    assert len(leaves) == 1
    flat.top_nodes["a"][leaves[0]] = 1
    ctx = gpu.Context(8, 8, flags=gpu.FLAG_PACKET_INTERSECT)
    ctx.upload_scene(flat)
    m = 1000
    rng = np.random.default_rng(n)
    xy = rng.uniform(0.05, 0.45, (m, 2)).astype(np.float32)
    o = np.concatenate([np.c_[xy, np.full(m, 100.0, np.float32)], np.c_[xy, np.full(m, -100.0, np.float32)]]).astype(np.float32)
    d = np.concatenate([np.tile([0, 0, -1], (m, 1)), np.tile([0, 0, 1], (m, 1))]).astype(np.float32)
    got = ctx.intersect(o, d)
    assert (got["prim"][:m] == n - 1).all() and (got["prim"][m:] == 1).all()  # triangle 0 hangs off node 0 only
    st = ctx.stats()
    assert st["stack_need"] >= n - 4, st["stack_need"]
    assert st["packet_launches"] == 0, "a tree that needs more than 64 stack entries must not reach the packet kernel"
    ctx.close()


@pytest.mark.parametrize("scene", ["general_30", "uniform_130", "mixed_40"])
def test_general_instance_route_finds_the_parked_routes_hits_bit_for_bit(gpu, scene):
    """pt_trace.h, LEVELS 2 (round 6): instances of ANY transform -- turned, scaled by three different factors -- and any number of them are entered as
    leaf-kind steps of the hot loop, the instance-space ray in a per-lane LDS slot, nothing parked.  The ray is taken into the instance's space by the
    parked route's arithmetic (rayIntoInstance) and both walk the same tree in the same order: every hit record -- t, u, v to the bit, triangle,
    instance -- and every occlusion verdict must be the parked route's (PT_FLAG_PARKED_INSTANCES), and the oracle's within the usual tolerances
    (scene.cl:116-139).  general_30: 30 turned + non-uniformly scaled instances; uniform_130: 130 translated + uniformly scaled ones (more than the
    95 the fold table holds); mixed_40: both kinds in one scene."""
    if scene == "general_30":
        b = scenes.instanced_crowd(64, 36, nx=6, nz=5, level=3, transform="general", sky_size=(16, 8))
    elif scene == "uniform_130":
        b = scenes.instanced_crowd(64, 36, nx=13, nz=10, level=2, transform="uniform", sky_size=(16, 8))
    else:
        b = scenes.instanced_crowd(64, 36, nx=8, nz=5, level=3, transform="mixed", sky_size=(16, 8))
    n_inst = {"general_30": 30, "uniform_130": 130, "mixed_40": 40}[scene]
    general = U.make_ctx(gpu, b, 64, 36, flags=gpu.FLAG_NO_BAKED_INSTANCES)
    parked = U.make_ctx(gpu, b, 64, 36, flags=gpu.FLAG_NO_BAKED_INSTANCES | gpu.FLAG_PARKED_INSTANCES)
    st = general.stats()
    assert st["general_route"] == 1 and st["folded_instances"] == 0 and st["entered_instances"] == n_inst + 2, st
    assert parked.stats()["general_route"] == 0
    span = 0.75 * max(b.crowd_extent)
    o, d = U.random_rays(120000, 17, (-span, 0.05, -3.0), (span, 3.0, b.crowd_extent[1]))
    o_c, d_c, _ = general.gen_rays(2, 64 * 36)
    # general_30: every transform goes through rayIntoInstance on both routes: the same bits everywhere.  Translated + uniformly scaled instances enter the general
    # route through its short form (the folded route's arithmetic: origin and direction to the same bits, 1 / direction = (1 / d) * s instead of 1 / (d / s): an ulp
    # apart) -- a box test may fall the other way, a hit record may not: same (t, u, v) bits for the same triangle, another winner only of an exact-t tie
    exact = scene == "general_30"
    for name, oo, dd in (("random", o, d), ("camera", o_c, d_c), ("ragged", o[:64 * 1000 + 17], d[:64 * 1000 + 17])):
        got, want = general.intersect(oo, dd), parked.intersect(oo, dd)
        same = (got["prim"] == want["prim"]) & (got["inst"] == want["inst"])
        assert same.all() if exact else (~same).mean() < 1e-4, (name, int((~same).sum()))
        assert np.allclose(got["t"][~same], want["t"][~same], rtol=1e-6), name
        for k in ("t", "u", "v"):
            assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), (name, k)
        assert (got["prim"] >= 0).mean() > 0.2, name
    tmax = np.random.default_rng(3).uniform(0.05, 6.0, len(o)).astype(np.float32)
    occ_g = general.intersect(o, d, tmax=tmax, any_hit=True)["prim"]
    occ_p = parked.intersect(o, d, tmax=tmax, any_hit=True)["prim"]
    assert (occ_g != occ_p).sum() <= (0 if exact else 2), int((occ_g != occ_p).sum())
    assert 0.05 < occ_g.mean() < 0.95
    sc = U.oracle_scene(b)
    sub = slice(0, 30000)
    # (small triangles seen from up to 30 units away, as in test_thousand_instances_of_a_small_mesh: a ray that grazes an edge may report the triangle behind)
    info = U.compare_hits(b.flat, general.intersect(o[sub], d[sub]), O.intersect_batch(sc, o[sub], d[sub], threads=8), edge_flip_frac=5e-4, t_outlier_frac=5e-4,
                          uv_atol=2e-2, t_atol=1e-5)
    assert info["n"] > 5000 and info["flips"] == 0
    ref = O.intersect_batch(sc, o[sub], d[sub], tmax=tmax[sub], any_hit=True, threads=8)["prim"]
    assert (occ_g[sub] != ref).sum() <= 3 + info["edge_flips"] + 15
    general.close()
    parked.close()
