"""Dynamic scenes (reference src/raytracer.cpp:183-189,497-595, src/bvh/refit_bvh.cpp, src/model/mesh_sequence.cpp:81-97):
the double-buffered dynamic part of the scene -- converted on the host and copied on a copy stream while frames keep rendering, adopted
by pt_frame_tick without a host-side wait -- and deforming meshes through refitted BVHs."""
import numpy as np
import pytest

import gpu_util as U
import orclib as O
from ptamd import host as H, layout as L, scenes

pytestmark = pytest.mark.gpu

# fractions of pixels / queue entries within tolerance of the oracle as measured on the MI355X (profiles/round6/parity_margins.json): gpu_util.fraction_gate holds
# every such comparison against 0.98 x its entry here (and never below the round-number gate of rounds 1-5)
MEASURED = {
    'deforming mesh (BVH_BINNED_SAH, device), frame 0: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_BINNED_SAH, device), frame 1: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_BINNED_SAH, device), frame 2: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_BINNED_SAH, host_nodes), frame 0: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_BINNED_SAH, host_nodes), frame 1: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_BINNED_SAH, host_nodes), frame 2: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_SPATIAL_SPLIT, device), frame 0: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_SPATIAL_SPLIT, device), frame 1: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_SPATIAL_SPLIT, device), frame 2: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_SPATIAL_SPLIT, host_nodes), frame 0: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_SPATIAL_SPLIT, host_nodes), frame 1: pixels within 1e-3 of the oracle': 1.0000,
    'deforming mesh (BVH_SPATIAL_SPLIT, host_nodes), frame 2: pixels within 1e-3 of the oracle': 1.0000,
}
W, Hh = 96, 54


def _states(n):
    """n scene states of a 2x2 instanced grid: the instances move, rescale and turn."""
    b = scenes.instanced_grid(W, Hh, nx=2, nz=2, level=3, sky_size=(16, 8))
    flats = [b.scene.flatten()]
    rng = np.random.default_rng(6)
    for k in range(1, n):
        for node in (2, 3, 4, 5):  # scene nodes 0/1 are the ground and the light
            a = rng.uniform(0, 2 * np.pi)
            s = float(rng.uniform(0.6, 1.2))
            b.scene.set_transform(node, location=(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0.6, 1.5)), float(rng.uniform(-1.5, 1.5))),
                                  scale=(s, s, s), orientation_wxyz=(float(np.cos(a / 2)), 0.0, float(np.sin(a / 2)), 0.0))
        flats.append(b.scene.flatten())
    return b, flats


@pytest.mark.parametrize("instances", ["copied", "entered_general_route"])
def test_asynchronous_uploads_render_every_state_like_a_fresh_context(gpu, instances):
    """(entered_general_route, round 6: nothing copied -- the turned instances are entered through the general route, whose entry records are part of the
    double-buffered dynamic set like the instance table.)  A frame loop that never waits: render state k (asynchronous), convert + copy state k+1 into the inactive buffers while it
    renders, tick, render k+1, ...  Each frame's accumulator is copied out in stream order by a torch op on the same stream.
    Every frame equals the render of a fresh context that only ever saw that state (counter PRNG: bit for bit) -- no frame
    reads a half-written or a too-new set although nothing synchronises with the host in between."""
    import torch
    b, flats = _states(5)
    s = torch.cuda.Stream()
    frames = []
    with torch.cuda.stream(s):
        acc = torch.zeros(W * Hh, 4, device="cuda")
        flags = gpu.FLAG_NO_BAKED_INSTANCES if instances == "entered_general_route" else 0
        ctx = gpu.Context(W, Hh, seed=5, flags=flags)
        ctx.set_stream(s.cuda_stream)
        ctx.set_accum_buffer(acc.data_ptr())
        ctx.upload_scene(flats[0], sky=b.sky)
        ctx.set_camera(b.camera)
        for k in range(len(flats)):
            ctx.clear()
            ctx.render(8, sync=False)  # state k, in flight ...
            frames.append(acc.clone())  # ... its image, taken in stream order
            if k + 1 < len(flats):
                ctx.upload_dynamic_async(flats[k + 1])  # host conversion + copy-stream upload overlap the render above
                ctx.frame_tick()
        got = [f.cpu().numpy() for f in frames]
        general = ctx.stats()["general_route"]
    ctx.close()
    assert general == (1 if flags else 0)  # (the last state: four turned instances entered)
    for k, flat in enumerate(flats):
        fresh = gpu.Context(W, Hh, seed=5, flags=flags)
        fresh.upload_scene(flat, sky=b.sky)
        fresh.set_camera(b.camera)
        fresh.render(8)
        want = fresh.read_accum()
        fresh.close()
        assert np.array_equal(got[k], want), f"frame {k} differs from a fresh render of its state"
        assert k == 0 or not np.array_equal(got[k], got[k - 1])


def test_an_upload_that_is_never_adopted_changes_nothing_and_can_be_replaced(gpu):
    b, flats = _states(3)
    ctx = U.make_ctx(gpu, flats[0], W, Hh, camera=b.camera, sky=b.sky, seed=2)
    ctx.render(4)
    a0 = ctx.read_accum().copy()
    ctx.upload_dynamic_async(flats[1])  # pending, not adopted: renders still see state 0
    ctx.clear()
    ctx.render(4)
    assert np.array_equal(ctx.read_accum(), a0)
    ctx.upload_dynamic_async(flats[2])  # replaces the pending state
    ctx.frame_tick()
    ctx.frame_tick()  # a second tick without an upload is a no-op
    ctx.clear()
    ctx.render(4)
    a2 = ctx.read_accum().copy()
    ctx.close()
    fresh = U.make_ctx(gpu, flats[2], W, Hh, camera=b.camera, sky=b.sky, seed=2)
    fresh.render(4)
    assert np.array_equal(a2, fresh.read_accum()) and not np.array_equal(a2, a0)
    fresh.close()


@pytest.mark.parametrize("route", ["host_nodes", "device"])
@pytest.mark.parametrize("builder", ["BVH_BINNED_SAH", "BVH_SPATIAL_SPLIT"])
def test_deforming_mesh_through_refit(gpu, builder, route):
    """Mesh.refit + pt_update_geometry + pt_upload_dynamic: a blob in the room twists and squashes over three frames, its BVH
    refitted, never rebuilt.  Every frame: closest hits equal the oracle's on the refitted arrays, the image equals the oracle's
    path by path, and a context created from scratch with the same (refitted) arrays renders the same bits.
    route "host_nodes": the reference's way -- boxes refitted on the host (refitBVH), the whole vertex and node arrays handed over
    (pt_update_geometry); "device" (round 5): only the blob's vertices travel (pt_refit_vertices), the device recomputes every box of its
    trees bottom-up -- and must end up with the very bytes the other route makes from the host-refitted nodes (the fresh context below is
    given those)."""
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    room = scenes.blob_room(W, Hh, level=3, builder=getattr(H, builder), material=mat)
    v, f = scenes.icosphere(3)
    p0 = (v * 0.5).astype(np.float32)
    blob = H.Mesh(p0, f.astype(np.uint32), [mat], builder=getattr(H, builder))
    scene = H.Scene()
    mb = scenes._MeshBuilder()
    mats = scenes._room_materials()
    scenes._room(mb, mats)
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    scene.add_node(blob, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
    flat = scene.flatten()
    ctx = U.make_ctx(gpu, flat, W, Hh, camera=room.camera, seed=8, samples_in_flight=1)
    o, d = U.random_rays(20000, 4, (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9))
    prev = None
    for frame in range(3):
        if frame:
            ang = 0.9 * frame * p0[:, 1]
            p = np.stack([np.cos(ang) * p0[:, 0] - np.sin(ang) * p0[:, 2], (1 - 0.2 * frame) * p0[:, 1] + 0.05 * frame * np.sin(5 * p0[:, 0]),
                          np.sin(ang) * p0[:, 0] + np.cos(ang) * p0[:, 2]], 1)
            blob.refit(p)
            if route == "device":
                first_vertex, _ = scene.mesh_offsets(blob)
                ctx.refit_vertices(first_vertex, blob.vertices_view())
                dyn, _ = scene.flatten_dynamic_only()
                ctx.upload_dynamic(dyn)
                flat = scene.flatten()  # (for the oracle and the fresh context: the host refits its boxes now, when they are asked for)
            else:
                flat = scene.flatten()
                ctx.update_geometry(flat)
                ctx.upload_dynamic(flat)
        ctx.clear()
        ctx.render(16)
        a, st = ctx.read_accum()[:, :3].copy(), ctx.stats()
        sc = O.BoundScene(flat)
        try:
            info = U.compare_hits(flat, ctx.intersect(o, d), O.intersect_batch(sc, o, d, threads=8), edge_flip_frac=5e-4, t_outlier_frac=5e-4)
        except AssertionError as e:
            raise AssertionError(f"frame {frame}: {e}") from e
        assert info["n"] > 5000
        ref, _ = O.render(sc, room.camera, W, Hh, 16, seed=8, threads=8)
        assert abs(a.mean() - ref[:, :3].mean()) / ref[:, :3].mean() < 1e-3
        U.fraction_gate(f"deforming mesh ({builder}, {route}), frame {frame}: pixels within 1e-3 of the oracle",
                        np.isclose(a, ref[:, :3], rtol=1e-3, atol=1e-3 * ref.max()).all(axis=1), MEASURED, legacy=0.97)
        fresh = U.make_ctx(gpu, flat, W, Hh, camera=room.camera, seed=8, samples_in_flight=1)
        fresh.render(16)
        assert np.array_equal(fresh.read_accum()[:, :3], a), f"frame {frame}: refitted context != fresh context on the same arrays"
        fresh.close()
        assert prev is None or not np.array_equal(prev, a)
        prev = a
    # a rebuilt tree is not a refit: the topology check refuses it
    other = H.Mesh(p0 * 0.9, f.astype(np.uint32), [mat], builder=H.BVH_BINNED_FAST)
    scene2 = H.Scene()
    scene2.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    scene2.add_node(other, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
    flat2 = scene2.flatten()
    if len(flat2.sub_nodes) == len(flat.sub_nodes) and not np.array_equal(flat2.sub_nodes["left"], flat.sub_nodes["left"]):
        with pytest.raises(gpu.PtError, match="topology"):
            ctx.update_geometry(flat2)
    ctx.close()


def _grid_scene(meshes, placements, seed):
    """ground + light + one instance per placement (mesh index, x, z, scale, yaw)"""
    scene = H.Scene()
    mb = scenes._MeshBuilder()
    mb.add_quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), 0)
    scene.add_node(mb.build([L.material_pbr_dielectric((0.5, 0.5, 0.5), 0.3)], H.BVH_BINNED_SAH))
    lb = scenes._MeshBuilder()
    lb.add_quad((-0.6, 0, -0.6), (0.6, 0, -0.6), (0.6, 0, 0.6), (-0.6, 0, 0.6), 0)
    scene.add_node(lb.build([L.material_emissive((1.0, 0.9, 0.75), 30.0)], H.BVH_BINNED_SAH), location=(0.0, 4.0, 0.0))
    for m, x, z, s, yaw in placements:
        scene.add_node(meshes[m], location=(x, 0.62 * s, z), scale=(s, s, s), orientation_wxyz=(float(np.cos(yaw / 2)), 0.0, float(np.sin(yaw / 2)), 0.0))
    return scene


@pytest.mark.parametrize("flags_name", ["copied", "entered"])
def test_states_with_more_and_fewer_instances(gpu, flags_name):
    """The static arrays (two meshes) are uploaded ONCE; the states that follow hold 2, 9 and 3 instances of them.  The dynamic sets grow
    (more world-space copies than any state before: fresh buffers, the static part put in again from the master copy) and are reused when
    a state needs less; every state renders like a fresh context that only ever saw it."""
    flags = 0 if flags_name == "copied" else gpu.FLAG_NO_BAKED_INSTANCES
    meshes = [scenes.blob_mesh(L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), level=3, seed=7),
              scenes.blob_mesh(L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7), level=3, seed=11)]
    rng = np.random.default_rng(3)

    def placements(n):
        return [(k % 2, float(rng.uniform(-2.5, 2.5)), float(rng.uniform(-2.5, 2.5)), float(rng.uniform(0.7, 1.3)), float(rng.uniform(0, 6.28))) for k in range(n)]
    flats = [_grid_scene(meshes, placements(n), 0).flatten() for n in (2, 9, 3, 9)]
    for f in flats[1:]:  # same meshes in the same order: the static arrays do not change
        assert np.array_equal(f.triangles, flats[0].triangles) and np.array_equal(f.sub_nodes, flats[0].sub_nodes)
    cam = scenes.instanced_grid(W, Hh, nx=2, nz=2, level=2, sky_size=(16, 8)).camera
    ctx = U.make_ctx(gpu, flats[0], W, Hh, camera=cam, seed=3, flags=flags)
    for k, flat in enumerate(flats):
        if k:
            ctx.upload_dynamic_async(flat)
            ctx.frame_tick()
        ctx.clear()
        ctx.render(8)
        a = ctx.read_accum().copy()
        fresh = U.make_ctx(gpu, flat, W, Hh, camera=cam, seed=3, flags=flags)
        fresh.render(8)
        assert np.array_equal(a, fresh.read_accum()), f"state {k} ({len(flat.top_nodes)} top-level nodes)"
        fresh.close()
    ctx.close()


@pytest.mark.parametrize("flags_name", ["copied", "entered"])
def test_top_level_leaf_that_names_an_interior_node(gpu, flags_name):
    """A top-level leaf may name ANY node of the caller's sub-BVH array as its root (the reference just starts traversing there,
    scene.cl:141-160).  The meshes' trees are packed root by root at pt_upload_static; a leaf that names an interior node makes that
    node a root of its own at the next upload (the static part is converted again, every dynamic set refreshes its copy).  Hits against
    the oracle on the same arrays: the instance now shows only the part of the mesh below that node."""
    flags = 0 if flags_name == "copied" else gpu.FLAG_NO_BAKED_INSTANCES
    b = scenes.instanced_grid(W, Hh, nx=2, nz=1, level=3, sky_size=(16, 8))
    flat = b.flat
    ctx = U.make_ctx(gpu, b, W, Hh, seed=1, flags=flags)
    o, d = U.random_rays(30000, 8, (-3, 0.05, -3), (3, 3, 3))
    U.compare_hits(flat, ctx.intersect(o, d), O.intersect_batch(U.oracle_scene(b), o, d, threads=8), edge_flip_frac=5e-4)
    import copy
    part = copy.copy(flat)
    part.top_nodes = flat.top_nodes.copy()
    leaves = np.flatnonzero(part.top_nodes["isLeaf"] != 0)
    big = [int(l) for l in leaves if flat.sub_nodes["count"][int(part.top_nodes["a"][l])] == 0]  # instances whose root is an inner node
    assert big
    leaf = big[-1]
    root = int(part.top_nodes["a"][leaf])
    child = int(flat.sub_nodes["left"][root]) + 1  # the right child of the mesh root: an interior node (or a leaf) of the mesh tree
    part.top_nodes["a"][leaf] = child
    ctx.upload_dynamic(part)
    got, want = ctx.intersect(o, d), O.intersect_batch(O.BoundScene(part, sky=b.sky), o, d, threads=8)
    info = U.compare_hits(part, got, want, edge_flip_frac=5e-4)
    assert info["n"] > 5000
    whole = O.intersect_batch(U.oracle_scene(b), o, d, threads=8)
    assert (want["prim"] != whole["prim"]).mean() > 0.005, "the modified instance must show less of its mesh"
    ctx.close()


@pytest.mark.parametrize("route", ["host_nodes", "device"])
def test_refit_of_a_mesh_with_leaves_larger_than_a_device_leaf(gpu, route):
    """100 coincident triangles make one 100-triangle leaf, which pt_upload_static splits into a small subtree of its own (device leaves hold
    <= 30): those pair nodes mirror no node of the caller's, so a refit (pt_update_geometry) recomputes their boxes from the moved triangles.
    After the refit the context must see what a fresh context sees on the same arrays."""
    pos = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (100, 1))
    pos[:, 2] = np.repeat(np.arange(100) * 1e-4, 3)
    mesh = H.Mesh(pos, np.arange(300, dtype=np.uint32).reshape(100, 3), [L.material_diffuse((1, 1, 1))], builder=H.BVH_BINNED_SAH)
    sc = H.Scene()
    sc.add_node(mesh)
    flat = sc.flatten()
    assert int(flat.sub_nodes["count"].max()) > 30
    ctx = gpu.Context(8, 8)
    ctx.upload_scene(flat)
    rng = np.random.default_rng(3)
    o = np.c_[rng.uniform(-0.5, 3.5, (4000, 2)), np.full(4000, -1.0)].astype(np.float32)
    d = np.tile(np.array([[0, 0, 1]], np.float32), (4000, 1))
    before = ctx.intersect(o, d)
    moved = pos.copy()
    moved[:, 0] = pos[:, 0] * 0.5 + 2.0  # the stack of triangles shrinks and slides to x in [2, 2.5]
    mesh.refit(moved)
    flat2 = sc.flatten()
    if route == "device":  # the device refits the sub-leaves of the split leaf from their triangles like any other leaf
        ctx.refit_vertices(sc.mesh_offsets(mesh)[0], mesh.vertices_view())
    else:
        ctx.update_geometry(flat2)
    ctx.upload_dynamic(flat2)
    got = ctx.intersect(o, d)
    fresh = gpu.Context(8, 8)
    fresh.upload_scene(flat2)
    want = fresh.intersect(o, d)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got[k], want[k]), k
    ref = O.intersect_batch(O.BoundScene(flat2), o, d)
    assert np.array_equal(got["prim"] >= 0, ref["prim"] >= 0)
    assert (got["prim"] >= 0).sum() > 50 and not np.array_equal(before["prim"] >= 0, got["prim"] >= 0)
    ctx.close()
    fresh.close()


def _deformed(p0, k):
    ang = 0.7 * k * p0[:, 1]
    return np.stack([np.cos(ang) * p0[:, 0] - np.sin(ang) * p0[:, 2], (1 - 0.15 * k) * p0[:, 1], np.sin(ang) * p0[:, 0] + np.cos(ang) * p0[:, 2]], 1).astype(np.float32)


def test_refit_before_the_first_dynamic_upload_and_rebuild_after_a_refit(gpu):
    """The two corners of pt_update_geometry's bookkeeping.  (1) A refit BEFORE anything of the scene is on the device (pt_upload_static, then
    pt_update_geometry, then the first pt_upload_dynamic): everything is refitted on the host and uploaded once.  (2) A refit on the device
    (k_refit_nodes / k_refit_tris: the host's mirrors of the converted arrays go stale), then a state whose top-level leaf names an interior
    node, which makes the whole conversion run again -- from the caller's LATEST arrays, which live in the refit's staging memory.  Either way
    the context must hold what a fresh context holds that only ever saw the final arrays."""
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    v, f = scenes.icosphere(3)
    p0 = (v * 0.5).astype(np.float32)
    blob = H.Mesh(p0, f.astype(np.uint32), [mat], builder=H.BVH_SPATIAL_SPLIT)
    scene = H.Scene()
    mb = scenes._MeshBuilder()
    mats = scenes._room_materials()
    scenes._room(mb, mats)
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    scene.add_node(blob, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
    flat0 = scene.flatten()
    cam = scenes.blob_room(W, Hh, level=3).camera
    o, d = U.random_rays(20000, 4, (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9))

    def fresh_hits(flat):
        c = U.make_ctx(gpu, flat, W, Hh, camera=cam, seed=8)
        h = c.intersect(o, d)
        c.render(8)
        a = c.read_accum().copy()
        c.close()
        return h, a

    # (1) refit before the first dynamic upload
    ctx = gpu.Context(W, Hh, seed=8)
    ctx._chk(gpu.lib().pt_upload_static(ctx._h, flat0.vertices.ctypes.data, len(flat0.vertices), flat0.triangles.ctypes.data, len(flat0.triangles),
                                        flat0.materials.ctypes.data, len(flat0.materials), flat0.sub_nodes.ctypes.data, len(flat0.sub_nodes)), "pt_upload_static")
    blob.refit(_deformed(p0, 1))
    flat1 = scene.flatten()
    ctx.update_geometry(flat1)
    ctx.upload_dynamic(flat1)
    ctx.set_camera(cam)
    want_h, want_a = fresh_hits(flat1)
    got_h = ctx.intersect(o, d)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got_h[k], want_h[k]), ("before the first upload", k)
    ctx.render(8)
    assert np.array_equal(ctx.read_accum(), want_a)
    # (2) a device-side refit, then a rebuild of the static part from the latest arrays
    blob.refit(_deformed(p0, 2))
    flat2 = scene.flatten()
    ctx.update_geometry(flat2)
    ctx.upload_dynamic(flat2)
    import copy
    part = copy.copy(flat2)
    part.top_nodes = flat2.top_nodes.copy()
    leaves = [int(l) for l in np.flatnonzero(part.top_nodes["isLeaf"] != 0) if flat2.sub_nodes["count"][int(part.top_nodes["a"][l])] == 0]
    leaf = max(leaves, key=lambda l: int(part.top_nodes["a"][l]))  # the blob's instance
    part.top_nodes["a"][leaf] = int(flat2.sub_nodes["left"][int(part.top_nodes["a"][leaf])]) + 1  # its root's right child: an interior node
    ctx.upload_dynamic(part)  # a new root: the conversion runs again
    want_h, want_a = fresh_hits(part)
    got_h = ctx.intersect(o, d)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got_h[k], want_h[k]), ("rebuild after a refit", k)
    ctx.clear()
    ctx.render(8)
    assert np.array_equal(ctx.read_accum(), want_a)
    # ... and refits keep working on the re-converted arrays
    blob.refit(_deformed(p0, 3))
    flat3 = scene.flatten()
    part3 = copy.copy(flat3)
    part3.top_nodes = flat3.top_nodes.copy()
    part3.top_nodes["a"][leaf] = part.top_nodes["a"][leaf]
    ctx.update_geometry(part3)
    ctx.upload_dynamic(part3)
    want_h, _ = fresh_hits(part3)
    got_h = ctx.intersect(o, d)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got_h[k], want_h[k]), ("refit after the rebuild", k)
    ctx.close()


def test_device_refit_then_a_rebuild_of_the_static_part_and_partial_ranges(gpu):
    """pt_refit_vertices leaves the host's mirrors behind twice over: the converted arrays AND the caller's node boxes (nobody hands nodes in).
    (1) Two meshes in one scene, only the second one refitted -- a vertex RANGE, not the whole array; (2) then a state whose top-level leaf names
    an interior node, which makes the whole conversion run again: from the latest vertices, with the node boxes recomputed on the host.  Either
    way the context must hold what a fresh context holds that was given the final, host-refitted arrays; (3) a range that does not fit is refused."""
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    v, f = scenes.icosphere(3)
    p0 = (v * 0.5).astype(np.float32)
    a_mesh = H.Mesh(p0 * 0.6, f.astype(np.uint32), [mat], builder=H.BVH_BINNED_SAH)
    blob = H.Mesh(p0, f.astype(np.uint32), [mat], builder=H.BVH_SPATIAL_SPLIT)
    scene = H.Scene()
    mb = scenes._MeshBuilder()
    mats = scenes._room_materials()
    scenes._room(mb, mats)
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    scene.add_node(a_mesh, location=(-0.5, 0.5, -0.3))
    scene.add_node(blob, location=(0.2, 0.8, 0.1), scale=(1.1, 1.1, 1.1))
    flat0 = scene.flatten()
    cam = scenes.blob_room(W, Hh, level=3).camera
    o, d = U.random_rays(20000, 4, (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9))

    def fresh_hits(flat):
        c = U.make_ctx(gpu, flat, W, Hh, camera=cam, seed=8)
        h = c.intersect(o, d)
        c.close()
        return h

    ctx = U.make_ctx(gpu, flat0, W, Hh, camera=cam, seed=8)
    first_vertex, first_node = scene.mesh_offsets(blob)
    assert first_vertex > 0 and first_vertex + len(p0) == len(flat0.vertices)
    with pytest.raises(gpu.PtError, match="not a range"):
        ctx.refit_vertices(first_vertex + 1, blob.vertices_view())
    for k in (1, 2):
        blob.refit(_deformed(p0, k))
        ctx.refit_vertices(first_vertex, blob.vertices_view())
        ctx.upload_dynamic(scene.flatten_dynamic_only()[0])
        flat = scene.flatten()
        got, want = ctx.intersect(o, d), fresh_hits(flat)
        for key in ("t", "u", "v", "prim", "inst"):
            assert np.array_equal(got[key], want[key]), (k, key)
    # (2) a new root below the blob's root: the whole conversion runs again, from the host's (stale, then refreshed) mirrors
    import copy
    part = copy.copy(flat)
    part.top_nodes = flat.top_nodes.copy()
    leaf = [int(l) for l in np.flatnonzero(part.top_nodes["isLeaf"] != 0) if int(part.top_nodes["a"][l]) == first_node][0]
    part.top_nodes["a"][leaf] = int(flat.sub_nodes["left"][first_node]) + 1
    ctx.upload_dynamic(part)
    got, want = ctx.intersect(o, d), fresh_hits(part)
    for key in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got[key], want[key]), ("rebuild after a device refit", key)
    ctx.close()


def test_a_rebuilt_tree_per_frame_is_adopted_at_the_tick_while_the_old_one_renders(gpu):
    """pt_upload_static_async (round 5; the other branch of MeshSequence::buildBvh, reference src/model/mesh_sequence.cpp:89-96: a NEW tree per frame):
    a rebuilt scene is converted and copied beside the one that renders; frames enqueued before the tick still show the old scene (bit for bit what a
    context that only ever saw it renders), frames after it the new one; closest hits equal the oracle's on the rebuilt arrays.  Three rebuilds in a
    row use both static sets in turn; a refit (device route) of the adopted scene and an async rebuild that is replaced before its tick work too."""
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    mb = scenes._MeshBuilder()
    mats = scenes._room_materials()
    scenes._room(mb, mats)
    room = mb.build(mats, H.BVH_BINNED_SAH)
    cam = scenes.blob_room(W, Hh, level=3).camera
    o, d = U.random_rays(20000, 4, (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9))

    def scene_of(level, k):
        v, f = scenes.icosphere(level)
        p = (v * (0.35 + 0.05 * k) * (1.0 + 0.15 * np.sin(3.0 * k + 5.0 * v[:, :1]))).astype(np.float32)
        mesh = H.Mesh(p, f.astype(np.uint32), [mat], builder=H.BVH_BINNED_FAST)
        sc = H.Scene()
        sc.add_node(room)
        sc.add_node(mesh, location=(0.05 * k, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
        return sc, mesh, sc.flatten()

    def fresh(flat):
        c = U.make_ctx(gpu, flat, W, Hh, camera=cam, seed=8, samples_in_flight=1)
        h = c.intersect(o, d)
        c.render(8)
        a = c.read_accum().copy()
        c.close()
        return h, a

    sc0, _, flat0 = scene_of(3, 0)
    ctx = U.make_ctx(gpu, flat0, W, Hh, camera=cam, seed=8, samples_in_flight=1)
    _, a_prev = fresh(flat0)
    for k, level in ((1, 3), (2, 2), (3, 3)):  # different triangle counts: the sets' buffers grow and are reused
        sck, meshk, flatk = scene_of(level, k)
        ctx.upload_static_async(flatk)
        ctx.upload_dynamic_async(flatk)
        ctx.clear()
        ctx.render(8, sync=False)  # enqueued BEFORE the tick: the old scene
        old = ctx.read_accum().copy()
        assert np.array_equal(old, a_prev), f"rebuild {k}: a frame before the tick must show the old scene"
        ctx.frame_tick()
        ctx.clear()
        ctx.render(8)
        want_h, want_a = fresh(flatk)
        assert np.array_equal(ctx.read_accum(), want_a), f"rebuild {k}: frames after the tick != a fresh context on the rebuilt arrays"
        got_h = ctx.intersect(o, d)
        for key in ("t", "u", "v", "prim", "inst"):
            assert np.array_equal(got_h[key], want_h[key]), (k, key)
        info = U.compare_hits(flatk, got_h, O.intersect_batch(O.BoundScene(flatk), o, d, threads=8), edge_flip_frac=5e-4)
        assert info["n"] > 5000
        a_prev = want_a
    # the adopted scene deforms: the device-route refit addresses it
    verts = meshk.geometry()[0]["vertex"][:, :3]
    meshk.refit((verts * np.float32(0.9)).astype(np.float32))
    ctx.refit_vertices(sck.mesh_offsets(meshk)[0], meshk.vertices_view())
    ctx.upload_dynamic(sck.flatten_dynamic_only()[0])
    flat_r = sck.flatten()
    got_h, (want_h, _) = ctx.intersect(o, d), fresh(flat_r)
    for key in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got_h[key], want_h[key]), ("refit of the adopted scene", key)
    # a rebuild that is replaced by another one before its tick: only the last one is adopted
    _, _, flat_a = scene_of(3, 7)
    _, _, flat_b = scene_of(2, 8)
    ctx.upload_static_async(flat_a)
    ctx.upload_dynamic_async(flat_a)
    ctx.upload_static_async(flat_b)
    ctx.upload_dynamic_async(flat_b)
    # a refit between a rebuild and its tick would address the scene that still renders with the rebuilt scene's offsets: refused, nothing changes
    with pytest.raises(gpu.PtError, match="waiting for pt_frame_tick"):
        ctx.refit_vertices(sck.mesh_offsets(meshk)[0], meshk.vertices_view())
    with pytest.raises(gpu.PtError, match="waiting for pt_frame_tick"):
        ctx.update_geometry(flat_r)
    ctx.frame_tick()
    ctx.clear()
    ctx.render(8)
    assert np.array_equal(ctx.read_accum(), fresh(flat_b)[1])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("records", ["device", "host"])
def test_a_rebuilt_scene_of_every_kind_of_tree_equals_a_fresh_context(gpu, records, monkeypatch):
    """The records of a rebuilt scene (pt_upload_static_async) are made ON THE DEVICE from the caller's arrays -- k_refit_nodes: exact boxes and quantised planes,
    k_refit_tris: intersection and shading records -- where pt_upload_static makes them on the host (PTAMD_HOST_RECORDS=1: there too).  Both must be the bytes a
    fresh context holds: images and hits bit-identical, on the kinds of trees the conversion treats differently -- SBVH meshes with duplicated references and
    three-triangle leaves (cut into 1 + 2 by appended pair nodes, whose boxes travel as `extra`), single-leaf meshes (the quads), instances copied to world
    space and entered, textured materials of five types.  Afterwards the adopted scene takes a host-refitted update (pt_update_geometry) like any other."""
    if records == "host":
        monkeypatch.setenv("PTAMD_HOST_RECORDS", "1")
    start = scenes.cornell_box(W, Hh)
    o, d = U.random_rays(20000, 11, (-3.0, 0.05, -3.0), (3.0, 2.0, 3.0))
    ctx = U.make_ctx(gpu, start, W, Hh, seed=5, samples_in_flight=1)
    ctx.render(2)
    sky = tex = None

    class Quads:  # nothing but single-leaf meshes: no packed node at all (k_refit_nodes has nothing to do, the leaves hang off the top level)
        sky = material_textures = None
        camera = start.camera
        mb = scenes._MeshBuilder()
        mats = scenes._room_materials()
        scenes._room(mb, mats)
        _scene = H.Scene()
        _v = np.array([[-0.5, 0.0, -0.5], [0.5, 0.0, -0.5], [0.5, 0.0, 0.5], [-0.5, 0.0, 0.5]], np.float32)
        for _k, _y in enumerate((0.4, 0.9, 1.4)):
            _scene.add_node(H.Mesh(_v * (1.0 - 0.2 * _k), np.array([[0, 1, 2], [0, 2, 3]], np.uint32), [L.material_diffuse((0.8, 0.5 + 0.2 * _k, 0.3))],
                                   builder=H.BVH_BINNED_SAH), location=(0.0, _y, 0.0))
        _scene.add_node(H.Mesh(_v * 0.4, np.array([[0, 2, 1], [0, 3, 2]], np.uint32), [L.material_emissive((1.0, 0.9, 0.8), 12.0)], builder=H.BVH_BINNED_SAH),
                        location=(0.0, 1.9, 0.0))
        flat = _scene.flatten()

    for name, bundle, flags in (("single-leaf meshes only", Quads, 0), ("grid of SBVH meshes, copied", scenes.instanced_grid(W, Hh, nx=3, nz=2, level=3), 0),
                                ("five material types on one mesh", scenes.mixed_material_room(W, Hh, level=3), 0),
                                ("crowd, turned instances", scenes.instanced_crowd(W, Hh, nx=3, nz=2, level=2), 0)):
        ctx.upload_static_async(bundle.flat)
        ctx.upload_dynamic_async(bundle.flat)
        ctx.frame_tick()
        ctx.set_camera(bundle.camera)
        if bundle.sky is not None:  # (textures belong to the context, not to a scene: the fresh context gets whatever this one holds by now)
            sky = bundle.sky
            ctx.upload_texture(1, sky)
        if bundle.material_textures is not None:
            tex = bundle.material_textures
            ctx.upload_texture(0, tex)
        ctx.clear()
        ctx.render(4)
        got_a, got_h = ctx.read_accum().copy(), ctx.intersect(o, d)
        c2 = gpu.Context(W, Hh, seed=5, samples_in_flight=1, flags=flags)
        c2.upload_scene(bundle.flat, sky=sky, material_textures=tex)
        c2.set_camera(bundle.camera)
        c2.render(4)
        want_a, want_h = c2.read_accum().copy(), c2.intersect(o, d)
        c2.close()
        for key in ("t", "u", "v", "prim", "inst"):
            assert np.array_equal(got_h[key], want_h[key]), (name, key)
        assert np.array_equal(got_a, want_a), name
    ctx.close()
