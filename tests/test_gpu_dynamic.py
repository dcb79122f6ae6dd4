"""Dynamic scenes (reference src/raytracer.cpp:183-189,497-595, src/bvh/refit_bvh.cpp, src/model/mesh_sequence.cpp:81-97):
the double-buffered dynamic part of the scene -- converted on the host and copied on a copy stream while frames keep rendering, adopted
by pt_frame_tick without a host-side wait -- and deforming meshes through refitted BVHs."""
import numpy as np
import pytest

import gpu_util as U
import orclib as O
from ptamd import host as H, layout as L, scenes

pytestmark = pytest.mark.gpu
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


def test_asynchronous_uploads_render_every_state_like_a_fresh_context(gpu):
    """A frame loop that never waits: render state k (asynchronous), convert + copy state k+1 into the inactive buffers while it
    renders, tick, render k+1, ...  Each frame's accumulator is copied out in stream order by a torch op on the same stream.
    Every frame equals the render of a fresh context that only ever saw that state (counter PRNG: bit for bit) -- no frame
    reads a half-written or a too-new set although nothing synchronises with the host in between."""
    import torch
    b, flats = _states(5)
    s = torch.cuda.Stream()
    frames = []
    with torch.cuda.stream(s):
        acc = torch.zeros(W * Hh, 4, device="cuda")
        ctx = gpu.Context(W, Hh, seed=5)
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
    ctx.close()
    for k, flat in enumerate(flats):
        fresh = gpu.Context(W, Hh, seed=5)
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


@pytest.mark.parametrize("builder", ["BVH_BINNED_SAH", "BVH_SPATIAL_SPLIT"])
def test_deforming_mesh_through_refit(gpu, builder):
    """Mesh.refit + pt_update_geometry + pt_upload_dynamic: a blob in the room twists and squashes over three frames, its BVH
    refitted, never rebuilt.  Every frame: closest hits equal the oracle's on the refitted arrays, the image equals the oracle's
    path by path, and a context created from scratch with the same (refitted) arrays renders the same bits."""
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
        assert np.isclose(a, ref[:, :3], rtol=1e-3, atol=1e-3 * ref.max()).all(axis=1).mean() > 0.97
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
