"""Host-side scene library: BVH builder invariants (the reference's BvhTester checks,
src/bvh/bvh_test.cpp:117-139, as real tests), top-level BVH, flattening with index rebasing
(src/raytracer.cpp:244-270) and camera derivation (src/camera.cpp:18-58)."""
import os
import sys
import time

import numpy as np
import pytest

from ptamd import host as H, layout as L, scenes


@pytest.mark.parametrize("builder", [H.BVH_BINNED_SAH, H.BVH_BINNED_FAST, H.BVH_SPATIAL_SPLIT])
@pytest.mark.parametrize("level", [0, 3, 5])
def test_bvh_invariants_on_blob(builder, level):
    m = scenes.blob_mesh(L.material_diffuse((0.8, 0.8, 0.8)), level=level, builder=builder)
    s = m.stats()
    assert s["children_inside_parents"] and s["triangles_inside_leaves"] and s["all_triangles_referenced"]
    assert s["num_input_triangles"] == 20 * 4 ** level
    assert s["reachable_triangle_refs"] == s["num_triangle_refs"] >= s["num_input_triangles"]
    if builder != H.BVH_SPATIAL_SPLIT:
        assert s["num_triangle_refs"] == s["num_input_triangles"]  # object splits never duplicate
    assert s["max_depth"] <= 60
    assert s["reachable_nodes"] == s["num_nodes"] - 1  # node 1 is the pad next to the root
    nodes, tris, orig = m.bvh()
    inner = nodes[nodes["count"] == 0]
    inner = inner[inner["left"] != 0]
    assert np.all(inner["left"] % 2 == 0), "sibling pairs are 2-aligned"
    assert sorted(set(orig.tolist())) == list(range(s["num_input_triangles"]))


def test_bvh_random_soup_with_degenerates():
    rng = np.random.default_rng(3)
    n = 500
    c = rng.uniform(-1, 1, (n, 1, 3))
    pos = (c + rng.normal(scale=0.05, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    pos[:9] = pos[0]  # zero-area triangles
    pos[30:60, 2] = 0.25  # coplanar cluster (zero extent on one axis)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    for b in (H.BVH_BINNED_SAH, H.BVH_BINNED_FAST, H.BVH_SPATIAL_SPLIT):
        s = H.Mesh(pos, idx, [L.material_diffuse((1, 1, 1))], builder=b).stats()
        assert s["children_inside_parents"] and s["triangles_inside_leaves"] and s["all_triangles_referenced"]


def test_identical_triangles_do_not_recurse_forever():
    pos = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (100, 1))
    idx = np.arange(300, dtype=np.uint32).reshape(100, 3)
    s = H.Mesh(pos, idx, [L.material_diffuse((1, 1, 1))], builder=H.BVH_SPATIAL_SPLIT).stats()
    assert s["all_triangles_referenced"] and s["max_leaf_size"] == 100  # no SAH gain: one big leaf, as the reference


def test_empty_mesh_and_bad_indices_are_errors():
    with pytest.raises(RuntimeError):
        H.Mesh(np.zeros((3, 3), np.float32), np.zeros((0, 3), np.uint32), [L.material_diffuse((1, 1, 1))])
    with pytest.raises(RuntimeError, match="out of range"):
        H.Mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32), [L.material_diffuse((1, 1, 1))])


@pytest.mark.parametrize("builder", [H.BVH_BINNED_SAH, H.BVH_BINNED_FAST, H.BVH_SPATIAL_SPLIT])
def test_the_worker_pool_builds_the_sequential_builders_arrays(builder, monkeypatch):
    """Meshes of >= 8 192 triangles are built by the worker pool -- the top of the tree (with every worker filling object-split bins of its own),
    then the subtrees below a twelfth of the references one per task -- and put together in the order the sequential build allocates: the node,
    triangle and original-index arrays must be the same bytes as with PTAMD_BUILD_THREADS=1."""
    v, f = scenes.icosphere(5)  # 20 480 triangles
    p = (v * 0.5 * (1.0 + 0.1 * np.sin(7.0 * v[:, :1]))).astype(np.float32)
    f = f.astype(np.uint32)
    mat = [L.material_diffuse((0.8, 0.8, 0.8))]
    pooled = H.Mesh(p, f, mat, builder=builder).bvh()
    monkeypatch.setenv("PTAMD_BUILD_THREADS", "1")
    alone = H.Mesh(p, f, mat, builder=builder).bvh()
    for a, b in zip(pooled, alone):
        assert a.shape == b.shape and a.tobytes() == b.tobytes()


def test_the_builders_that_work_in_place_make_the_trees_of_the_list_based_ones():
    """tests/golden/builder_trees.json: digests of (nodes, triangles, original triangle) written by the builders of commit b2849b6 -- one list of references
    per node, every one of the 31 planes swept.  The object-split builders now partition ONE array in place (keeping the order on both sides) and sweep the
    occupied bins only; the spatial-split builder shares the sweep.  Same bytes, whatever the thread count (the reference's in-place builder:
    src/bvh/bvh_build.cpp:195-215, bins: src/bvh/bvh_object_split.cpp:26-70)."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_builder_trees
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "builder_trees.json")) as fh:
        want = json.load(fh)
    assert make_builder_trees.trees() == want


@pytest.mark.parametrize("env", [{"PTAMD_HOST_PIN": "l3all", "PTAMD_HOST_THREADS": "3"}, {"PTAMD_HOST_PIN": "numa", "PTAMD_HOST_SPIN_US": "0"},
                                 {"PTAMD_HOST_THREADS": "16", "PTAMD_HOST_SPIN_US": "2000"}])
def test_the_worker_pools_knobs_change_no_tree(env):
    """PTAMD_HOST_THREADS / PTAMD_HOST_SPIN_US / PTAMD_HOST_PIN (size of the pool, how long an idle worker polls, where its threads may run -- read once, when
    the pool starts: a fresh process per setting): the same golden digests whatever they say, also where /sys has no cache or node files to pin by."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import make_builder_trees as m; "
            "print(json.dumps(m.trees(levels=(4, 5))))" % os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    got = json.loads(r.stdout.strip().splitlines()[-1])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "builder_trees.json")) as fh:
        want = json.load(fh)
    assert got and all(want[k] == v for k, v in got.items())


def test_a_forked_child_builds_without_the_parents_worker_threads():
    """fork() copies the calling thread only: a child of a process whose worker pool has started must not wait for workers that are not there --
    neither in its loops nor at exit (glibc's pthread_cond_destroy waits for the waiters a copied condition variable still counts)."""
    v, f = scenes.icosphere(5)
    p = (v * 0.5).astype(np.float32)
    f = f.astype(np.uint32)
    mat = [L.material_diffuse((0.8, 0.8, 0.8))]
    here = H.Mesh(p, f, mat, builder=H.BVH_BINNED_FAST).bvh()  # (the pool is running now)
    pid = os.fork()
    if pid == 0:
        try:
            there = H.Mesh(p, f, mat, builder=H.BVH_BINNED_FAST).bvh()
            same = all(a.tobytes() == b.tobytes() for a, b in zip(here, there))
        except BaseException:
            os._exit(4)
        sys.stdout.flush()
        os.closerange(0, 3)  # (pytest's capture files: not this process's to flush)
        import ctypes
        ctypes.CDLL(None).exit(0 if same else 3)  # exit(3) of the C library: the host library's static destructors run -- they must not wait for the workers either
    for _ in range(600):  # (a deadlocked child would hang the suite: poll with a deadline)
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        pytest.fail("the forked child did not finish its build within 30 s")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_flatten_rebases_indices_and_shares_instanced_meshes():
    b = scenes.instanced_grid(64, 36, nx=3, nz=2, level=2, sky_size=(16, 8))
    f = b.flat
    # 4 unique meshes (ground, light, 2 blobs) but 8 instances
    assert f.num_instances == 8 and (f.top_nodes["isLeaf"] != 0).sum() == 8
    assert len(f.top_nodes) == 2 * 8 - 1 and f.top_root == len(f.top_nodes) - 1  # root is the last node
    assert f.triangles["indices"].max() < len(f.vertices) and f.triangles["materialIndex"].max() < len(f.materials)
    leaves = f.sub_nodes[f.sub_nodes["count"] != 0]
    assert (leaves["left"].astype(np.int64) + leaves["count"]).max() <= len(f.triangles)
    roots = f.top_nodes["a"][f.top_nodes["isLeaf"] != 0]
    assert len(set(roots.tolist())) == 4, "instances of one mesh share its sub-BVH"
    assert f.instanced_triangles == 2 + 2 + 3 * 320 + 3 * 320
    # every top-level inner box encloses its children; leaves carry inverse(world)
    for n in f.top_nodes[f.top_nodes["isLeaf"] == 0]:
        for c in (f.top_nodes[n["a"]], f.top_nodes[n["b"]]):
            assert np.all(n["min"][:3] <= c["min"][:3]) and np.all(n["max"][:3] >= c["max"][:3])
    leaf = f.top_nodes[f.top_nodes["isLeaf"] != 0][2]
    m = leaf["invTransform"].reshape(4, 4).T  # column-major
    assert abs(np.linalg.det(m[:3, :3])) > 0 and np.allclose(m[3], [0, 0, 0, 1])
    # lights are world-space: the emissive quad was placed at y = 4
    assert len(f.lights) == 2 and np.allclose(f.lights["vertices"][:, :, 1], 4.0)


def test_inverse_transform_round_trip():
    sc = H.Scene()
    m = scenes.blob_mesh(L.material_diffuse((1, 1, 1)), level=1)
    q = np.array([np.cos(0.3), 0, np.sin(0.3), 0], np.float32)
    sc.add_node(m, location=(1, 2, 3), orientation_wxyz=q, scale=(2, 2, 2))
    f = sc.flatten()
    inv = f.top_nodes[0]["invTransform"].reshape(4, 4).T.astype(np.float64)
    c, s = np.cos(0.6), np.sin(0.6)
    world = np.array([[2 * c, 0, 2 * s, 1], [0, 2, 0, 2], [-2 * s, 0, 2 * c, 3], [0, 0, 0, 1]])
    assert np.allclose(inv @ world, np.eye(4), atol=1e-5)


def test_camera_data_closed_form():
    cam = H.camera_data((0, 0, 0), (1, 0, 0, 0), 90.0, 2.0, focal_distance=1.0, thin_lens=True)
    f = 0.05
    proj = 1.0 / (1.0 / f - 1.0)
    half_w = np.tan(np.radians(45.0)) * proj
    assert np.allclose(cam["u"][:3], [2 * half_w, 0, 0], rtol=1e-6) and np.allclose(cam["v"][:3], [0, -half_w, 0], rtol=1e-6)
    assert np.allclose(cam["screenPoint"][:3], [-half_w, half_w / 2, proj], rtol=1e-6)
    assert np.isclose(cam["apertureRadius"], f / 8.0 / 2.0) and cam["thinLensEnabled"] == 1
    assert np.isclose(cam["relativeAperture"], 8.0) and np.isclose(cam["shutterTime"], 1 / 32) and np.isclose(cam["ISO"], 1200)
    assert abs(np.dot(cam["u"][:3], cam["v"][:3])) < 1e-4  # assert of src/camera.cpp:52


def test_cornell_has_36_triangles_and_inward_normals():
    f = scenes.cornell_box(64, 64).flat
    assert len(f.triangles) == 36 and len(f.lights) == 2 and len(f.top_nodes) == 1
    centre = np.array([0, 1, 0], np.float32)
    for t in f.triangles[:10]:  # the 5 walls: geometric normal faces the room centre
        p = f.vertices["vertex"][t["indices"], :3]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        assert np.dot(n, centre - p[0]) > 0


def test_bvh_cache_file_is_the_reference_format_and_is_validated(tmp_path):
    """Mesh::storeBvh / loadBvh (reference src/model/mesh.cpp:202-263): u32 version 1, u32 root, u32 numNodes,
    48-B nodes, u32 numTriangles, 16-B triangles, one trailing newline.  Round trip is exact; a file that is
    truncated, of another version, for another mesh, or structurally broken is ignored and rebuilt."""
    import struct
    def blob(seed):
        v, f = scenes.icosphere(3)
        return (v * (1.0 + 0.25 * np.sin(4.0 * v[:, :1] + seed) * np.cos(3.0 * v[:, 1:2]))).astype(np.float32), f.astype(np.uint32)
    pos, idx = blob(3)
    mats = [L.material_diffuse((0.5, 0.5, 0.5))]
    path = tmp_path / "blob.bvh"
    a = H.Mesh(pos, idx, mats, builder=H.BVH_SPATIAL_SPLIT, bvh_cache=path)
    assert not a.bvh_from_cache and path.exists()
    nodes, tris, _ = a.bvh()
    raw = path.read_bytes()
    version, root, n_nodes = struct.unpack_from("<III", raw, 0)
    assert (version, root, n_nodes) == (1, 0, len(nodes))
    assert raw[12:12 + 48 * n_nodes] == nodes.tobytes()
    (n_tris,) = struct.unpack_from("<I", raw, 12 + 48 * n_nodes)
    assert n_tris == len(tris) and raw[16 + 48 * n_nodes:16 + 48 * n_nodes + 16 * n_tris] == tris.tobytes()
    assert raw[16 + 48 * n_nodes + 16 * n_tris:] == b"\n"
    b = H.Mesh(pos, idx, mats, builder=H.BVH_SPATIAL_SPLIT, bvh_cache=path)  # second start-up: loaded, not built
    assert b.bvh_from_cache
    nb, tb, ob = b.bvh()
    assert nb.tobytes() == nodes.tobytes() and tb.tobytes() == tris.tobytes()
    assert (idx[ob] == tb["indices"]).all()  # the reference -> input triangle map is recovered from the file
    st = b.stats()
    assert st["children_inside_parents"] and st["triangles_inside_leaves"] and st["all_triangles_referenced"]

    def rebuilt_from(corrupt):
        path.write_bytes(corrupt)
        c = H.Mesh(pos, idx, mats, builder=H.BVH_SPATIAL_SPLIT, bvh_cache=path)
        ok = not c.bvh_from_cache and c.bvh()[0].tobytes() == nodes.tobytes()
        return ok and path.read_bytes() == raw  # and the good file is back on disk

    assert rebuilt_from(raw[:len(raw) // 2])  # truncated
    assert rebuilt_from(struct.pack("<I", 2) + raw[4:])  # other format version
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 12 + 32, 0)  # root's left child = 0: a cycle
    assert rebuilt_from(bytes(bad))
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 16 + 48 * n_nodes, 0xFFFFFF)  # a vertex index this mesh does not have
    assert rebuilt_from(bytes(bad))
    pos2, _ = blob(4)  # same topology, other vertices: boxes no longer hold the triangles
    c = H.Mesh(pos2 * 1.7, idx, mats, builder=H.BVH_SPATIAL_SPLIT, bvh_cache=path)
    assert not c.bvh_from_cache


def test_obj_import_follows_the_reference_material_and_geometry_rules(tmp_path):
    """Mesh::loadFromFile / addSubMesh (reference src/model/mesh.cpp:36-200) for Wavefront OBJ: polygons are
    triangulated, lines dropped, corners welded, Ke != 0 -> Emissive(Ke) else Diffuse(Kd), the offset transform is
    baked into positions and its inverse-transpose into normals, an override material replaces the MTL."""
    (tmp_path / "box.mtl").write_text("newmtl red\nKd 0.8 0.1 0.1\nnewmtl lamp\nKd 0 0 0\nKe 2 2 1\n")
    (tmp_path / "box.obj").write_text(
        "# unit quad pair with uv + normals, one n-gon, one line element\n"
        "mtllib box.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\nv 0.5 1.5 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 0 -1\nvn 0 0 1\n"
        "usemtl red\n"
        "f 1/1/1 2/2/1 3/3/1 4/4/1\n"          # quad -> 2 triangles
        "l 1 2\n"                               # dropped
        "usemtl lamp\n"
        "f -5/1/2 -4/2/2 -3/3/2 -1/3/2 -2/4/2\n")  # pentagon with negative indices -> 3 triangles
    m = H.Mesh.from_obj(tmp_path / "box.obj", builder=H.BVH_BINNED_SAH)
    st = m.stats()
    assert st["num_input_triangles"] == 5 and st["num_vertices"] == 9  # 4 + 5 welded corners
    verts, mats = m.geometry()
    assert len(mats) == 2
    assert mats[0]["type"] == L.MAT_DIFFUSE and np.allclose(mats[0]["colour"][:3], (0.8, 0.1, 0.1))
    assert mats[1]["type"] == L.MAT_EMISSIVE and np.allclose(mats[1]["colour"][:3], np.array((2, 2, 1)) * 500.0)  # Emissive(colour): 500 lm default
    _, tris, orig = m.bvh()
    by_input = tris[np.argsort(orig)]
    assert (by_input["materialIndex"] == [0, 0, 1, 1, 1]).all()
    front = verts[by_input["indices"][0]]
    assert np.allclose(front["normal"][:, :3], (0, 0, -1)) and np.allclose(sorted(front["texCoord"][:, 0]), (0, 1, 1))
    assert np.allclose(verts["vertex"][:, 3], 1.0)
    # offset transform baked in: scale (2,1,1), rotate 90 deg about y, translate (0,0,5); normals by inverse transpose
    q = (np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4), 0.0)
    t = H.Mesh.from_obj(tmp_path / "box.obj", location=(0, 0, 5), orientation_wxyz=q, scale=(2, 1, 1), builder=H.BVH_BINNED_SAH)
    tv, _ = t.geometry()
    src = verts["vertex"][:, :3].astype(np.float64)
    want = np.stack([src[:, 2], src[:, 1], -2 * src[:, 0] + 5], 1)  # R_y(90): (x,y,z) -> (z, y, -x) after the scale
    assert np.allclose(tv["vertex"][:, :3], want, atol=1e-5)
    n = tv["normal"][:, :3] / np.linalg.norm(tv["normal"][:, :3], axis=1, keepdims=True)
    assert np.allclose(np.abs(n[:4]), (1, 0, 0), atol=1e-5)  # (0,0,-1) -> (-1,0,0)
    # override material (loadFromFile's optional argument): one material, MTL ignored, emissive list follows it
    o = H.Mesh.from_obj(tmp_path / "box.obj", material=L.material_pbr_metal((0.9, 0.6, 0.5), 0.8), builder=H.BVH_BINNED_SAH)
    assert len(o.geometry()[1]) == 1 and (o.bvh()[1]["materialIndex"] == 0).all()
    with pytest.raises(RuntimeError, match="out of range"):
        (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
        H.Mesh.from_obj(tmp_path / "bad.obj")


def _write_hdr(path, rgbe, rle, resolution=None):
    """rgbe: (h, w, 4) uint8, top row first."""
    h, w, _ = rgbe.shape
    body = bytearray()
    for row in rgbe:
        if not rle:
            body += row.tobytes()
            continue
        body += bytes([2, 2, w >> 8, w & 255])
        for comp in range(4):
            data, x = row[:, comp], 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and data[x + run] == data[x]:
                    run += 1
                if run >= 3:
                    body += bytes([128 + run, int(data[x])])
                    x += run
                else:
                    lit = min(w - x, 5)
                    body += bytes([lit]) + data[x:x + lit].tobytes()
                    x += lit
    header = b"#?RADIANCE\n# made by the test\nFORMAT=32-bit_rle_rgbe\n\n" + (resolution or f"-Y {h} +X {w}").encode() + b"\n"
    path.write_bytes(header + bytes(body))


def test_radiance_hdr_layers_are_prepared_like_the_reference_texture_array(tmp_path):
    """CLTextureArray::loadImage (reference src/opencl/texture.cpp:72-120) for .hdr files: RGBE decode (flat and
    run-length scanlines), RGBA32F with alpha 1, brightness multiplier, rows bottom-up (FreeImage order), rescale
    to the array's layer size."""
    rng = np.random.default_rng(11)
    h, w = 12, 40
    rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(120, 136, (h, w))
    rgbe[2, 5:30, :] = rgbe[2, 5, :]  # long runs for the RLE path
    rgbe[7, :, 3] = 0  # exponent 0 = black
    want = rgbe[..., :3].astype(np.float64) * np.exp2(rgbe[..., 3:4].astype(np.float64) - 136.0)
    want[rgbe[..., 3] == 0] = 0
    for rle in (False, True):
        f = tmp_path / f"sky_{int(rle)}.hdr"
        _write_hdr(f, rgbe, rle)
        got = H.load_hdr(f)
        assert got.shape == (1, h, w, 4) and (got[..., 3] == 1).all()
        assert np.allclose(got[0, ::-1, :, :3], want, rtol=1e-6)  # bottom-up rows
        assert np.allclose(H.load_hdr(f, brightness=2.5)[0, ::-1, :, :3], 2.5 * want, rtol=1e-6)
    _write_hdr(tmp_path / "flip.hdr", rgbe, True, resolution=f"+Y {h} -X {w}")  # bottom-up, right-to-left file
    assert np.allclose(H.load_hdr(tmp_path / "flip.hdr")[0, :, ::-1, :3], want, rtol=1e-6)
    # rescale: a constant picture stays constant, a smooth one keeps its mean, sizes come out as asked
    const = np.tile(np.array([64, 128, 32, 130], np.uint8), (16, 32, 1))
    _write_hdr(tmp_path / "const.hdr", const, True)
    big = H.load_hdr(tmp_path / "const.hdr", 80, 24)
    assert big.shape == (1, 24, 80, 4) and np.allclose(big[0, ..., :3], np.array([64, 128, 32]) / 64.0, rtol=1e-5)
    yy, xx = np.mgrid[0:32, 0:64]
    smooth = np.stack([128 + 100 * np.sin(xx / 10.0), 128 + 100 * np.cos(yy / 6.0), 128 + 0 * xx, 129 + 0 * xx], -1).astype(np.uint8)
    _write_hdr(tmp_path / "smooth.hdr", smooth, False)
    full, half = H.load_hdr(tmp_path / "smooth.hdr"), H.load_hdr(tmp_path / "smooth.hdr", 32, 16)
    assert abs(half[..., :3].mean() - full[..., :3].mean()) < 0.01 * full[..., :3].mean()
    assert np.abs(half[0, :, :, :3] - full[0, ::2, ::2, :3]).max() < 0.08 * full.max()
    (tmp_path / "bad.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n\x01\x02")
    with pytest.raises(RuntimeError, match="truncated"):
        H.load_hdr(tmp_path / "bad.hdr")


def _png_chunk(tag, data):
    import struct, zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _write_png(path, rows_of_bytes_per_pass, width, height, depth, colour, interlace=0, plte=None, trns=None, filters=(0, 1, 2, 3, 4), idat_split=3):
    """Hand-rolled PNG writer for what PIL cannot produce: chosen scanline filters, Adam7, split IDAT.
    rows_of_bytes_per_pass: per pass, a list of raw (unfiltered) scanlines as bytes; bpp from depth/colour."""
    import struct, zlib
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    bpp = max(1, channels * depth // 8)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    raw = bytearray()
    for rows in rows_of_bytes_per_pass:
        prev = None
        for y, row in enumerate(rows):
            f = filters[y % len(filters)]
            out = bytearray([f])
            for i, v in enumerate(row):
                a = row[i - bpp] if i >= bpp else 0
                b = prev[i] if prev is not None else 0
                c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
                pred = (0, a, b, (a + b) // 2, paeth(a, b, c))[f]
                out.append((v - pred) & 0xFF)
            raw += out
            prev = row
    comp = zlib.compress(bytes(raw), 6)
    n = max(1, len(comp) // idat_split)
    body = b"".join(_png_chunk(b"IDAT", comp[i:i + n]) for i in range(0, len(comp), n))
    ihdr = struct.pack(">IIBBBBB", width, height, depth, colour, 0, 0, interlace)
    extra = (_png_chunk(b"PLTE", plte) if plte else b"") + (_png_chunk(b"tRNS", trns) if trns else b"") + _png_chunk(b"tEXt", b"Comment\x00hand made")
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", ihdr) + extra + body + _png_chunk(b"IEND", b""))


def test_png_decoder_against_pil_and_hand_made_files(tmp_path):
    """PNG files (the only texture format the reference ships: 51 of them) decode to the same RGBA8 as PIL: every
    colour type, sub-byte and 16-bit depths, palette + tRNS, all five scanline filters, Adam7, split IDAT, CRC."""
    from PIL import Image
    rng = np.random.default_rng(21)
    h, w = 37, 53
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgba[5:9, 3:20] = rgba[5, 3]  # flat runs so that the encoder picks different filters
    cases = {"RGBA": Image.fromarray(rgba), "RGB": Image.fromarray(rgba[..., :3]), "L": Image.fromarray(rgba[..., 0]),
             "LA": Image.fromarray(rgba[..., :2]), "1": Image.fromarray((rgba[..., 0] > 127)), "I16": Image.fromarray(rgba[..., 0].astype(np.uint16) * 257)}
    pal = Image.fromarray(rgba[..., :3]).quantize(31)
    cases["P"] = pal
    for name, im in cases.items():
        f = tmp_path / f"{name}.png"
        im.save(f)
        want = np.asarray(Image.open(f).convert("RGBA"))
        if name == "I16":  # PIL maps 16-bit grey to 8 bits by clipping; PNG semantics (and FreeImage) keep the high byte
            want = np.stack([rgba[..., 0]] * 3 + [np.full((h, w), 255, np.uint8)], -1)
        got = H.load_png(f)
        assert got.shape == (h, w, 4) and np.array_equal(got, want), name
    pal.save(tmp_path / "Pt.png", transparency=bytes(rng.integers(0, 256, 31, dtype=np.uint8)))
    assert np.array_equal(H.load_png(tmp_path / "Pt.png"), np.asarray(Image.open(tmp_path / "Pt.png").convert("RGBA")))
    Image.fromarray(rgba[..., :3]).save(tmp_path / "key.png", transparency=tuple(int(v) for v in rgba[2, 2, :3]))
    got = H.load_png(tmp_path / "key.png")
    key = (rgba[..., :3] == rgba[2, 2, :3]).all(-1)
    assert np.array_equal(got[..., 3], np.where(key, 0, 255)) and np.array_equal(got[..., :3], rgba[..., :3])
    # hand-made: every filter type row by row, IDAT in pieces, an ancillary chunk to skip
    rows = [bytes(rgba[y].reshape(-1)) for y in range(h)]
    _write_png(tmp_path / "filters.png", [rows], w, h, 8, 6)
    assert np.array_equal(H.load_png(tmp_path / "filters.png"), rgba)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "filters.png").convert("RGBA")), rgba)  # the writer itself is sane
    # Adam7, RGB 8 bit and 2-bit grey
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    sub = [[bytes(rgba[y, x0::dx, :3].reshape(-1)) for y in range(y0, h, dy)] if x0 < w and y0 < h else [] for (x0, y0, dx, dy) in passes]
    _write_png(tmp_path / "adam7.png", sub, w, h, 8, 2, interlace=1)
    assert np.array_equal(H.load_png(tmp_path / "adam7.png")[..., :3], rgba[..., :3])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "adam7.png").convert("RGB")), rgba[..., :3])
    g2 = rgba[..., 0] >> 6

    def pack2(v):
        v = np.concatenate([v, np.zeros((-len(v)) % 4, np.uint8)])
        return bytes((v[0::4] << 6) | (v[1::4] << 4) | (v[2::4] << 2) | v[3::4])
    _write_png(tmp_path / "grey2i.png", [[pack2(g2[y, x0::dx]) for y in range(y0, h, dy)] for (x0, y0, dx, dy) in passes], w, h, 2, 0, interlace=1)
    assert np.array_equal(H.load_png(tmp_path / "grey2i.png")[..., 0], g2 * 85)
    # damage: a flipped byte inside IDAT breaks its CRC; a truncated file is refused
    data = bytearray(open(tmp_path / "filters.png", "rb").read())
    data[len(data) // 2] ^= 0x40
    open(tmp_path / "bad.png", "wb").write(data)
    with pytest.raises(RuntimeError, match="CRC|inflate"):
        H.load_png(tmp_path / "bad.png")
    open(tmp_path / "short.png", "wb").write(bytes(data[:200]))
    with pytest.raises(RuntimeError, match="truncated|IHDR|CRC"):
        H.load_png(tmp_path / "short.png")
    with pytest.raises(RuntimeError, match="not a PNG"):
        open(tmp_path / "no.png", "wb").write(b"GIF89a" + bytes(40))
        H.load_png(tmp_path / "no.png")


def test_material_texture_layers_are_prepared_like_the_reference_texture_array(tmp_path):
    """CLTextureArray::loadImage for the 8-bit material array (reference src/opencl/texture.cpp:84-92,112-131): rescale to
    the layer size, FreeImage_AdjustGamma(1/2.2) on the colour channels unless linear, alpha kept, rows bottom-up, and
    read_imagef's byte / 255."""
    from PIL import Image
    rng = np.random.default_rng(3)
    h, w = 16, 24
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgba[:4, :, 3] = 0  # a cut-out band (alpha-0 texels pass rays through, shading.cl:587-601)
    Image.fromarray(rgba).save(tmp_path / "t.png")
    lin = H.load_material_png(tmp_path / "t.png", is_linear=True)
    assert lin.shape == (1, h, w, 4) and np.array_equal(lin[0, ::-1], rgba.astype(np.float32) / 255.0)
    srgb = H.load_material_png(tmp_path / "t.png")
    lut = np.minimum(255, np.floor(255.0 * (np.arange(256) / 255.0) ** 2.2 + 0.5)).astype(np.uint8)
    assert np.array_equal(srgb[0, ::-1, :, :3], lut[rgba[..., :3]].astype(np.float32) / 255.0)
    assert np.array_equal(srgb[0, ::-1, :, 3], rgba[..., 3].astype(np.float32) / 255.0)
    # rescale: sizes as asked, a constant picture stays constant, values stay bytes / 255
    const = np.tile(np.array([200, 100, 50, 255], np.uint8), (8, 8, 1))
    Image.fromarray(const).save(tmp_path / "c.png")
    big = H.load_material_png(tmp_path / "c.png", 32, 16, is_linear=True)
    assert big.shape == (1, 16, 32, 4) and np.array_equal(big[0, 0, 0], np.array([200, 100, 50, 255], np.float32) / 255.0) and (big == big[0, 0, 0]).all()
    small = H.load_material_png(tmp_path / "t.png", 12, 8, is_linear=True)
    assert np.array_equal(small * 255.0, np.round(small * 255.0)) and abs(small[..., :3].mean() - lin[..., :3].mean()) < 0.03


@pytest.mark.skipif(not os.path.isdir("/root/reference/assets"), reason="the reference's assets are only present in the build container")
def test_every_png_the_reference_ships_decodes_like_pil():
    from PIL import Image
    import glob
    files = sorted(glob.glob("/root/reference/assets/**/*.png", recursive=True))
    assert len(files) >= 40
    for f in files:
        want = np.asarray(Image.open(f).convert("RGBA"))
        assert np.array_equal(H.load_png(f), want), f


def test_obj_diffuse_maps_register_texture_files_like_the_reference(tmp_path):
    """addSubMesh (reference src/model/mesh.cpp:61-66): a non-emissive material with a diffuse map becomes
    Material::Diffuse(textureArray.add(folder / map_Kd), Kd); UniqueTextureArray::add gives one id per file
    (src/opencl/texture.cpp:9-19); the array is then loaded at one fixed layer size."""
    from PIL import Image
    (tmp_path / "tex").mkdir()
    a = np.zeros((8, 8, 3), np.uint8)
    a[..., 0] = 255
    Image.fromarray(a).save(tmp_path / "tex" / "red.png")
    b = np.zeros((4, 16, 4), np.uint8)
    b[..., 1] = 128
    b[..., 3] = 255
    b[0, :, 3] = 0
    Image.fromarray(b).save(tmp_path / "green.png")
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 1 1\nmap_Kd tex\\red.png\nnewmtl b\nKd 0.5 0.5 0.5\nmap_Kd -s 1 1 1 green.png\n"
                                    "newmtl c\nKd 0.2 0.2 0.2\nnewmtl d\nmap_Kd tex/red.png\nnewmtl lamp\nKe 1 1 1\nmap_Kd green.png\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n"
                                    + "".join(f"usemtl {n}\nf 1/1 2/2 3/3\n" for n in "abcd") + "usemtl lamp\nf 1/1 2/2 3/3\n")
    tf = H.TextureFiles()
    m = H.Mesh.from_obj(tmp_path / "m.obj", builder=H.BVH_BINNED_SAH, textures=tf)
    mats = m.geometry()[1]
    assert [int(x) for x in mats["textureId"][:4]] == [0, 1, -1, 0] and mats[4]["type"] == L.MAT_EMISSIVE
    assert np.allclose(mats[1]["colour"][:3], 0.5)
    files = tf.files()
    assert [os.path.basename(p) for p, _, _ in files] == ["red.png", "green.png"] and not files[0][1]
    assert tf.add(files[0][0]) == 0 and tf.add(tmp_path / "other.png", is_linear=True) == 2  # same file, same id
    tf2 = H.TextureFiles()
    H.Mesh.from_obj(tmp_path / "m.obj", builder=H.BVH_BINNED_SAH, textures=tf2)
    layers = tf2.load(16, 8)
    assert layers.shape == (2, 8, 16, 4)
    assert np.allclose(layers[0, ..., 0], 1.0) and np.allclose(layers[0, ..., 1:3], 0.0) and np.allclose(layers[0, ..., 3], 1.0)
    g = np.floor(255.0 * (128 / 255.0) ** 2.2 + 0.5) / 255.0  # FreeImage_AdjustGamma(1 / 2.2) on the rescaled bytes
    assert np.allclose(layers[1, 0, :, 1], g) and layers[1, -1, :, 3].max() < 0.5 and np.allclose(layers[1, 0, :, 3], 1.0)  # bottom-up: the cut-out row is last
    assert H.TextureFiles().load(4, 4).shape == (1, 4, 4, 4)  # an empty registry still makes one layer
    # without a registry the map is ignored and the colour stays Kd
    assert (H.Mesh.from_obj(tmp_path / "m.obj", builder=H.BVH_BINNED_SAH).geometry()[1]["textureId"][:4] == -1).all()


@pytest.mark.skipif(not os.path.isfile("/root/reference/assets/3dmodels/plane/plane.obj"), reason="the reference's assets are only present in the build container")
def test_the_reference_light_plane_imports_as_main_cpp_uses_it():
    """main.cpp:131: Mesh(assets/3dmodels/plane/plane.obj, Material::Emissive(5500 K, 1000 lm), textureArray) -- two
    emissive triangles; without the override its `usemtl Material.001` is not in plane.mtl, i.e. the default grey."""
    f = "/root/reference/assets/3dmodels/plane/plane.obj"
    m = H.Mesh.from_obj(f, material=L.material_emissive((1.0, 0.9, 0.8), 1000.0), builder=H.BVH_BINNED_SAH)
    st = m.stats()
    assert st["num_input_triangles"] == 2 and st["num_vertices"] == 4 and st["all_triangles_referenced"]
    verts, mats = m.geometry()
    assert len(mats) == 1 and mats[0]["type"] == L.MAT_EMISSIVE
    assert np.allclose(np.abs(verts["vertex"][:, [0, 2]]), 1.0) and np.abs(verts["vertex"][:, 1]).max() < 1e-4
    sc = H.Scene()
    sc.add_node(m, location=(0.0, 3.0, 0.0))
    flat = sc.flatten()
    assert len(flat.lights) == 2 and np.allclose(flat.lights["vertices"][..., 1], 3.0, atol=1e-4)
    tf = H.TextureFiles()
    plain = H.Mesh.from_obj(f, builder=H.BVH_BINNED_SAH, textures=tf).geometry()[1]
    assert plain[0]["type"] == L.MAT_DIFFUSE and np.allclose(plain[0]["colour"][:3], 0.6) and tf.files() == []


BUNNY = "/root/reference/assets/3dmodels/stanford/bunny/bun_zipper.ply"


@pytest.mark.skipif(not os.path.isfile(BUNNY), reason="the reference's assets are only present in the build container")
@pytest.mark.parametrize("builder", [H.BVH_BINNED_SAH, H.BVH_SPATIAL_SPLIT])
def test_the_stanford_bunny_ply_loads_and_builds(builder):
    """The mesh BASELINE configs 2-5 are quoted on (bun_zipper.ply: 35 947 vertices, 69 451 faces, no normals) through the
    PLY reader, the area-weighted smooth normals and both builders the configs name, with the reference's BvhTester
    checks (src/bvh/bvh_test.cpp:117-139; SBVH leaves hold clipped boxes, so 'triangles inside leaves' is exact for the
    object-split tree only)."""
    m = H.Mesh.from_ply(BUNNY, L.material_diffuse((0.8, 0.8, 0.8)), builder=builder)
    verts, mats = m.geometry()
    nodes, tris, orig = m.bvh()
    v, n = verts["vertex"][:, :3], verts["normal"][:, :3]
    assert len(v) == 35947 and len(mats) == 1 and tris["indices"].max() == 35946
    assert sorted(set(orig.tolist())) == list(range(69451)), "every face of the file is referenced by some leaf"
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-4), "smooth normals are generated and unit length"
    lo, hi = v.min(axis=0), v.max(axis=0)
    assert np.allclose(lo, [-0.0947, 0.0330, -0.0619], atol=1e-3) and np.allclose(hi, [0.0610, 0.1873, 0.0588], atol=1e-3)
    s = m.stats()
    assert s["num_input_triangles"] == 69451 and s["children_inside_parents"] and s["all_triangles_referenced"]
    assert s["reachable_triangle_refs"] == s["num_triangle_refs"] >= 69451
    if builder == H.BVH_BINNED_SAH:
        assert s["triangles_inside_leaves"] and s["num_triangle_refs"] == 69451
    else:
        assert s["num_triangle_refs"] < 1.6 * 69451, "spatial splits duplicate a bounded share of the references"
    assert s["max_leaf_size"] <= 8 and s["max_depth"] <= 60


def test_material_layer_as_the_bgra8_bitmap_the_reference_uploads(tmp_path):
    """The 8-bit storage of the material array (CL_BGRA / CL_UNORM_INT8, src/opencl/texture.cpp:112-131,148): the same layer as
    bytes b g r a; byte / 255 with the channels swapped back is exactly the float layer (what read_imagef returns for it)."""
    from PIL import Image
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (24, 40, 4), dtype=np.uint8)
    img[5:9, :, 3] = 0
    Image.fromarray(img).save(tmp_path / "t.png")
    for w, h, lin in ((0, 0, False), (16, 32, True)):
        f = H.load_material_png(tmp_path / "t.png", w or None, h or None, is_linear=lin)
        b = H.load_material_png(tmp_path / "t.png", w or None, h or None, is_linear=lin, as_bgra8=True)
        assert b.dtype == np.uint8 and b.shape == f.shape
        assert np.array_equal(b[..., [2, 1, 0, 3]].astype(np.float32) / np.float32(255.0), f)
    tf = H.TextureFiles()
    tf.add(tmp_path / "t.png")
    assert tf.load(8, 8, as_bgra8=True).shape == (1, 8, 8, 4) and H.TextureFiles().load(8, 8, as_bgra8=True).dtype == np.uint8


@pytest.mark.parametrize("builder", [H.BVH_BINNED_SAH, H.BVH_BINNED_FAST, H.BVH_SPATIAL_SPLIT])
def test_refit_keeps_topology_and_restores_the_invariants(builder):
    """refitBVH (reference src/bvh/refit_bvh.cpp:6-34) through Mesh.refit: a deformed frame of the same mesh keeps node links,
    leaf ranges and triangle order; every box is recomputed bottom-up, so the BvhTester invariants hold again -- for spatial-split
    trees too (their references then get whole-triangle boxes) -- and a refit back to the first frame restores the first boxes of
    an object-split tree exactly."""
    v, f = scenes.icosphere(3)
    p0 = (v * 0.5).astype(np.float32)
    m = H.Mesh(p0, f.astype(np.uint32), [L.material_diffuse((0.8, 0.8, 0.8))], builder=builder)
    nodes0, tris0, orig0 = m.bvh()
    verts0, _ = m.geometry()
    # deform: twist about y and squash
    ang = 1.5 * p0[:, 1]
    p1 = np.stack([np.cos(ang) * p0[:, 0] - np.sin(ang) * p0[:, 2], 0.6 * p0[:, 1] + 0.1 * np.sin(4 * p0[:, 0]), np.sin(ang) * p0[:, 0] + np.cos(ang) * p0[:, 2]], 1)
    m.refit(p1)
    s = m.stats()
    assert s["children_inside_parents"] and s["triangles_inside_leaves"] and s["all_triangles_referenced"]
    nodes1, tris1, orig1 = m.bvh()
    verts1, _ = m.geometry()
    assert np.array_equal(nodes1["left"], nodes0["left"]) and np.array_equal(nodes1["count"], nodes0["count"])
    assert np.array_equal(tris1, tris0) and np.array_equal(orig1, orig0)
    assert np.allclose(verts1["vertex"][:, :3], p1) and not np.array_equal(nodes1["min"], nodes0["min"])
    assert np.allclose(np.linalg.norm(verts1["normal"][:, :3], axis=1), 1.0, atol=1e-4) and not np.allclose(verts1["normal"], verts0["normal"])
    # the root box is the box of the deformed vertices
    assert np.allclose(nodes1["min"][0, :3], p1.min(axis=0)) and np.allclose(nodes1["max"][0, :3], p1.max(axis=0))
    m.refit(p0, normals=verts0["normal"][:, :3])
    nodes2, _, _ = m.bvh()
    if builder != H.BVH_SPATIAL_SPLIT:
        assert np.array_equal(nodes2["min"], nodes0["min"]) and np.array_equal(nodes2["max"], nodes0["max"])
    assert np.array_equal(m.geometry()[0]["normal"], verts0["normal"])
    with pytest.raises(RuntimeError):
        m.refit(p0[:-1])


def test_flatten_dynamic_is_the_per_tick_half_of_flatten():
    """pth_scene_flatten_dynamic = flattenDynamic alone (what RayTracer::frameTick runs on the host): after a node moved it yields the
    lights and the top-level BVH a full flatten yields, and leaves the static arrays alone; a new node needs a full flatten first."""
    b = scenes.instanced_grid(64, 36, level=2, sky_size=(16, 8))
    flat = b.flat
    b.scene.set_transform(3, location=(0.1, 0.7, 0.2))
    part, _ = b.scene.flatten_dynamic(flat)
    full = b.scene.flatten()
    assert np.array_equal(part.top_nodes, full.top_nodes) and np.array_equal(part.lights, full.lights) and part.top_root == full.top_root
    assert not np.array_equal(part.top_nodes, flat.top_nodes)
    assert part.vertices is flat.vertices and part.sub_nodes is flat.sub_nodes
    b.scene.add_node(scenes.blob_mesh(L.material_diffuse((1, 1, 1)), level=1))
    with pytest.raises(RuntimeError, match="pth_scene_flatten first"):
        b.scene.flatten_dynamic(flat)


def test_flatten_dynamic_refuses_static_arrays_a_refit_left_behind():
    """after Mesh.refit the vertices and sub-BVH boxes of the last full flatten are stale, while the lights and top-level boxes
    flatten_dynamic makes follow the NEW mesh -- a mix that renders wrongly without any error.  The host library holds a generation counter
    per mesh against the ones it flattened and refuses; a full flatten makes it current again."""
    v, f = scenes.icosphere(2)
    mesh = H.Mesh((v * 0.5).astype(np.float32), f.astype(np.uint32), [L.material_diffuse((0.8, 0.8, 0.8))])
    scene = H.Scene()
    scene.add_node(mesh)
    scene.add_node(mesh, location=(2.0, 0.0, 0.0))
    flat = scene.flatten()
    scene.flatten_dynamic(flat)
    mesh.refit((v * 0.7).astype(np.float32))
    with pytest.raises(RuntimeError, match="refitted since pth_scene_flatten"):
        scene.flatten_dynamic(flat)
    flat2 = scene.flatten()
    assert not np.array_equal(flat2.vertices["vertex"], flat.vertices["vertex"])
    part, _ = scene.flatten_dynamic(flat2)
    assert np.array_equal(part.top_nodes, flat2.top_nodes)


def test_top_level_tree_of_geometrically_spaced_instances_stays_shallow():
    """instances spaced geometrically along a line make a SAH split peel ONE box off per level -- a top-level tree as deep as the
    instance count, which the device library refuses for its traversal stack (112 entries) although a balanced tree over the same boxes is
    fine.  Lop-sided splits are only taken while the depth stays within ~2 log2(n)."""
    v, f = scenes.icosphere(0)
    mesh = H.Mesh((v * 0.4).astype(np.float32), f.astype(np.uint32), [L.material_diffuse((0.8, 0.8, 0.8))], builder=H.BVH_BINNED_SAH)
    scene = H.Scene()
    n = 600
    for k in range(n):
        s = 1.03 ** k
        scene.add_node(mesh, location=(3.0 * s * s, 0.0, 0.0), scale=(s, s, s))
    flat = scene.flatten()
    top = flat.top_nodes
    assert len(top) == 2 * n - 1
    depth = np.zeros(len(top), np.int64)
    for i in range(len(top) - 1, -1, -1):  # parents are stored after their children: one reverse sweep pushes depths down
        if not top[i]["isLeaf"]:
            depth[int(top[i]["a"])] = depth[int(top[i]["b"])] = depth[i] + 1
    assert depth.max() <= 2 * (2 + 2 * int(np.log2(n))), depth.max()


@pytest.mark.parametrize("n", [300, 3000])
def test_top_level_build_for_many_instances(n):
    """More than kAgglomerativeMaxInstances (256) instances: top-down SAH over the instance boxes instead of the reference's O(n^2)
    agglomerative clustering (top_bvh_build.cpp:42-93).  The contract of the array is the reference's: every instance is a leaf reached
    exactly once from the root, children are stored before their parent and lie inside it, the root is the last node."""
    b = scenes.instance_field(64, 36, n=n, level=1)
    import time
    t0 = time.perf_counter()
    flat = b.scene.flatten()
    dt = time.perf_counter() - t0
    top = flat.top_nodes
    assert flat.top_root == len(top) - 1 and len(top) == 2 * flat.num_instances - 1
    seen, stack = np.zeros(len(top), bool), [flat.top_root]
    while stack:
        i = stack.pop()
        assert not seen[i]
        seen[i] = True
        if top[i]["isLeaf"]:
            continue
        for ch in (int(top[i]["a"]), int(top[i]["b"])):
            assert ch < i
            assert (top[ch]["min"][:3] >= top[i]["min"][:3]).all() and (top[ch]["max"][:3] <= top[i]["max"][:3]).all()
            stack.append(ch)
    assert seen.all() and int((top["isLeaf"] != 0).sum()) == flat.num_instances
    assert dt < 1.0, f"{n} instances flattened in {dt:.2f} s"


def test_loaders_survive_a_seeded_mutation_fuzz(tmp_path):
    """The parsers of untrusted bytes (PNG, Radiance .hdr, OBJ, MTL, PLY, the .bvh cache file) through 200 mutated inputs each
    (tools/fuzz_loaders.py: bit flips, interesting integers, truncation, duplicated / deleted runs, replaced tokens and lines; PNG chunk
    CRCs repaired so that mutations reach the decoder): every input yields a result or an error return, and a mesh or cache file that is
    accepted passes the BvhTester invariants.  The 10 000-input runs on the AddressSanitizer / UBSan build are recorded in DESIGN.md."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_loaders
    summary = fuzz_loaders.run(200, seed=3, tmp=str(tmp_path), verbose=False, with_reference=False)
    assert sorted(summary) == ["bvh", "hdr", "mtl", "obj", "ply", "png"]
    for name, s in summary.items():
        assert s["accepted"] + s["rejected"] == 200, name
    assert summary["png"]["rejected"] > 100 and summary["bvh"]["rejected"] > 100  # (most mutations must be noticed)


def test_headers_that_announce_more_than_the_file_holds_are_refused_at_once(tmp_path):
    """Counts in a header are untrusted: a PLY that announces four billion vertices, a Radiance picture of 65 535 x 65 535 pixels and a PNG of
    32 768 x 32 768 in a file of a few hundred bytes must cost an error message -- not gigabytes of memory or minutes of reading an
    exhausted stream (found by the fuzzer: 886 s for 1 500 PLY inputs before the counts were held against the file size)."""
    import struct
    import sys
    import time
    import zlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_loaders as F
    rng = np.random.default_rng(0)
    ply = F.make_ply(rng, True).replace(b"element vertex 42", b"element vertex 4000000000")
    assert b"4000000000" in ply
    hdr = F.make_hdr(16, 4, rng).replace(b"-Y 4 +X 16", b"-Y 65535 +X 65535")
    png = F.make_png(8, 8, 6, 8, 0, rng)
    ihdr = struct.pack(">IIBBBBB", 32768, 32768, 8, 6, 0, 0, 0)
    png = png[:8] + struct.pack(">I", 13) + b"IHDR" + ihdr + struct.pack(">I", zlib.crc32(b"IHDR" + ihdr) & 0xFFFFFFFF) + png[8 + 25:]
    for name, data in (("big.ply", ply), ("big.hdr", hdr), ("big.png", png)):
        (tmp_path / name).write_bytes(data)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="announces"):
        H.Mesh.from_ply(tmp_path / "big.ply", L.material_diffuse((1, 1, 1)))
    with pytest.raises(RuntimeError, match="truncated"):
        H.load_hdr(tmp_path / "big.hdr", 16, 8)
    with pytest.raises(RuntimeError, match="inflate"):
        H.load_png(tmp_path / "big.png")
    assert time.perf_counter() - t0 < 2.0
