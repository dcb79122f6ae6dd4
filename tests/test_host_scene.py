"""Host-side scene library: BVH builder invariants (the reference's BvhTester checks,
src/bvh/bvh_test.cpp:117-139, as real tests), top-level BVH, flattening with index rebasing
(src/raytracer.cpp:244-270) and camera derivation (src/camera.cpp:18-58)."""
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
