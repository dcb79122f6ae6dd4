"""Generates tests/golden/bunny_flat.npz: the REAL Stanford bunny of BASELINE configs 2 / 3 (/root/reference/assets/3dmodels/stanford/bunny/bun_zipper.ply,
69 451 triangles; it does not travel to the GPU box, bench.py times a seeded substitute of the same size) flattened through the host library -- binned-SAH
tree, diffuse (config 2) and SBVH, rough glass (config 3), each in the five-wall room with the area light -- as the typed arrays pt_upload_static /
pt_upload_dynamic consume, plus what the oracle finds on them: 4 096 closest hits, 4 096 occlusion verdicts and a 64 x 36 / 256-spp accumulator per builder.
Data only (vertices, indices, nodes, expected outputs): no reference source.  Build container only:   python tests/golden/make_bunny_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib as O  # noqa: E402
import gpu_util as U  # noqa: E402
from ptamd import host as H, layout as L, scenes  # noqa: E402

PLY = "/root/reference/assets/3dmodels/stanford/bunny/bun_zipper.ply"
W, Hh, SPP, NRAYS = 64, 36, 256, 4096


def bunny_room(builder, material):
    """configs 2 / 3: the bunny (bounding box ~0.16 x 0.15 x 0.12 m around (-0.02, 0.11, 0)) scaled to ~0.95 m and set on the floor of the room"""
    mats = scenes._room_materials()
    mb = scenes._MeshBuilder()
    scenes._room(mb, mats)
    scene = H.Scene()
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    bunny = H.Mesh.from_ply(PLY, material, builder=builder)
    s = 6.0
    scene.add_node(bunny, location=(0.1, -0.033 * s, 0.1), scale=(s, s, s))
    cam = scenes._camera(W, Hh, (0.0, 1.0, -3.9), (0.0, 0.7, 0.0), min(40.0 * (W / Hh) ** 0.5, 75.0))
    return scenes.SceneBundle(scene, cam, W, Hh, name="bunny_room"), bunny


def main():
    out = {}
    for name, builder, material in (("binned", H.BVH_BINNED_SAH, L.material_diffuse((0.8, 0.8, 0.8))),
                                    ("sbvh", H.BVH_SPATIAL_SPLIT, L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0))):
        b, bunny = bunny_room(builder, material)
        f = b.flat
        st = bunny.stats()
        assert st["num_input_triangles"] == 69451, st
        p = f"{name}_"
        for k, dt in (("vertices", "VERTEX"), ("triangles", "TRIANGLE"), ("materials", "MATERIAL"), ("sub_nodes", "SUB_BVH_NODE"), ("lights", "EMISSIVE_TRIANGLE"),
                      ("top_nodes", "TOP_BVH_NODE")):
            arr = getattr(f, k)
            assert arr.dtype == getattr(L, dt), (k, arr.dtype)
            out[f"{p}{k}@{dt}"] = np.frombuffer(arr.tobytes(), np.uint8)
        out[p + "top_root"] = np.uint32(f.top_root)
        out[p + "camera@CAMERA"] = np.frombuffer(np.asarray(b.camera).tobytes(), np.uint8)
        out[p + "stats"] = np.array([st["num_input_triangles"], st["num_triangle_refs"], st["num_nodes"], st["max_depth"]], np.uint32)
        sc = O.BoundScene(f)
        o, d = U.random_rays(NRAYS, 31, (-0.9, 0.05, -0.9), (0.9, 1.9, 0.9))
        # the second half aimed at the bunny's bounding box (random directions rarely meet it)
        aim = np.random.default_rng(33).uniform((-0.35, 0.02, -0.25), (0.55, 0.9, 0.45), (NRAYS // 2, 3))
        dd = aim - o[NRAYS // 2:]
        d[NRAYS // 2:] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
        h = O.intersect_batch(sc, o, d, threads=8)
        tmax = np.random.default_rng(32).uniform(0.05, 2.5, NRAYS).astype(np.float32)
        occ = O.intersect_batch(sc, o, d, tmax=tmax, any_hit=True, threads=8)["prim"]
        out[p + "ray_o"], out[p + "ray_d"], out[p + "ray_tmax"] = o, d, tmax
        for k in ("t", "u", "v", "prim", "inst"):
            out[p + "hit_" + k] = h[k]
        out[p + "occluded"] = occ.astype(np.uint8)
        ref, cnt = O.render(sc, b.camera, W, Hh, SPP, seed=6, threads=8)
        out[p + "accum"] = ref[:, :3].astype(np.float32)
        out[p + "counts"] = np.array([cnt["raysGenerated"], cnt["raysExtension"], cnt["raysShadow"], cnt["shadeHits"]], np.uint64)
        print(name, st, "hit fraction", float((h["prim"] >= 0).mean()), "on the bunny", float((h["inst"] == 1).mean()), "occluded", float(occ.mean()),
              "image mean", float(ref[:, :3].mean()) / SPP, cnt)
    path = os.path.join(HERE, "bunny_flat.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
