"""sha256 (first 16 hex digits) of (nodes, triangles, original triangle) of the host library's three builders on a handful of meshes -> builder_trees.json.
The committed file was written by the list-based builders of commit b2849b6 (one std::vector of references per node, 31 planes swept per node); the builders
that work in place on one array and sweep the occupied bins only must reproduce it byte for byte (tests/test_host_scene.py).
    python tests/golden/make_builder_trees.py            # rewrites tests/golden/builder_trees.json with what the CURRENT library builds"""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import host as H, layout as L, scenes  # noqa: E402


def digest(arrays):
    m = hashlib.sha256()
    for x in arrays:
        m.update(np.ascontiguousarray(x).tobytes())
    return m.hexdigest()[:16]


def trees(levels=(2, 4, 5, 6)):
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    out = {}
    for lvl in levels:
        v, f = scenes.icosphere(lvl)
        for k in (0, 3):
            p = (v * 0.5 * (1.0 + 0.1 * np.sin(k + 1.0 + 5.0 * v[:, :1]))).astype(np.float32)
            for b in (H.BVH_BINNED_SAH, H.BVH_BINNED_FAST, H.BVH_SPATIAL_SPLIT):
                if b != H.BVH_BINNED_FAST and lvl == 6:
                    continue
                out[f"ico{lvl}_{k}_{b}"] = digest(H.Mesh(p, f.astype(np.uint32), [mat], builder=b).bvh())
    rng = np.random.default_rng(5)  # a soup: coincident vertices (centroids in one bin: no split), long thin triangles
    p = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    p[:600] = p[0]
    f = rng.integers(0, 3000, (5000, 3)).astype(np.uint32)
    for b in (0, 1, 2):
        out[f"soup_{b}"] = digest(H.Mesh(p, f, [mat], builder=b).bvh())
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "builder_trees.json")
    with open(path, "w") as fh:
        json.dump(trees(), fh, indent=1, sort_keys=True)
    print("wrote", path)
