"""The host library's Mesh() for the rebuilt-tree-per-frame case of bench.py (20 480 triangles, fast binned builder), stage by stage: no GPU needed.
PTAMD_BUILD_TIMING=1 makes the library print its own stages (arrays, smooth normals, tree: top phase / subtrees + assembly, emissive list)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import host as H, layout as L, scenes  # noqa: E402


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    v, f = scenes.icosphere(level)
    p0 = (v * 0.5).astype(np.float32)
    f = f.astype(np.uint32)
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    mb = scenes._MeshBuilder()
    mats = scenes._room_materials()
    scenes._room(mb, mats)
    room = mb.build(mats, H.BVH_BINNED_SAH)
    build, flatten = [], []
    for k in range(40):
        p = (p0 * (1.0 + 0.1 * np.sin(k + 1.0 + 5.0 * p0[:, :1]))).astype(np.float32)
        t0 = time.perf_counter()
        mesh = H.Mesh(p, f, [mat], builder=H.BVH_BINNED_FAST)
        t1 = time.perf_counter()
        scene = H.Scene()
        scene.add_node(room)
        scene.add_node(mesh, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
        scene.flatten()
        t2 = time.perf_counter()
        build.append((t1 - t0) * 1e3)
        flatten.append((t2 - t1) * 1e3)
    b, fl = np.sort(build), np.sort(flatten)
    print(f"{len(f)} triangles, threads {os.environ.get('PTAMD_BUILD_THREADS', 'all')}: Mesh() ms min {b[0]:.3f} median {b[20]:.3f}; flatten ms min {fl[0]:.3f} median {fl[20]:.3f}")


main()
