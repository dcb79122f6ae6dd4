"""GPU box: bench.py's `dynamic` object alone (rigid instance motion per tick, 14 / 1 000 / 10 000 instances; deforming meshes through refit)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bench
from ptamd import device as D, host as H, layout as L, scenes
W, Hh = 1920, 1080
which = sys.argv[1:] or ["rigid", "refit"]
out = {}
if "rigid" in which:
    b = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT)
    out.update(bench.dynamic_update_times(D, H, L, scenes, b, W, Hh, 0))
if "refit" in which:
    out["refit"] = bench.dynamic_refit_times(D, H, L, scenes, W, Hh, 0)
print(json.dumps(out, indent=1))
