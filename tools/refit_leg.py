"""GPU box: bench.py's `dynamic.refit` object alone (deform ticks of two meshes + the rebuilt tree per frame), without the rest of the bench;
--update: the rigid-motion ticks instead (`dynamic.benchmark_scene_*`, `instances_1000`, `instances_10000`).  The process is pinned like bench.py's."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bench  # noqa: E402
from ptamd import device as D, host as H, layout as L, scenes  # noqa: E402

print(bench.pin_to_one_l3_domain(0))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
W, Hh = (int(args[0]), int(args[1])) if len(args) > 1 else (1920, 1080)  # (bench.py's default frame)
if "--update" in sys.argv:
    out = bench.dynamic_update_times(D, H, L, scenes, scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6), W, Hh, 0)
else:
    out = bench.dynamic_refit_times(D, H, L, scenes, W, Hh, 0)
print(f"{W}x{Hh}")
for name, o in out.items():
    if not isinstance(o, dict):
        continue
    print(name, json.dumps({k: v for k, v in o.items() if k not in ("what",)})[:900])
