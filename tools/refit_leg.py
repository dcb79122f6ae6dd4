"""GPU box: bench.py's `dynamic.refit` object alone (deform ticks of two meshes + the rebuilt tree per frame), without the rest of the bench."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bench  # noqa: E402
from ptamd import device as D, host as H, layout as L, scenes  # noqa: E402

W, Hh = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)  # (bench.py's default frame)
out = bench.dynamic_refit_times(D, H, L, scenes, W, Hh, 0)
print(f"{W}x{Hh}")
for name, o in out.items():
    if not isinstance(o, dict):
        continue
    print(name, json.dumps({k: v for k, v in o.items() if k not in ("what",)})[:900])
