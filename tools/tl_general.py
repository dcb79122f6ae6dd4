"""GPU box: bench.py's `two_level_general` object alone (432 turned + non-uniformly scaled / 208 translated + uniformly scaled instances of the 82 k-triangle
meshes: entered by the general route, by the parked route, the default, all copied), one compact line per way.
    python tools/tl_general.py [--only general_432] [--ways entered,copied] [--in-flight 256] [--json out.json]      (PTAMD_LIB=... for a variant library)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bench  # noqa: E402
from ptamd import device as D, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--ways", default="entered,entered_parked,default_flags,copied")
ap.add_argument("--in-flight", type=int, default=256)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--json", default="")
a = ap.parse_args()
res = bench.two_level_general_times(D, scenes, 1920, 1080, a.level, 0, a.in_flight, only=a.only.split(",") if a.only else None, ways=tuple(a.ways.split(",")))
for name, r in res.items():
    print(name, "--", r["what"][:90])
    for way in a.ways.split(","):
        if way in r:
            m = r[way]
            print(f"  {way:15s} {m['mrays_per_s']:9.1f} Mrays/s  per-ray kernels {m['per_ray_kernels_mrays_per_s']:8.1f}  by class "
                  + " / ".join(f"{k} {v['in_kernel_mrays_per_s']}" for k, v in m["rays_by_class"].items())
                  + f"  kernel ms {m['kernel_ms_per_step']}  instances {m['instances']}  upload {m['scene_upload_s']} s", flush=True)
    print("  ratios", {k: v for k, v in r.items() if k.endswith("_over_copied") or k.endswith("_per_ray_kernels")}, flush=True)
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
