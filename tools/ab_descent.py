"""A/B on one box: the shared descent (pt_descend.h) off / shadow rays only / extension rays only / both, on the benchmark scene
(and with every instance entered), measured like the headline (bench.measure_scene).  Usage: tools/ab_descent.py [in_flight] [flags...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bench
from ptamd import scenes, device as D, host as H

in_flight = int(sys.argv[1]) if len(sys.argv) > 1 else 256
flag_sets = [int(x) for x in sys.argv[2:]] or [0]
uses = [int(x) for x in os.environ.get("AB_USES", "0,3,1,2").split(",")]
W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT)
for flags in flag_sets:
    for use in uses:
        os.environ["PTAMD_DESCENT"] = str(use)
        r = bench.measure_scene(D, b, W, Hh, 0, in_flight, flags=flags, steps=3, warmup=1, rounds=1)
        print(json.dumps({"flags": flags, "descent": use, "mrays_per_s": r["mrays_per_s"], "ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms_per_step"]}), flush=True)
