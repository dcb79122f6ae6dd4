import sys, os
sys.path[:0] = ["opencl-path-tracer_amd", "oracle", "tests"]
import numpy as np
import golden_io, gpu_util as U
from ptamd import device as D
g = golden_io.load()
for name in ("mixed", "inst"):
    flat, cam, sky, tex = golden_io.scene_inputs(g, name)
    ctx = U.make_ctx(D, flat, 64, 36, camera=cam, sky=sky, tex=tex)
    o, d = g[f"isect_{name}_o"], g[f"isect_{name}_d"]
    got = ctx.intersect(o, d)
    wp = g[f"isect_{name}_prim"]
    gh, wh = got["prim"] >= 0, wp >= 0
    bad = np.flatnonzero(gh != wh)
    print(name, "flips", len(bad), bad[:10])
    for i in bad[:6]:
        print("  ray", i, "o", o[i], "d", d[i], "gpu", got["prim"][i], got["t"][i], "ref", wp[i], g[f"isect_{name}_t"][i])
    both = gh & wh
    print("  max rel dt", np.max(np.abs(got["t"][both] - g[f"isect_{name}_t"][both]) / g[f"isect_{name}_t"][both]))
    occ = ctx.intersect(o, d, tmax=g[f"shadow_{name}_len"], any_hit=True)["prim"]
    print("  occlusion diffs", (occ != g[f"shadow_{name}_occluded"]).sum())
    ctx.close()
