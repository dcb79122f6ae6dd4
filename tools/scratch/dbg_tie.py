"""Diagnostic: is the one pixel that differs between 4 + 4 and 8 samples (packets of 16 pixels x 4 samples vs 8 x 8) an exact-t tie?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("opencl-path-tracer_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from ptamd import scenes, device as D
import gpu_util as U
W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, level=6)
pk = U.make_ctx(D, b, W, Hh, seed=1, flags=D.FLAG_PACKET_INTERSECT)
pr = U.make_ctx(D, b, W, Hh, seed=1)
rays = [pr.gen_rays(s, W * Hh) for s in range(8)]
pix = rays[0][2]
target = 685 * W + 1801
pos = int(np.flatnonzero(pix == target)[0])
for group, per in ((range(0, 4), 16), (range(4, 8), 16), (range(0, 8), 8)):
    g0 = pos // per * per
    sel = np.arange(g0, g0 + per)
    o = np.stack([rays[s][0][sel] for s in group], 1).reshape(-1, 3)
    d = np.stack([rays[s][1][sel] for s in group], 1).reshape(-1, 3)
    got, want = pk.intersect(o, d), pr.intersect(o, d)
    diff = np.flatnonzero((got["prim"] != want["prim"]) | (got["inst"] != want["inst"]))
    print(f"samples {list(group)}: packet of {len(o)} rays, {len(diff)} records differ, packet launches {pk.stats()['packet_launches']}")
    for k in diff:
        print(f"   lane {k} pixel {pix[sel[k // len(group)]]} sample {list(group)[k % len(group)]}: packet prim {got['prim'][k]} t {got['t'][k]:.8g} u {got['u'][k]:.6f} v {got['v'][k]:.6f} | per-ray prim {want['prim'][k]} t {want['t'][k]:.8g} u {want['u'][k]:.6f} v {want['v'][k]:.6f}")
tri = b.flat.triangles
