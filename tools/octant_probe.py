"""GPU box: is k_trace faster on secondary rays grouped by direction octant (ray reordering feasibility)?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import numpy as np
from ptamd import scenes, device as D
W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, level=6)
ctx = D.Context(W, Hh, seed=1)
ctx.upload_scene(b.flat, sky=b.sky); ctx.set_camera(b.camera)
o, d, _ = ctx.gen_rays(0, W * Hh)
o, d = np.tile(o, (4, 1)), np.tile(d, (4, 1))
r = ctx.intersect(o, d)
hit = r["prim"] >= 0
rng = np.random.default_rng(0)
p = (o[hit] + d[hit] * r["t"][hit][:, None] * 0.999).astype(np.float32)
nd = rng.normal(size=p.shape).astype(np.float32); nd /= np.linalg.norm(nd, axis=1, keepdims=True); nd[:, 1] = np.abs(nd[:, 1])
def run(tag, po, pd):
    ms = min(ctx.intersect(po, pd, repeat=3)["ms"] for _ in range(2))
    print(f"{tag:34s} {len(po) / ms / 1e3:8.1f} Mrays/s", flush=True)
run("secondary, queue order", p, nd)
octant = (nd[:, 0] < 0).astype(np.int32) | ((nd[:, 1] < 0) << 1) | ((nd[:, 2] < 0) << 2)
order = np.argsort(octant, kind="stable")
run("secondary, grouped by octant", p[order], nd[order])
# finer: octant + dominant axis (24 classes)
dom = np.abs(nd).argmax(1)
order = np.argsort(octant * 3 + dom, kind="stable")
run("secondary, octant + major axis", p[order], nd[order])
perm = rng.permutation(len(p))
run("secondary, random order", p[perm], nd[perm])
