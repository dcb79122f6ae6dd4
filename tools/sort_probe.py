"""GPU box: what would regrouping the incoherent rays buy (VERDICT r1 task 3)?  k_trace on synthetic bounce rays -- hit points of
the primary rays (one sample of every pixel, 8x8-block pixel order like the queue) + random upper-hemisphere directions, and
shadow-like segments of length 3 -- in (a) queue order, (b) sorted by Morton code of the origin cell (32^3 grid over the scene
box) x direction octant, (c) octant only, (d) shuffled.  The sort itself is done on the host and NOT timed: (b) - (a) is the upper
bound of what a device counting sort could win before paying for itself (a pass over 48-B records: ~0.2 ms per 10 M rays)."""
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


def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


lo, hi = p.min(0), p.max(0)
cell = np.clip(((p - lo) / (hi - lo) * 32).astype(np.int64), 0, 31)
morton = part1by2(cell[:, 0]) | (part1by2(cell[:, 1]) << 1) | (part1by2(cell[:, 2]) << 2)
octant = (nd[:, 0] < 0).astype(np.uint64) | ((nd[:, 1] < 0).astype(np.uint64) << 1) | ((nd[:, 2] < 0).astype(np.uint64) << 2)
orders = {"queue order": np.arange(len(p)), "Morton(origin cell 32^3) x octant": np.argsort(morton * 8 + octant, kind="stable"),
          "octant x Morton": np.argsort(octant * (1 << 15) + morton, kind="stable"), "octant only": np.argsort(octant, kind="stable"),
          "shuffled": rng.permutation(len(p))}
for any_hit in (False, True):
    tm = np.full(len(p), 3.0, np.float32) if any_hit else None
    base = None
    for tag, order in orders.items():
        ms = min(ctx.intersect(p[order], nd[order], tmax=None if tm is None else tm[order], any_hit=any_hit, repeat=3)["ms"] for _ in range(2))
        base = base or ms
        print(f"{'any-hit' if any_hit else 'closest'} {tag:36s} {len(p) / ms / 1e3:8.1f} Mrays/s  {ms:7.3f} ms  ({base / ms:5.3f} x queue order)", flush=True)
