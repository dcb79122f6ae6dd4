"""GPU box: what would ordering k_shade's outputs by direction INSIDE a 512-entry block buy?  The real queue after the first shade pass:
runs of rays that leave from one pixel's footprint (here: 64 copies of the primary hit point, jittered by 1e-3) -- bounce-like
(random upper-hemisphere directions) and shadow-like (towards random points of the light quad, length = distance).  Orders: as
queued; every block of 512 entries stably sorted by direction octant; by a 6-bit direction code (octant x 8 cells of the dominant
components).  The sort is done on the host and not timed (inside k_shade it would only change the slot an entry is written to)."""
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
r = ctx.intersect(o, d)
hit = np.flatnonzero(r["prim"] >= 0)
rng = np.random.default_rng(0)
hit = hit[:: max(1, len(hit) // 500000)][:500000]  # every k-th hit pixel, in queue order
p1 = (o[hit] + d[hit] * r["t"][hit][:, None] * 0.999).astype(np.float32)
SPP = 64
p = np.repeat(p1, SPP, axis=0) + rng.normal(0, 1e-3, (len(p1) * SPP, 3)).astype(np.float32)
n = len(p)
nd = rng.normal(size=p.shape).astype(np.float32); nd /= np.linalg.norm(nd, axis=1, keepdims=True); nd[:, 1] = np.abs(nd[:, 1])
# light quad of scenes.instanced_grid: read it from the flattened scene's emissive triangles
lv = np.asarray(b.flat.lights["vertices"], np.float32).reshape(-1, b.flat.lights["vertices"].shape[-1])[:, :3]
lmin, lmax = lv.min(0), lv.max(0)
print("light box", lmin, lmax, "rays", n, flush=True)
lp = (lmin + rng.random((n, 3)).astype(np.float32) * (lmax - lmin)).astype(np.float32)
sd = lp - p
sl = np.linalg.norm(sd, axis=1).astype(np.float32)
sd = (sd / sl[:, None]).astype(np.float32)


def codes(v):
    octant = (v[:, 0] < 0).astype(np.int64) | ((v[:, 1] < 0).astype(np.int64) << 1) | ((v[:, 2] < 0).astype(np.int64) << 2)
    a = np.abs(v)
    fine = (np.minimum((a[:, 0] * 2).astype(np.int64), 1) | (np.minimum((a[:, 1] * 2).astype(np.int64), 1) << 1) | (np.minimum((a[:, 2] * 2).astype(np.int64), 1) << 2))
    return octant, octant * 8 + fine


def block_sorted(key, block=512):
    blk = np.arange(n, dtype=np.int64) // block
    return np.argsort(blk * 64 + key, kind="stable")


for name, dirs, tmax, any_hit in (("bounce-like closest", nd, None, False), ("shadow-like any-hit", sd, sl * 0.999, True)):
    oc, fine = codes(dirs)
    orders = {"as queued": np.arange(n), "octant inside 512-blocks": block_sorted(oc), "6-bit direction code inside 512-blocks": block_sorted(fine),
              "octant inside 2048-blocks": block_sorted(oc, 2048)}
    base = None
    for tag, order in orders.items():
        ms = min(ctx.intersect(p[order], dirs[order], tmax=None if tmax is None else tmax[order], any_hit=any_hit, repeat=2)["ms"] for _ in range(1))
        base = base or ms
        print(f"{name:22s} {tag:42s} {n / ms / 1e3:8.1f} Mrays/s  {ms:7.3f} ms  ({base / ms:5.3f} x as queued)", flush=True)
