"""Micro-benchmark of k_trace through pt_intersect: primary rays (coherent) and shuffled rays (incoherent)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import numpy as np
from ptamd import scenes, device as D, host as H

def run(name, b, W, Hh):
    ctx = D.Context(W, Hh, seed=1)
    ctx.upload_scene(b.flat, sky=b.sky)
    ctx.set_camera(b.camera)
    o, d, _ = ctx.gen_rays(0, W * Hh)
    r = ctx.intersect(o, d, repeat=5)
    hit = r["prim"] >= 0
    print(f"{name:28s} primary   : {len(o)/r['ms']/1e3:8.1f} Mrays/s  ({r['ms']:.3f} ms, hit {hit.mean():.2f})")
    # secondary-like rays: from hit points, random hemisphere-ish directions
    rng = np.random.default_rng(0)
    p = o[hit] + d[hit] * r["t"][hit][:, None] * 0.999
    nd = rng.normal(size=p.shape).astype(np.float32); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    nd[:, 1] = np.abs(nd[:, 1])
    r2 = ctx.intersect(p, nd, repeat=5)
    print(f"{name:28s} secondary : {len(p)/r2['ms']/1e3:8.1f} Mrays/s  ({r2['ms']:.3f} ms, hit {(r2['prim']>=0).mean():.2f})")
    perm = rng.permutation(len(p))
    r3 = ctx.intersect(p[perm], nd[perm], repeat=5)
    print(f"{name:28s} shuffled  : {len(p)/r3['ms']/1e3:8.1f} Mrays/s  ({r3['ms']:.3f} ms)")
    ctx.close()

W, Hh = 1920, 1080
run("instanced grid 4x3 (12 inst)", scenes.instanced_grid(W, Hh, level=6), W, Hh)
run("blob room (1 big inst)", scenes.blob_room(W, Hh, level=6, builder=H.BVH_SPATIAL_SPLIT), W, Hh)
