"""Diagnostic: 4 + 4 samples vs 8 samples in one call, with the packet kernel generating the primary rays / reading k_gen's / unused."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("opencl-path-tracer_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from ptamd import scenes, device as D
import gpu_util as U
W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, level=6)
imgs = {}
for name, flags in (("fused packets", 0), ("k_gen + packets", D.FLAG_QUEUE_PRIMARY_RAYS), ("no packets", D.FLAG_NO_PACKETS)):
    c1 = U.make_ctx(D, b, W, Hh, seed=1, flags=flags)
    c1.render(4); c1.render(4)
    a = c1.read_accum()[:, :3]; s1 = c1.stats(); c1.close()
    c2 = U.make_ctx(D, b, W, Hh, seed=1, flags=flags)
    c2.render(8)
    bb = c2.read_accum()[:, :3]; s2 = c2.stats(); c2.close()
    bad = ~np.isclose(a, bb, rtol=1e-5, atol=1e-5 * bb.max()).all(axis=1)
    print(f"{name:16s}: 4+4 vs 8: {bad.sum()} pixels differ; packet launches {s1['packet_launches']} / {s2['packet_launches']}; ext rays {s1['rays_extension']} / {s2['rays_extension']}", flush=True)
    if bad.any():
        k = np.flatnonzero(bad)[:8]
        print("   pixels", [(int(i % W), int(i // W)) for i in k], a[k].round(4).tolist()[:3], bb[k].round(4).tolist()[:3])
    imgs[name] = bb
for n1 in imgs:
    for n2 in imgs:
        if n1 < n2:
            bad = ~np.isclose(imgs[n1], imgs[n2], rtol=1e-5, atol=1e-5 * imgs[n1].max()).all(axis=1)
            print(f"8 samples, {n1} vs {n2}: {bad.sum()} pixels differ")
