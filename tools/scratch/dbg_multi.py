"""Diagnostic: first pass of a batch (pt_primary_pass: bundles where pt_render uses them) against the per-ray kernel, field by field."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("opencl-path-tracer_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
from ptamd import device as D, scenes
import gpu_util as U

W, Hh, spp = 96, 54, 64
b = scenes.instanced_grid(W, Hh, level=4, sky_size=(16, 8))
first = U.make_ctx(D, b, W, Hh, samples_in_flight=spp)
queued = U.make_ctx(D, b, W, Hh, flags=D.FLAG_QUEUE_PRIMARY_RAYS | D.FLAG_NO_PACKETS, samples_in_flight=spp)
n = W * Hh * spp
o, d, pixel, got = first.primary_pass(0, spp, n)
_, _, _, want = queued.primary_pass(0, spp, n)
same = (got["prim"] == want["prim"])
print("entries", n, "different prim", (~same).sum())
for k in ("t", "u", "v"):
    a, w = got[k][same].view(np.int32).astype(np.int64), want[k][same].view(np.int32).astype(np.int64)
    dd = np.abs(a - w)
    print(k, "differ", (dd > 0).sum(), "max ulps", dd.max(), "hist", np.bincount(np.minimum(dd, 8))[:9])
idx = np.nonzero(same & (got["t"] != want["t"]))[0][:8]
for i in idx:
    print(i, i // 256, i % 64, got["prim"][i], [float(got[k][i]) for k in "tuv"], [float(want[k][i]) for k in "tuv"])
