"""Diagnostic: where do beam packets (k_trace_packet) and the per-ray kernel disagree?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from ptamd import scenes, device as D
import gpu_util as U
W, Hh = 256, 144
for thin in (False, True):
    b = scenes.instanced_grid(W, Hh, level=4, thin_lens=thin, sky_size=(16, 8))
    packet = U.make_ctx(D, b, W, Hh, flags=D.FLAG_PACKET_INTERSECT)
    per_ray = U.make_ctx(D, b, W, Hh)
    for sample in range(3):
        o, d, _ = per_ray.gen_rays(sample, W * Hh)
        got, want = packet.intersect(o, d), per_ray.intersect(o, d)
        diff = np.flatnonzero((got["prim"] != want["prim"]) | (got["inst"] != want["inst"]))
        print(f"thin={thin} sample={sample}: {len(diff)} differ of {len(o)}")
        for k in diff[:12]:
            print(f"   ray {k} (packet {k // 64} lane {k % 64}): packet prim {got['prim'][k]} inst {got['inst'][k]} t {got['t'][k]:.7g} u {got['u'][k]:.5f} v {got['v'][k]:.5f} | per-ray prim {want['prim'][k]} inst {want['inst'][k]} t {want['t'][k]:.7g} u {want['u'][k]:.5f} v {want['v'][k]:.5f}")
