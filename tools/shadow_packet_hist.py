"""Diagnostic (needs the -DPT_TRACE_STATS build): how many leaves / nodes does a packet of 64 consecutive first-pass shadow rays touch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTAMD_LIB", os.path.join(ROOT, "opencl-path-tracer_amd", "csrc", "variants", "libptamd_stats.so"))
os.environ["PTAMD_PACKET"] = "3"  # primary AND shadow rays of the first pass through the packet kernel
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import ctypes as C
import numpy as np
from ptamd import scenes, device as D
W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, level=6)
ctx = D.Context(W, Hh, seed=1, samples_in_flight=64, max_bounces=1)
ctx.upload_scene(b.flat, sky=b.sky); ctx.set_camera(b.camera)
out = (C.c_ulonglong * 64)()
D.lib().pt_debug_trace_stats(out, 64)
ctx.render(64)
D.lib().pt_debug_trace_stats(out, 64)
h = np.array(list(out), np.float64)
leaves, nodes = h[:24], h[24:]
print("packets", int(leaves.sum()), "stats", ctx.stats()["rays_shadow"], "shadow rays")
print("leaves per packet: cumulative share", np.round(np.cumsum(leaves) / leaves.sum(), 3).tolist())
print("nodes/4 per packet: cumulative share", np.round(np.cumsum(nodes) / nodes.sum(), 3).tolist())
