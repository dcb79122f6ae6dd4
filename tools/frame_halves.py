"""Experiment (GPU box): a 1-spp 1280x720 frame as TWO contexts on one GPU, each owning half of the frame's 32x32 tiles (what two ranks of a multi-GPU job
own), rendering concurrently on their own streams, against one context owning the whole frame.  How much of a frame's latency-bound launch chain does a second,
independent chain hide?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
sys.path.insert(0, ROOT)
from ptamd import device as D, host as H, scenes, layout as L
import bench

W, Hh, frames = 1280, 720, 300
b = scenes.blob_room(W, Hh, material=L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), builder=H.BVH_SPATIAL_SPLIT, level=6)


def make(tiles=None):
    ctx = D.Context(W, Hh, seed=1, samples_in_flight=1)
    ctx.upload_scene(b.flat, sky=b.sky, material_textures=b.material_textures)
    ctx.set_camera(b.camera)
    if tiles is not None:
        ctx.set_tiles(tiles)
    return ctx


def run(ctxs):
    for _ in range(20):
        for c in ctxs:
            c.render(1, sync=False)
        for c in ctxs:
            c.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        for c in ctxs:
            c.render(1, sync=False)
        ctxs[0].resolve_device()
        for c in ctxs:
            c.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


one = make()
print("one context, whole frame: %.4f ms per frame" % run([one]))
one.close()
for n in (2, 3, 4):
    ctxs = [make(bench.tile_rects(W, Hh, r, n)) for r in range(n)]
    print("%d contexts, 1/%d of the tiles each: %.4f ms per frame" % (n, n, run(ctxs)))
    for c in ctxs:
        c.close()
