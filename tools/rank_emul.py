"""GPU box: per-rank throughput of an N-GPU job, emulated on one GPU (rank 0's tile share, 256*N samples in flight)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from ptamd import scenes, host as H, device as D
W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT)
for world in (int(x) for x in (sys.argv[1:] or ["1", "2", "8"])):
    infl = min(256 * world, 4096)
    ctx = D.Context(W, Hh, seed=1, samples_in_flight=infl)
    ctx.upload_scene(b.flat, sky=b.sky)
    ctx.set_camera(b.camera)
    if world > 1:
        ctx.set_tiles(bench.tile_rects(W, Hh, 0, world))
    ctx.render(infl)
    ctx.synchronize()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.render(infl, sync=False)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(f"world {world}: rank-0 share {infl} spp/step, {(st['rays_extension'] + st['rays_shadow']) / dt / 1e6:8.1f} Mrays/s per rank, {dt / 3 * 1e3:6.1f} ms/step", flush=True)
    ctx.close()
