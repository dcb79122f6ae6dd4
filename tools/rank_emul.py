"""GPU box: an N-GPU job of bench.py emulated on ONE GPU, rank by rank -- every rank's tile share is rendered in turn, exactly as
bench.py would set it up on that rank (same tile lists, same samples in flight, same spp per step), and timed; the job's step time is
the SLOWEST share plus the one reduce of the HDR accumulator.

    python tools/rank_emul.py [--config 4|5] [--scaling weak|strong] [--all-ranks] [--worlds 1 2 4 8] [--steps 2] [--rounds 1]

Prints one line per (N, rank) and a summary per N: min / max share time (the slowest rank named), the reduce at one xGMI link's
153 GB/s (33 MB at 1080p, 133 MB at 4K: the whole message over one link -- a ring moves (N-1)/N of it per link), the predicted
whole-job rate and the predicted speed-up over N = 1 (weak: aggregate rays/s ratio; strong: T1 / TN for the fixed job).  What this
cannot see: xGMI contention, host-side launch skew between ranks, clocks of eight GPUs under load."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (the HIP runtime torch ships is the one libptamd.so binds to)
import bench
from ptamd import scenes, host as H, device as D

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4, choices=[4, 5])
ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
ap.add_argument("--all-ranks", action="store_true", help="every rank's share (default: rank 0's only)")
ap.add_argument("--worlds", type=int, nargs="*", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--rounds", type=int, default=1, help="batches per step (bench.py: 3)")
ap.add_argument("--in-flight", type=int, default=bench.IN_FLIGHT)
ap.add_argument("--json", default=None)
args = ap.parse_args()

W, Hh = (1920, 1080) if args.config == 4 else (3840, 2160)
b = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT, thin_lens=args.config == 5)
LINK_GBS = 153.0
reduce_ms = W * Hh * 16 / (LINK_GBS * 1e9) * 1e3
summary = {"config": args.config, "scaling": args.scaling, "width": W, "height": Hh, "rounds": args.rounds, "reduce_ms_at_153_GBs": round(reduce_ms, 3), "worlds": {}}
t1_ms = rate1 = None
for world in args.worlds:
    ranks = range(world) if args.all_ranks else [0]
    # what bench.py does on every rank: samples in flight from the rank's share, the smallest share decides (all_reduce MIN)
    shares = [bench.tile_rects(W, Hh, r, world) if world > 1 else [] for r in range(world)]
    owned = [sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in s) if s else W * Hh for s in shares]
    in_flight = min(bench.plan_in_flight(args.in_flight, world, o, bench.MAX_ENTRIES) for o in owned)
    if args.scaling == "strong":
        in_flight = min(in_flight, args.in_flight * args.rounds)
    spp_step = in_flight * args.rounds if args.scaling == "weak" else args.in_flight * args.rounds
    ctx = D.Context(W, Hh, seed=1, samples_in_flight=in_flight, ext_queue_fraction=bench.EXT_QUEUE_FRACTION, shadow_queue_fraction=bench.SHADOW_QUEUE_FRACTION)  # as bench.py's ranks
    ctx.upload_scene(b.flat, sky=b.sky)
    ctx.set_camera(b.camera)
    times, rays = {}, {}
    for r in ranks:
        if world > 1:
            ctx.set_tiles(shares[r])
        ctx.render(spp_step)
        ctx.synchronize()
        ctx.clear()
        ctx.reset_stats()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.render(spp_step, sync=False)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        st = ctx.stats()
        times[r], rays[r] = dt * 1e3, (st["rays_extension"] + st["rays_shadow"]) / args.steps
        print(f"config {args.config} {args.scaling:6s} N={world} rank {r}: {owned[r]:8d} pixels, {in_flight:4d} in flight, {spp_step:5d} spp/step, {times[r]:8.2f} ms/step, "
              f"{rays[r] / dt / 1e6:8.1f} Mrays/s", flush=True)
    ctx.close()
    slow = max(times, key=times.get)
    step_ms = times[slow] + (reduce_ms if world > 1 else 0.0)
    job_rays = sum(rays.values()) * (world / len(times))  # (rank 0 only: the other shares assumed alike)
    rate = job_rays / step_ms / 1e3
    if world == 1:
        t1_ms, rate1 = step_ms, rate
    entry = {"samples_in_flight": in_flight, "spp_per_step": spp_step, "share_ms_min": round(min(times.values()), 2), "share_ms_max": round(times[slow], 2), "slowest_rank": slow,
             "share_ms": {str(k): round(v, 2) for k, v in times.items()}, "step_ms_with_reduce": round(step_ms, 2), "predicted_mrays_per_s": round(rate, 1)}
    if rate1:
        entry["predicted_speedup"] = round(rate / rate1 if args.scaling == "weak" else t1_ms / step_ms, 3)
    summary["worlds"][str(world)] = entry
    print(f"== config {args.config} {args.scaling} N={world}: shares {entry['share_ms_min']} .. {entry['share_ms_max']} ms (slowest: rank {slow}), + reduce {reduce_ms if world > 1 else 0:.2f} ms "
          f"-> {entry['step_ms_with_reduce']} ms/step, {entry['predicted_mrays_per_s']} Mrays/s whole job"
          + (f", predicted speed-up {entry['predicted_speedup']}x" if "predicted_speedup" in entry else ""), flush=True)
if args.json:
    with open(args.json, "w") as f:
        json.dump(summary, f, indent=1)
