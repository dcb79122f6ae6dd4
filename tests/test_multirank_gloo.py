"""N > 1 path on CPU (gloo, world_size 2): the tile partition bench.py uses, per-rank rendering of
disjoint tiles and ONE reduce of the HDR accumulator onto rank 0, through bench.py's own functions (tile_rects,
plan_in_flight, reduce_accumulator, aggregate).  The per-rank 'renderer' here is the oracle (CPU) restricted to the
rank's pixels -- there is no GPU in this container; the same job with the HIP contexts as renderers, two ranks on
one GPU, is tests/test_gpu_multirank.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
W, H, SPP = 64, 40, 2


def _pixels_of(rects, width):
    px = []
    for x0, y0, x1, y1 in rects:
        ys, xs = np.mgrid[y0:y1, x0:x1]
        px.append((ys * width + xs).reshape(-1))
    return np.concatenate(px).astype(np.uint32)


def _worker(rank, world, port, out_path):
    for p in (os.path.join(ROOT, "opencl-path-tracer_amd"), os.path.join(ROOT, "oracle"), ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import orclib as O
    from ptamd import scenes
    b = scenes.instanced_grid(W, H, level=2, sky_size=(16, 8))
    sc = O.BoundScene(b.flat, sky=b.sky)
    rects = bench.tile_rects(W, H, rank, world, tile=16)
    px = _pixels_of(rects, W)
    assert bench.plan_in_flight(256, world, len(px)) == 256 * world  # weak scaling: N x the samples in flight on 1/N of the pixels
    assert bench.plan_in_flight(256, 8, 3840 * 2160 // 8) * (3840 * 2160 // 8) <= bench.MAX_ENTRIES  # 4K: capped by memory
    acc, cnt = O.render(sc, b.camera, W, H, SPP, seed=1, pixels=px, threads=1)
    accum = torch.from_numpy(acc)
    # the exchange and the whole-job aggregation are bench.py's own functions (host tensors: the gloo rehearsal path)
    bench.reduce_accumulator(dist, accum, "gloo", world)
    (rays,), tmax = bench.aggregate(dist, torch, [cnt["raysExtension"] + cnt["raysShadow"]], float(rank + 1), "gloo", world)
    if rank == 0:
        np.savez(out_path, accum=accum.numpy(), rays=np.array([rays]), tmax=np.array([tmax]), owned=len(px))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_partition_is_exact():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 4, 8):
        cover = np.zeros(1080 * 1920, np.int32)
        counts = []
        for r in range(world):
            px = _pixels_of(bench.tile_rects(1920, 1080, r, world), 1920)
            np.add.at(cover, px, 1)
            counts.append(len(px))
        assert cover.min() == 1 and cover.max() == 1, "every pixel belongs to exactly one rank"
        assert max(counts) - min(counts) <= 0.02 * 1920 * 1080 / world, "interleaved tiles balance the ranks"


@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_render_and_reduce(tmp_path, world):
    """world = 8: the rank count of the driver's scaling run (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`) -- eight
    processes through bench.py's planning (tile lists, samples in flight from the rank's share), the one reduce and the aggregation."""
    out = str(tmp_path / "rank0.npz")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    z = np.load(out)
    for p in (os.path.join(ROOT, "opencl-path-tracer_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import orclib as O
    from ptamd import scenes
    b = scenes.instanced_grid(W, H, level=2, sky_size=(16, 8))
    full, cnt = O.render(O.BoundScene(b.flat, sky=b.sky), b.camera, W, H, SPP, seed=1, threads=2)
    assert np.array_equal(z["accum"], full), "reduced tile renders != single-rank render"
    assert z["rays"][0] == cnt["raysExtension"] + cnt["raysShadow"]
    assert z["tmax"][0] == float(world) and 0 < z["owned"] < W * H  # elapsed = the slowest rank's


def test_planning_of_the_eight_rank_jobs():
    """What every rank of the driver's N = 8 runs computes before it touches its GPU (bench.main): config 4 and config 5, weak and strong."""
    sys.path.insert(0, ROOT)
    import bench
    for (w, h), want_weak in (((1920, 1080), 4096), ((3840, 2160), 1024)):
        owned = []
        for r in range(8):
            rects = bench.tile_rects(w, h, r, 8)
            owned.append(sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in rects))
        assert sum(owned) == w * h
        plans = [bench.plan_in_flight(bench.IN_FLIGHT, 8, o, bench.MAX_ENTRIES) for o in owned]  # the driver's default request
        batch = min(plans)  # what the MIN all-reduce of bench.main agrees on
        assert batch == want_weak, (plans, want_weak)  # 1080p: the library's 4 096 planes; 4K: capped by the entries that fit in HBM
        assert batch % 256 == 0, "only multiples of 256 keep the samples of a pixel together in the queues (bundles, coherent first bounce)"
        assert batch * max(owned) <= bench.MAX_ENTRIES * 1.016  # the budget's 1/64 of slack for the uneven shares
        assert bench.resident_bytes(batch, max(owned)) < 230e9, "queues and planes of the largest share fit a 288 GB part beside the scene"
        # strong scaling: the job is IN_FLIGHT x rounds samples per pixel whatever N is; a rank never keeps more of them in flight than the job holds
        job = bench.IN_FLIGHT * 3
        strong = min(batch, job)
        assert strong == min(job, want_weak) and strong % 256 == 0  # 1080p: the whole 1 536-sample job in flight at once; 4K: 1 024 of them


def test_in_flight_falls_back_to_what_the_device_has_free():
    """bench.fit_in_flight: a card with another tenant / less HBM costs the line a percent or two, not the run."""
    sys.path.insert(0, ROOT)
    import bench
    px = 1920 * 1080
    assert bench.fit_in_flight(512, px, 280 << 30) == 512
    n = bench.fit_in_flight(512, px, 150 << 30)
    assert n == 256 and bench.resident_bytes(n, px) + (6 << 30) <= 150 << 30  # multiples of 256: 512 -> 256
    assert bench.fit_in_flight(512, px, 60 << 30) == 128 and bench.fit_in_flight(512, px, 1 << 30) == 1
    assert [bench.shrink_in_flight(x) for x in (4096, 768, 600, 256, 3, 1)] == [3840, 512, 512, 128, 1, 1]


def test_bench_without_a_launcher_starts_its_ranks_as_child_processes():
    """`python bench.py --gpus 2` outside torchrun must not bail out on WORLD_SIZE: it starts the ranks itself (torch.distributed.run
    as a CHILD, before the parent has touched a GPU) and exits with the children's status.  Here (no GPU) the children get as far as
    bench.py's own 'needs a GPU' message -- which proves they were started as ranks of a 2-rank job -- and the parent reports failure."""
    import subprocess
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: tests/test_gpu_multirank.py runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "launch with torch.distributed.run" not in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_bench_line_reads_the_counters_of_the_newest_round():
    """roofline.traffic and the per-kernel issue figures of the bench line come from the committed PMC summaries (profiles/roundN/traffic*.json) of the
    timed configuration: the NEWEST round's, and the file written by tools/traffic_json.py for that build -- not an archived copy that happens to sort
    behind it (round 5: `traffic_r5p_before_plain_fma.json` did, and two bench lines quoted the instructions per ray of the build before)."""
    import re
    import bench
    rounds = sorted(int(m.group(1)) for m in (re.match(r"round(\d+)$", d) for d in os.listdir(os.path.join(ROOT, "profiles"))) if m)
    for flags, name in ((0, "traffic.json"), (2, "traffic_two_level.json")):
        tj = bench.newest_traffic_json(1920, 1080, 6, bench.IN_FLIGHT, 1, flags)
        assert tj is not None and tj["_path"] == os.path.join("profiles", f"round{rounds[-1]}", name), tj and tj["_path"]


def test_resident_bytes_follow_the_queue_fractions():
    """Round 6: queues smaller than the batch.  bench.bytes_per_entry is ensureQueues' accounting (csrc/ptamd.hip): 164 B per entry with queues
    as large as the batch; with the headline's fractions the 512-sample batch of the 1080p frame stays under 130 GB, and the fallback planner follows."""
    sys.path.insert(0, ROOT)
    import bench
    px = 1920 * 1080
    assert bench.bytes_per_entry() == bench.bytes_per_entry(0.0, 0.0) == bench.bytes_per_entry(1.0, 1.0) == bench.BYTES_PER_QUEUE_ENTRY == 164
    assert abs(bench.bytes_per_entry(0.30, 0.55) - (36 + 80 * 0.30 + 48 * 0.55)) < 1e-9 and 85 < bench.bytes_per_entry(bench.EXT_QUEUE_FRACTION, bench.SHADOW_QUEUE_FRACTION) < 95  # a pinhole's bundles: the first queue's origin / throughput planes shrink too
    assert abs(bench.bytes_per_entry(0.30, 0.55, thin_lens=True) - (68 + 48 * 0.30 + 48 * 0.55)) < 1e-9
    assert bench.bytes_per_entry(0.5, 1.0) == 36 + 80 * 0.5 + 48
    per = bench.bytes_per_entry(bench.EXT_QUEUE_FRACTION, bench.SHADOW_QUEUE_FRACTION)
    assert bench.resident_bytes(512, px, per) < 130e9 < bench.resident_bytes(512, px)  # 113 GB against 191 GB
    assert bench.resident_bytes(1, px, per) == int(px * per)  # one sample in flight: no extra planes
    # the planner: what fits 150 GB with the fractions is the whole 512, without them 256
    assert bench.fit_in_flight(512, px, 150 << 30, per_entry=per) == 512 and bench.fit_in_flight(512, px, 150 << 30) == 256
    assert bench.fit_in_flight(512, px, 80 << 30, per_entry=per) == 256
