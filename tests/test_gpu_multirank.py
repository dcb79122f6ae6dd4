"""The N > 1 code path of bench.py ON THE DEVICE: two ranks (one process each, torch.distributed launcher) sharing the one
GPU of the test box, gloo for the host-side exchange -- tile lists, per-rank accumulator planes indexed by owned-pixel
ordinal, the reduce ordered after the render on the ranks' streams, the whole-job aggregation -- against a single-rank run
of the same job.  (RCCL itself needs two GPUs; what it replaces here is only the transport of the one reduce.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
COMMON = ["--steps", "1", "--warmup", "1", "--rounds", "2", "--width", "192", "--height", "128", "--level", "3", "--no-cpu-baseline", "--no-roofline",
          "--no-frame"]


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    return json.loads(lines[0])


def test_two_rank_job_on_one_gpu_equals_the_single_rank_job(gpu, tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    port = 29600 + os.getpid() % 300
    j2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(port), "bench.py", "--gpus", "2", "--share-gpu", "--backend", "gloo", "--in-flight", "8", "--dump-accum", two] + COMMON, env)
    j1 = _run([sys.executable, "bench.py", "--gpus", "1", "--in-flight", "16", "--dump-accum", one] + COMMON, env)
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1 and j2["scaling"] == "weak"
    assert j2["config"]["samples_in_flight"] == 16 and j2["config"]["pixels_per_rank"] == 192 * 128 // 2
    # the same paths were traced: 2 ranks x half the pixels x 16 samples x 2 batches == 1 rank x all pixels x the same
    assert j2["rays"] == j1["rays"] and j2["rays"]["primary"] == 192 * 128 * 32
    assert 0 < j2["rays"]["deposits"] <= j2["rays"]["shadow"] + j2["rays"]["extension"]
    a1, a2 = np.load(one), np.load(two)
    assert a1.shape == (192 * 128, 4) and a1[:, :3].mean() > 0
    assert np.array_equal(a1, a2), "tile-sharded render + reduce must equal the single-rank image bit for bit"
    assert j2["value"] > 0 and j2["image_mean_radiance"] == j1["image_mean_radiance"]


def test_bench_starts_its_own_ranks_and_strong_scaling_keeps_the_job_fixed(gpu, tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (fresh child processes) and relays rank
    0's line.  With --scaling strong the JOB is fixed -- in_flight x rounds samples per pixel of the whole image per step -- so two ranks
    trace exactly the paths one rank traces, half of the pixels each: same ray counts, same image bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    j2 = _run([sys.executable, "bench.py", "--gpus", "2", "--share-gpu", "--backend", "gloo", "--scaling", "strong", "--in-flight", "8",
               "--dump-accum", two] + COMMON, env)  # rounds 2: 16 samples per pixel per step, 16 in flight per rank
    j1 = _run([sys.executable, "bench.py", "--gpus", "1", "--scaling", "strong", "--in-flight", "16", "--dump-accum", one]
              + [a if a != "2" or COMMON[i - 1] != "--rounds" else "1" for i, a in enumerate(COMMON)], env)
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j1["scaling"] == "strong"
    assert j2["config"]["spp_per_step"] == 16 == j1["config"]["spp_per_step"] and j2["config"]["samples_in_flight"] == 16
    assert j2["rays"] == j1["rays"] and j2["rays"]["primary"] == 192 * 128 * 16
    assert np.array_equal(np.load(one), np.load(two))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_four_rank_job_on_one_gpu_equals_the_single_rank_job(gpu, tmp_path, scaling):
    """Four ranks on the test box's one GPU (the box admits six processes on its card; the driver's own scaling run uses N = 1, 2, 4, 8
    GPUs -- the eight-rank planning is rehearsed on the CPU, tests/test_multirank_gloo.py): interleaved 16x16 tiles over four ranks, the
    smallest share deciding the batch, the reduce and the aggregation over four processes."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    one, four = str(tmp_path / "one.npy"), str(tmp_path / "four.npy")
    j4 = _run([sys.executable, "bench.py", "--gpus", "4", "--share-gpu", "--backend", "gloo", "--scaling", scaling, "--in-flight", "4", "--dump-accum", four] + COMMON, env)
    if scaling == "weak":  # 4 ranks x 4 x 4 = 16 in flight on a quarter of the pixels, two batches per step: 32 samples per pixel
        j1 = _run([sys.executable, "bench.py", "--gpus", "1", "--in-flight", "16", "--dump-accum", one] + COMMON, env)
        assert j4["config"]["samples_in_flight"] == 16 and j4["rays"]["primary"] == 192 * 128 * 32
    else:  # the job is 4 x 2 = 8 samples per pixel whatever N is
        j1 = _run([sys.executable, "bench.py", "--gpus", "1", "--scaling", "strong", "--in-flight", "8", "--dump-accum", one]
                  + [a if a != "2" or COMMON[i - 1] != "--rounds" else "1" for i, a in enumerate(COMMON)], env)
        assert j4["config"]["spp_per_step"] == 8 == j1["config"]["spp_per_step"] and j4["rays"]["primary"] == 192 * 128 * 8
    assert j4["n_gpus"] == 4 and j4["scaling"] == scaling and j4["config"]["pixels_per_rank"] == 192 * 128 // 4
    assert j4["rays"] == j1["rays"]
    assert np.array_equal(np.load(one), np.load(four)), "tile-sharded render + reduce must equal the single-rank image bit for bit"


def test_oversized_jobs_and_overlapping_tiles_fail_with_a_message(gpu):
    """BASELINE config 5 sized naively -- 4K, a rank's eighth of the pixels, 2 048 samples in flight = 2.1 G queue entries,
    ~350 GB -- must be refused when the queues are set up, with a message that says what to change, not die in a later
    hipMalloc; tile rectangles that share a pixel are refused too (deposits are plain read-modify-writes that rely on one
    live path per pixel and plane)."""
    import bench
    from ptamd import scenes
    import gpu_util as U
    W4, H4 = 3840, 2160
    b = scenes.cornell_box(W4, H4)
    ctx = U.make_ctx(gpu, b, W4, H4, samples_in_flight=2048)
    ctx.set_tiles(bench.tile_rects(W4, H4, 0, 8))
    with pytest.raises(gpu.PtError, match="samples in flight"):
        ctx.render(1)
    with pytest.raises(gpu.PtError, match="overlaps"):
        ctx.set_tiles([(0, 0, 64, 64), (32, 32, 96, 96)])
    ctx.close()
    # what bench.py would actually run there fits: the same share at the in-flight count plan_in_flight allows
    owned = sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in bench.tile_rects(W4, H4, 0, 8))
    n = bench.plan_in_flight(256, 8, owned)
    assert 256 <= n < 2048 and n * owned <= bench.MAX_ENTRIES
    ctx = U.make_ctx(gpu, b, W4, H4, samples_in_flight=n)
    ctx.set_tiles(bench.tile_rects(W4, H4, 0, 8))
    ctx.render(n)
    st = ctx.stats()
    assert st["rays_generated"] == owned * n and ctx.samples_per_pixel == n
    ctx.close()
