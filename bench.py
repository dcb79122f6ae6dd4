#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the ray-queue render path on the ~1M-triangle two-level-BVH scene at
1080p (BASELINE.json metric), one process per GPU.

A *step* is one pass of the hot path over one batch: `512*N*R` samples per pixel for the pixels this rank
owns (N = ranks, R = --rounds, default 3: with the driver's 20 steps the timed region is ~11 s), i.e. R x [gen ->
4 x (intersect, shade, shadow intersect)] over ~1.06 G path segments in the first launches on every rank.  Every launch ends in a latency-bound tail of ~0.15 ms, so
large batches matter (64 / 128 / 256 samples in flight: 7.8 / 8.1 / 8.3 Grays/s in round 1; 256 / 512 / 768 on round 4's build, one box: 10 916 / 11 122 / 11 160
Mrays/s); 512 in flight keep ~200 GB of queues + accumulator planes resident (round 1-3: 256, ~100 GB), which is what 288 GB of HBM are for.  Image tiles (16x16, interleaved)
shard across ranks, every rank traces the same number of paths per step whatever N is (weak scaling: the image
simply receives N x more samples per step), and there is no data-path collective: the only exchange is ONE
RCCL reduce of the HDR accumulator at the end of the job (torch.distributed, backend nccl == RCCL), inside the
timed region.  A ray = one traceRay invocation on a live queue entry (extension or shadow), counted by the
device queues.

`--mode frame` (and the `frame` object of the default line) times what the reference publishes instead (BASELINE.md 1a:
35.6-56.8 ms per 1-spp 1280x720 frame, src/main.cpp:106-119): RayTracer::rayTrace = one sample per pixel + the
accumulate kernel, on the 82k-triangle mesh in the five-wall room.

`--gpus N` without WORLD_SIZE in the environment starts the N ranks itself (fresh child processes through
torch.distributed.run, before this process has touched the GPU) and relays rank 0's line; under torchrun it is a rank.
`--scaling strong` fixes the job instead of the per-rank load: `in_flight x rounds` samples per pixel of the WHOLE image per
step whatever N is, every rank tracing its 1/N of the pixels (time-to-image; the line says which mode ran).

Secondary objects of the N = 1 line (each its own short measurement on the same box, none inside the timed region of `value`):
`two_level` -- the same scene with every instance ENTERED instead of copied to world space (PT_FLAG_NO_BAKED_INSTANCES; and with
only the single-leaf meshes copied, PT_FLAG_TWO_LEVEL_ONLY), `dynamic` -- host and device time of one pt_upload_dynamic_async
per frame tick, `configs` -- BASELINE configs 2, 3 and 5 on this GPU, `frame` -- the 1-spp interactive frame.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the render stream in a separate,
profiled pass, for every kernel of the path (`roofline.kernels`); its top-level fields are those of the kernel
with the largest share of device time (k_trace<true>, the any-hit traversal of the shadow rays);
`cpu_baseline` times the oracle (our CPU restatement of the reference's path -- the reference has no CPU
traversal code) on this box's host cores over a bounded sample of the same frame.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what a device copy reaches on the box at hand is measured live
                       # (roofline.peak_measured_copy: 4.9 - 6.3 TB/s box to box)
BYTES_PER_EXT_RAY = 48  # closest-hit intersect reads 28 B ray, writes 20 B hit record
BYTES_PER_SHADOW_RAY = 44  # 28 B ray + 12 B contribution + 4 B pixel
BYTES_PER_DEPOSIT = 24  # 12 B read + 12 B written per accumulator update (unoccluded shadow ray, emissive hit, sky miss)
IN_FLIGHT = 512  # samples in flight per pixel at 1080p on one rank (x N on 1/N of the pixels)
MAX_ENTRIES = 1920 * 1080 * IN_FLIGHT  # path segments resident per rank (~200 GB of queues and planes): the 1080p job at any N; caps 4K
BYTES_PER_SHADED_HIT = 160  # read 28 + 20 + 20, write 48 + 44
BYTES_PER_QUEUE_ENTRY = 164  # resident per queue entry with queues as large as the batch (rounds 1-5): two extension queues (2 x 48 B), the shadow queue (48 B), the hit record (20 B)
# Round 6: the queues that only hold what a batch's FIRST pass emits are sized by what it emits (pt_config.ext_queue_fraction / shadow_queue_fraction): on config 4
# 26 % of a batch's entries go on after the first hit and 58 % spawn a shadow ray there (measured by the library itself, roofline.first_pass_ratios of the line); with
# these fractions a camera-ray batch of a pinhole keeps 36 + 80 x 0.30 + 48 x 0.62 = 90 B per entry resident instead of 164 (a thin lens: 68 + 48 x 0.30 + 48 x 0.62 =
# 112: its camera rays keep their origins).  The library cuts a batch that would emit more to what fits -- never a wrong image, at worst smaller batches.
EXT_QUEUE_FRACTION = 0.30
SHADOW_QUEUE_FRACTION = 0.62
TILE = int(os.environ.get("PTAMD_TILE", "16"))  # edge of the image tiles dealt to the ranks of an N-GPU job.  16 since round 5: every rank's share of the N = 8 job emulated on one
# GPU (tools/rank_emul.py), slowest rank over fastest: 32 x 32 tiles 1.037 (7.70 x predicted), 16 x 16 1.024 (7.80 x), 8 x 8 1.015 (7.84 x); the variable: diagnostics
BYTES_PER_GEN_RAY = 32  # origin + pixel, direction + state; a primary ray's throughput is 1 and is not stored


def tile_rects(width, height, rank, world, tile=None):
    tile = tile or TILE
    rects, t = [], 0
    for y in range(0, height, tile):
        for x in range(0, width, tile):
            if t % world == rank:
                rects.append((x, y, min(x + tile, width), min(y + tile, height)))
            t += 1
    return rects


def usable_cores():
    """Host threads this process may really use: affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def _cpu_list(path):
    cpus = set()
    for part in open(path).read().strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_memory_node(local_rank=0):
    """The memory node the rank's GPU hangs off, from the KFD topology (the compute nodes this process may open, in order; location_id = bus << 8 | devfn)
    and the PCI device's numa_node; None where that cannot be read.  No HIP call: this runs before anything touches the GPU."""
    try:
        gpus = []
        for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*"), key=lambda p: int(os.path.basename(p))):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(d, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        if not gpus:
            return None
        g = gpus[local_rank] if local_rank < len(gpus) else gpus[0]
        loc, dom = int(g["location_id"]), int(g.get("domain", "0"))
        node = int(open(f"/sys/bus/pci/devices/{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}/numa_node").read())
        return node if node >= 0 else None
    except Exception:
        return None


def pin_to_one_l3_domain(local_rank=0):
    """The application's share of thread placement: this process (and every thread it starts from here on -- the host library's worker pool, the HIP
    runtime's helpers) is restricted to the CPUs that share ONE last-level cache.  On the two-socket, sixteen-domain hosts of this pool the scheduler
    otherwise spreads a loop's eight threads over domains and sockets: a rebuilt-tree tick took 4.7-5.7 ms unpinned against 3.9-4.2 pinned, `Mesh()`
    1.6-2.2 against 1.0-1.15 (EXPERIMENTS.md, round 6).  N = 1: the quietest domain of the memory node the GPU hangs off (else: this process runs on; two /proc/stat samples 50 ms
    apart); N > 1: the ranks of a node spread evenly over all its domains in order, so that no two share one.  Returns what it did, for the bench line; PTAMD_BENCH_PIN=0: nothing."""
    if os.environ.get("PTAMD_BENCH_PIN", "1") in ("0", "off") or not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "why": "disabled"}
    try:
        allowed = os.sched_getaffinity(0)
        here = C.CDLL(None).sched_getcpu()
        gpu_node = gpu_memory_node(local_rank)  # (the GPU's side of the machine: 1-spp frames are 1.5-2 % faster from there than from the other socket)
        node = None
        if gpu_node is not None and os.path.exists(f"/sys/devices/system/node/node{gpu_node}/cpulist"):
            node = _cpu_list(f"/sys/devices/system/node/node{gpu_node}/cpulist")
            if not (node & allowed):
                node = None
        if node is None:
            node = next((_cpu_list(os.path.join(d, "cpulist")) for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
                         if here in _cpu_list(os.path.join(d, "cpulist"))), allowed)
        domains = {}
        for c in sorted(allowed & node):
            d = frozenset(_cpu_list(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list") & allowed)
            if d:
                domains.setdefault(min(d), d)
        if len(domains) < 2:
            return {"pinned": False, "why": "one last-level cache domain"}

        def busy():
            out = {}
            for line in open("/proc/stat"):
                f = line.split()
                if f[0].startswith("cpu") and f[0] != "cpu":
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = sum(v) - v[3] - v[4]  # everything but idle and iowait
            return out
        keys = sorted(domains)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        if local_world > 1 and gpu_node is not None:
            pick = keys[local_rank % len(keys)]  # a domain of the rank's GPU's memory node: ranks whose GPUs share a node take different ones
        elif local_world > 1:
            # the ranks of a node spread evenly over ALL its domains, in order (8 ranks on 16 domains: every second one -- ranks 0-3 on the first socket,
            # 4-7 on the second, as the GPUs usually are)
            every = {}
            for c in sorted(allowed):
                d = frozenset(_cpu_list(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list") & allowed)
                if d:
                    every.setdefault(min(d), d)
            domains, keys = every, sorted(every)
            pick = keys[(local_rank * max(1, len(keys) // local_world)) % len(keys)]
        else:
            b0 = busy()
            time.sleep(0.05)
            b1 = busy()
            pick = min(keys, key=lambda k: sum(b1.get(c, 0) - b0.get(c, 0) for c in domains[k]))
        os.sched_setaffinity(0, domains[pick])
        return {"pinned": True, "cpus": len(domains[pick]), "first_cpu": int(pick), "domains_on_node": len(keys), "gpu_memory_node": gpu_node, "node_cpus": sorted(allowed & node),
                "what": "process restricted to one last-level cache domain (bench.py pin_to_one_l3_domain); the CPU baseline's threads run on every CPU of that "
                        "domain's memory node (where the process allocated the scene)"}
    except Exception as e:  # a host without these files: run where the scheduler puts us
        return {"pinned": False, "why": str(e)[:120]}


def reference_kernels_baseline(O, sc, bundle, width, height, seconds=3.0):
    """Secondary figure: assets/cl/kernel.cl itself (oracle/_ref, serial NDRange, LFSR113 streams) on a 1/10-scale
    frame of the same scene and camera.  Its traversal stack is the reference's 32 entries per ray, so it is only
    run when the scene's trees are shallower than that."""
    if not O.have_ref():
        return None
    flat, depth = bundle.flat, 0
    for root in {int(r) for r in flat.top_nodes[flat.top_nodes["isLeaf"] != 0]["a"]}:
        todo = [(root, 1)]
        while todo:
            i, dpt = todo.pop()
            depth = max(depth, dpt)
            if flat.sub_nodes[i]["count"] == 0:
                left = int(flat.sub_nodes[i]["left"])
                todo += [(left, dpt + 1), (left + 1, dpt + 1)]
    if depth + 8 >= 32:
        return {"skipped": f"bottom-level BVH depth {depth}: too deep for the reference's 32-entry traversal stack"}
    w, h = width // 10, height // 10
    st, streams = O.QueueState(w, h, (w * h + 63) // 64 * 64), O.create_streams((w * h + 63) // 64 * 64, use_ref=True)
    rays, t_used, spp = 0, 0.0, 0
    while t_used < seconds:
        t0 = time.perf_counter()
        trace, _ = O.trace_rays("ref", sc, bundle.camera, st, streams)
        t_used += time.perf_counter() - t0
        rays += int(trace[:, 0].sum())  # numInRays of every pass = extension rays traced (shadow rays not counted)
        spp += 1
    return {"value": round(rays / t_used / 1e6, 4), "unit": "M extension rays/s", "cores": 1, "kind": "reference",
            "sample": f"{spp} spp of a {w}x{h} frame, {rays} extension rays in {t_used:.1f} s, reference kernels -O1, serial"}


def cpu_baseline(bundle, seconds, width, height):
    """Oracle (kind 'port') on all host cores: 1 spp over 8x8 pixel blocks spread over the frame.  One warm-up pass (page-in, thread
    pool, the rate estimate that sizes the chunks) is excluded; then `seconds` of wall time in three equal segments of equally sized
    chunks, whose rates give the spread stated next to the value."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib as O  # checker / reported baseline only
    O.build(fast=True, ref=False)  # -O3 -march=native for THIS host
    cores = int(os.environ.get("PTAMD_BENCH_CPU_THREADS", "0")) or usable_cores()
    sc = O.BoundScene(bundle.flat, sky=bundle.sky, material_textures=bundle.material_textures)
    bx, by = width // 8, height // 8
    order = np.random.default_rng(0).permutation(bx * by)
    yy, xx = np.mgrid[0:8, 0:8]
    state = {"pos": 0, "sample": 0}
    work = {"innerSteps": 0, "triangleTests": 0, "topVisits": 0}  # reference-layout traversal work (SURVEY 8d, L2-level model)

    def run(nblocks, count_work=True):
        if state["pos"] + nblocks > len(order):  # the whole frame is done: next sample index
            state["pos"], state["sample"] = 0, state["sample"] + 1
        blocks = order[state["pos"]:state["pos"] + nblocks]
        state["pos"] += len(blocks)
        px = (((blocks // bx)[:, None, None] * 8 + yy) * width + (blocks % bx)[:, None, None] * 8 + xx).reshape(-1)
        t0 = time.perf_counter()
        _, cnt = O.render(sc, bundle.camera, width, height, 1, seed=1, first_sample=state["sample"], pixels=px.astype(np.uint32),
                          threads=cores, fast=True)
        dt = time.perf_counter() - t0
        if count_work:
            for k in work:
                work[k] += cnt[k]
        return cnt["raysExtension"] + cnt["raysShadow"], len(px), dt

    rays_w, _, dt_w = run(1024, count_work=False)  # warm-up, excluded
    chunk = int(min(8192, max(256, 1024 * 0.5 / max(dt_w, 1e-3))))  # ~0.5 s of work per call
    segments, rays, pixels_done, t_used = [], 0, 0, 0.0
    for _ in range(3):
        r_seg, t_seg = 0, 0.0
        while t_seg < seconds / 3.0:
            r, npx, dt = run(chunk)
            r_seg, t_seg, pixels_done = r_seg + r, t_seg + dt, pixels_done + npx
        segments.append(r_seg / t_seg / 1e6)
        rays, t_used = rays + r_seg, t_used + t_seg
    ref = None
    try:  # the reference's own kernels, compiled for the host and run one work-item at a time (oracle/_ref): 1 core
        ref = reference_kernels_baseline(O, sc, bundle, width, height)
    except Exception as e:  # never let the secondary figure take the bench down
        ref = {"error": str(e)[:200]}
    value = rays / t_used / 1e6
    return {"value": round(value, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "segments_mrays_per_s": [round(x, 3) for x in segments],
            "spread": round((max(segments) - min(segments)) / value, 4),
            "reference_kernels": ref,
            "per_ray": {k: round(work[k] / max(rays, 1), 2) for k in ("innerSteps", "triangleTests", "topVisits")},
            "sample": f"{pixels_done} pixel-samples (random 8x8 blocks of the {width}x{height} frame, 1 spp each pass; one warm-up pass of "
                      f"65 536 pixel-samples excluded), {rays} rays in {t_used:.1f} s (three segments), oracle -O3 -march=native, {cores} threads"}


def plan_in_flight(requested, world, owned_pixels, max_entries=MAX_ENTRIES):
    """Samples in flight per pixel on one rank: `requested` x ranks (weak scaling: every rank keeps the same number of path
    segments resident whatever N is), bounded by the library's 4096 planes and by the entries that fit in HBM -- and a MULTIPLE OF 256
    (a power of two below that): the library keeps up to 256 samples of a pixel next to each other in its queues, which is what the
    bundle kernel and the coherent first bounce live on, only when the batch divides that way.  (Round 3 returned 2 046 for an eighth of
    the 1080p frame -- the 32 x 32 tiles leave the ranks' shares uneven by a fraction of a percent, 259 328 instead of 259 200 pixels --
    and a batch of 2 046 = 2 x 1 023 keeps only TWO samples of a pixel together: the emulated rank traced 7 100 instead of 10 600
    Mrays/s, profiles/round4/r4e_rank_emul_config4_weak.txt.  The budget now carries 1.5 % of slack for the uneven shares.)"""
    fit = (max_entries + max_entries // 64) // max(owned_pixels, 1)
    n = max(1, min(requested * world, 4096, fit))
    if n >= 256:
        n -= n % 256
    elif n >= 2:
        n = 1 << (n.bit_length() - 1)
    return n


def bytes_per_entry(ext_fraction=1.0, shadow_fraction=1.0, thin_lens=False):
    """Resident bytes per entry of a batch (ensureQueues' accounting, csrc/ptamd.hip): camera-ray directions 16 + hit record 20 for every entry; the first queue's
    origin and throughput planes (32) for every entry of a thin lens / for the fraction that goes on with a pinhole's bundles; the second queue (48) for the
    fraction that goes on; the shadow queue (48) for the fraction that spawns a shadow ray."""
    fe = ext_fraction if 0.0 < ext_fraction < 1.0 else 1.0
    fs = shadow_fraction if 0.0 < shadow_fraction < 1.0 else 1.0
    return 36.0 + 32.0 * (1.0 if thin_lens or fe == 1.0 else fe) + 48.0 * fe + 48.0 * fs


def resident_bytes(in_flight, owned_pixels, per_entry=BYTES_PER_QUEUE_ENTRY):
    """HBM the queues and the extra accumulator planes of one context keep resident (what ensureQueues, csrc/ptamd.hip, checks against hipMemGetInfo)."""
    return int(owned_pixels * in_flight * per_entry) + max(in_flight - 1, 0) * owned_pixels * 16


def shrink_in_flight(n):
    """The next smaller batch the library keeps coherent: multiples of 256 down to 256, then powers of two."""
    if n > 256:
        return n - 256 if n % 256 == 0 else n - n % 256
    return max(1, n // 2)


def fit_in_flight(in_flight, owned_pixels, free_bytes, reserve=6 << 30, per_entry=BYTES_PER_QUEUE_ENTRY):
    """`in_flight`, lowered until its queues and planes fit what the device has free (another tenant on the card, a context that is not
    freed yet, a part with less HBM): the bench line degrades by a percent or two (512 -> 256 in flight: -1.9 %) instead of failing.  `reserve`
    covers the scene, the sky, the spill region and the accumulator."""
    while in_flight > 1 and resident_bytes(in_flight, owned_pixels, per_entry) + reserve > free_bytes:
        in_flight = shrink_in_flight(in_flight)
    return in_flight


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reduce_accumulator(dist, accum, backend, world):
    """The one exchange step of the job: SUM the HDR accumulator onto rank 0 (RCCL over xGMI; gloo rehearsals reduce on the
    host).  Runs on the CURRENT torch stream -- the stream the render was enqueued on -- so it is ordered after the kernels."""
    if world == 1:
        return
    if backend == "nccl":
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    else:
        host = accum.cpu()  # synchronises with the current stream
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        accum.copy_(host)


def aggregate(dist, torch, counts, elapsed, backend, world, device="cuda"):
    """Whole-job figures: ray counters summed over ranks, elapsed time = the slowest rank's."""
    dev = device if backend == "nccl" else "cpu"
    c = torch.tensor([float(x) for x in counts], dtype=torch.float64, device=dev)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in c.tolist()], float(t.item())


def frame_times(D, H, scenes, L, device, frames, width=1280, height=720, only=""):
    """What the reference publishes: wall ms per interactive frame = RayTracer::rayTrace, one sample per
    pixel through the whole queue loop + the accumulate kernel into a device image (the reference writes a GL texture),
    stream synchronised every frame as its queue.finish() does (src/raytracer.cpp:88-151, src/main.cpp:106-119)."""
    mats = {"glass (refractive 0.9, ior 1.5, SBVH)": L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0),
            "copper (PBR metal 0.8)": L.material_pbr_metal((0.955, 0.638, 0.538), 0.8),
            "ceramic (PBR dielectric 0.7)": L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7),
            "diffuse 0.8": L.material_diffuse((0.8, 0.8, 0.8))}
    out = {}
    for name, mat in mats.items():
        if only and only not in name:
            continue
        b = scenes.blob_room(width, height, material=mat, builder=H.BVH_SPATIAL_SPLIT, level=6)
        ctx = D.Context(width, height, seed=1, device=device, samples_in_flight=1)
        ctx.upload_scene(b.flat, sky=b.sky, material_textures=b.material_textures)
        ctx.set_camera(b.camera)
        for _ in range(20):
            ctx.render(1, sync=False)
            ctx.resolve_device()
        ctx.synchronize()
        ctx.reset_stats()
        t0 = time.perf_counter()
        for _ in range(frames):
            ctx.render(1, sync=False)
            ctx.resolve_device()
            ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        out[name] = {"ms_per_frame": round(dt / frames * 1e3, 4), "rays_per_frame": int((st["rays_extension"] + st["rays_shadow"]) / frames),
                     "mrays_per_s": round((st["rays_extension"] + st["rays_shadow"]) / dt / 1e6, 1),
                     "team_kernel_launches_per_frame": round(st["team_launches"] / frames, 2)}  # of the 8 traversal launches of a frame: four lanes per ray (pt_team.h)
        ctx.close()
    return {"what": f"RayTracer::rayTrace: 1 spp of a {width}x{height} frame + accumulate kernel, synchronised per frame; 81 920-triangle mesh "
                    f"(SBVH) in the five-wall room with an area light, {frames} frames each",
            "scenes": out, "reference_published_ms_per_frame": "35.6 - 56.8 (images/*.png overlays, hardware not stated; BASELINE.md 1a)"}


def measure_scene(D, bundle, W, Hh, device, in_flight, flags=0, steps=3, warmup=1, rounds=1, what="", queue_fractions=(0.0, 0.0)):
    """Throughput of one scene / flag set on this GPU, the way the headline is measured (warm-up, clear, K steps between two
    synchronisations), plus the per-kernel-family device times of one extra, profiled step.  Its own context, closed before it returns."""
    ctx = D.Context(W, Hh, seed=1, device=device, samples_in_flight=in_flight, flags=flags, ext_queue_fraction=queue_fractions[0], shadow_queue_fraction=queue_fractions[1])
    try:
        t0 = time.perf_counter()
        ctx.upload_scene(bundle.flat, sky=bundle.sky, material_textures=bundle.material_textures)
        ctx.synchronize()
        upload_s = time.perf_counter() - t0
        ctx.set_camera(bundle.camera)
        spp = in_flight * rounds
        for _ in range(warmup):
            ctx.render(spp, sync=False)
        ctx.synchronize()
        ctx.clear()
        ctx.reset_stats()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.render(spp, sync=False)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        rays = st["rays_extension"] + st["rays_shadow"]
        ctx.reset_stats()
        ctx.profile_kernels(True)
        ctx.render(spp, sync=True)
        ps = ctx.stats()
        ctx.profile_kernels(False)
        pr = ps["rays_extension"] + ps["rays_shadow"]
        primary = ps["rays_generated"] if ps["packet_launches"] > 0 else 0  # (camera rays through the packet / bundle kernel; otherwise they are per-ray kernel work)
        per_ray_ms = ps["ms_intersect"] - ps["ms_packet"]
        out = {"what": what, "mrays_per_s": round(rays / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
               "spp_per_step": spp, "samples_in_flight": in_flight, "rays_per_step": int(rays / steps),
               "rays_per_primary": round(rays / max(st["rays_generated"], 1), 3),
               "packet_kernel_for_primary_rays": ps["packet_launches"] > 0, "scene_upload_s": round(upload_s, 3),
               "kernel_ms_per_step": {"gen": round(ps["ms_gen"], 2), "closest_hit": round(ps["ms_intersect"], 2),
                                      "of_which_packet": round(ps["ms_packet"], 2), "shade": round(ps["ms_shade"], 2),
                                      "any_hit": round(ps["ms_shadow"], 2)},
               # rays per class and the rate INSIDE the kernels that trace them (device ms of the profiled step): the whole-job figure above is a mix of these
               "rays_by_class": rays_by_class(ps),
               "batch_samples": ps["batch_samples"],  # samples per pixel of the batches the library cut (< samples_in_flight where the queue fractions made it)
               "first_pass_ratios": {"ext": round(ps["first_pass_ext_ratio"], 4), "shadow": round(ps["first_pass_shadow_ratio"], 4)},
               "instances": {"entered": ps["entered_instances"], "folded": ps["folded_instances"], "general_route": bool(ps["general_route"])},
               "per_ray_kernels_mrays_per_s": round((pr - primary) / max(per_ray_ms + ps["ms_shadow"], 1e-9) / 1e3, 1),  # bounce + shadow rays over their two kernels' time
               "shade_ns_per_entry": round(ps["ms_shade"] * 1e6 / max(ps["shade_hits"], 1), 3),
               "mrays_per_s_profiled_step": round(pr / max(ps["ms_last_render"], 1e-6) / 1e3, 1)}
        return out
    finally:
        ctx.close()


def rays_by_class(ps):
    """{camera, bounce, shadow}: rays of a profiled step and the rate inside the kernel family that traces them (pt_stats of a pass with per-kernel events)."""
    packets = ps["packet_launches"] > 0
    camera = ps["rays_generated"]
    bounce = ps["rays_extension"] - camera
    ms_cam = ps["ms_packet"] if packets else 0.0
    ms_bounce = ps["ms_intersect"] - ms_cam
    out = {"camera": {"rays": int(camera), "kernel": "k_trace_multi / k_trace_packet" if packets else "k_trace<false> (with the bounce rays)",
                      "in_kernel_mrays_per_s": round(camera / ms_cam / 1e3, 1) if ms_cam > 0 else None},
           "bounce": {"rays": int(bounce), "kernel": "k_trace<false>",
                      "in_kernel_mrays_per_s": round((bounce if packets else bounce + camera) / ms_bounce / 1e3, 1) if ms_bounce > 0 else None},
           "shadow": {"rays": int(ps["rays_shadow"]), "kernel": "k_trace<true>",
                      "in_kernel_mrays_per_s": round(ps["rays_shadow"] / ps["ms_shadow"] / 1e3, 1) if ps["ms_shadow"] > 0 else None}}
    return out


def _torus_mesh(H, L, nu, nv, material, builder):
    """2 * nu * nv triangles on a lumpy torus (no poles: every vertex has six neighbours), for the deforming-mesh figures."""
    u, v = np.meshgrid(np.arange(nu) * (2 * np.pi / nu), np.arange(nv) * (2 * np.pi / nv), indexing="ij")
    r = 0.16 * (1.0 + 0.25 * np.sin(5 * u) * np.cos(3 * v))
    p = np.stack([(0.42 + r * np.cos(v)) * np.cos(u), r * np.sin(v), (0.42 + r * np.cos(v)) * np.sin(u)], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a, b, c, d = (i * nv + j).ravel(), (((i + 1) % nu) * nv + j).ravel(), (((i + 1) % nu) * nv + (j + 1) % nv).ravel(), (i * nv + (j + 1) % nv).ravel()
    f = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)]).astype(np.uint32)
    return H.Mesh(p, f, [material], builder=builder), p


def dynamic_update_times(D, H, L, scenes, bundle, W, Hh, device, ticks=12):
    """RayTracer::frameTick's cost (src/raytracer.cpp:183-189,497-595) with EVERY instance moved a little each tick.  Per tick, apart:
    `host_flatten_ms` -- the scene-graph walk, the lights and the top-level BVH build inside the host library (flattenDynamic: what
    RayTracer::frameTick does before it calls the device library); `host_upload_ms` -- the time pt_upload_dynamic_async blocks the calling
    thread (top level collapsed to 4-wide nodes, instance table, lights, copy jobs; a few KB into pinned staging); `ms_until_adopted` --
    flatten + upload + pt_frame_tick + a synchronisation (the world-space copies of the instances are made by the device meanwhile),
    nothing rendering.  The benchmark scene (14 instances) with its instances copied to world space (the default) and entered
    (PT_FLAG_NO_BAKED_INSTANCES), then 1 000 and 10 000 instances of a 320-triangle mesh.  Rigid motion only: a deforming mesh is the
    `refit` object (that is what the reference's 4-6 ms per frame contain)."""
    out = {}

    def run(name, b, flags, moved_nodes, what):
        scene, flat = b.scene, b.flat
        ctx = D.Context(W, Hh, seed=1, device=device, samples_in_flight=1, flags=flags)
        try:
            ctx.upload_scene(flat, sky=None)
            ctx.set_camera(b.camera)
            ctx.render(1)
            rng = np.random.default_rng(7)
            base = rng.uniform(-1, 1, (len(moved_nodes), 3)).astype(np.float32)
            flatten_ms, upload_ms, total_ms = [], [], []
            for k in range(ticks + 2):
                for n, (node, loc) in enumerate(moved_nodes):  # (driving the scene graph is the application's business: not timed)
                    scene.set_transform(node, location=(loc[0] + 0.02 * k * base[n, 0], loc[1], loc[2] + 0.02 * k * base[n, 2]))
                t0 = time.perf_counter()
                f2, lib_s = scene.flatten_dynamic(flat)
                t1 = time.perf_counter()
                ctx.upload_dynamic_async(f2)
                t2 = time.perf_counter()
                ctx.frame_tick()
                ctx.synchronize()
                t3 = time.perf_counter()
                if k >= 2:  # the first two ticks size the two buffer sets
                    flatten_ms.append(lib_s * 1e3)
                    upload_ms.append((t2 - t1) * 1e3)
                    total_ms.append((t3 - t0) * 1e3 - ((t1 - t0) - lib_s) * 1e3)  # (the binding's array copies out of the host library are not the library's)
            ctx.render(1)  # the last state renders
            out[name] = {"what": what, "instances": int(f2.num_instances), "top_level_nodes": int(len(f2.top_nodes)), "instanced_triangles": int(f2.instanced_triangles),
                         "host_flatten_ms": round(float(np.median(flatten_ms)), 3), "host_upload_ms": round(float(np.median(upload_ms)), 3),
                         "ms_until_adopted": round(float(np.median(total_ms)), 3), "host_upload_ms_min_max": [round(min(upload_ms), 3), round(max(upload_ms), 3)]}
        finally:
            ctx.close()

    # the benchmark scene: nodes 0 / 1 are the ground and the light, 2 .. the mesh instances (scenes.instanced_grid adds them in this order)
    grid = [(2 + k, ((k % 4 - 1.5) * 1.5, 0.62 * 1.2, (k // 4 - 1.0) * 1.5)) for k in range(12)]
    run("benchmark_scene_copied", bundle, 0, grid, "config 4's scene, every mesh instance moved each tick; instances copied to world space by two device kernels per tick")
    run("benchmark_scene_entered", bundle, D.FLAG_NO_BAKED_INSTANCES, grid, "the same with every instance entered at traversal: nothing to copy")
    for n in (1000, 10000):
        field = scenes.instance_field(W, Hh, n=n, level=2)
        side = 0.45 * n ** 0.5
        rng = np.random.default_rng(9)
        moved = [(2 + k, (float(rng.uniform(-side, side)), 0.5, float(rng.uniform(-side, side)))) for k in range(n)]
        run(f"instances_{n}", field, 0, moved, f"{n} instances of a 320-triangle mesh, all moved each tick (top level: top-down SAH above 256 instances, "
                                                "the reference's agglomerative clustering -- O(n^2) -- up to there)")
    out["what"] = f"medians over {ticks} ticks, every instance moved each tick; see the entries"
    out["reference_ms_per_frame"] = ("4.1 - 6.3 per frame for ONE deforming 36.5 k-triangle mesh (CPU refit + upload of the dynamic data; lab report Table 2, RX 480): "
                                     "compare with the `refit` object -- rigid instance motion is not timed apart by the reference")
    return out


def dynamic_refit_times(D, H, L, scenes, W, Hh, device, ticks=10):
    """A deforming mesh per frame, the reference's way (src/model/mesh_sequence.cpp:81-97 + src/bvh/refit_bvh.cpp:6-34 on the host,
    transferDynamicData src/raytracer.cpp:510-568): per tick Mesh::refit (new vertex positions, smooth normals regenerated, boxes refitted
    bottom-up -- host library), the scene flattened again, pt_update_geometry (the caller's vertices and nodes through pinned staging; packed nodes
    re-quantised and triangle records re-made by two device kernels on the copy stream), pt_upload_dynamic_async, pt_frame_tick
    and a synchronisation.  Host ms per stage and ms until the new geometry is adopted; the mesh in the five-wall room, SBVH."""
    out = {}
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    for name, make in (("mesh_36k", lambda: _torus_mesh(H, L, 135, 135, mat, H.BVH_SPATIAL_SPLIT)),
                       ("mesh_82k", lambda: (scenes.blob_mesh(mat, level=6, builder=H.BVH_SPATIAL_SPLIT), None))):
        mesh, p0 = make()
        if p0 is None:
            verts, _ = mesh.geometry()
            p0 = verts["vertex"][:, :3].copy()
        scene = H.Scene()
        mb = scenes._MeshBuilder()
        mats = scenes._room_materials()
        scenes._room(mb, mats)
        scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
        scene.add_node(mesh, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
        flat = scene.flatten()
        cam = scenes.blob_room(W, Hh, level=2).camera
        ctx = D.Context(W, Hh, seed=1, device=device, samples_in_flight=1)
        try:
            ctx.upload_scene(flat, sky=None)
            ctx.set_camera(cam)
            ctx.render(1)
            first_vertex, _ = scene.mesh_offsets(mesh)

            def deformed(k):
                ang = 0.15 * np.sin(0.7 * (k + 1)) * p0[:, 1] * 4.0
                return np.stack([np.cos(ang) * p0[:, 0] - np.sin(ang) * p0[:, 2], p0[:, 1] * (1.0 + 0.05 * np.sin(k + 1.0)),
                                 np.sin(ang) * p0[:, 0] + np.cos(ang) * p0[:, 2]], 1).astype(np.float32)

            # (a) round 5: the mesh's vertices alone travel (pt_refit_vertices); the device refits its trees bottom-up, the host library leaves its own
            # boxes to whoever asks for them (Mesh::refit = positions + smooth normals on the library's worker threads)
            stage = {k: [] for k in ("mesh_refit", "refit_vertices", "flatten_dynamic", "upload_dynamic", "total")}
            for k in range(ticks + 2):
                p = deformed(k)
                t0 = time.perf_counter()
                mesh.refit(p)
                t1 = time.perf_counter()
                ctx.refit_vertices(first_vertex, mesh.vertices_view())
                t2 = time.perf_counter()
                dyn, _ = scene.flatten_dynamic_only()
                t3 = time.perf_counter()
                ctx.upload_dynamic_async(dyn)
                t4 = time.perf_counter()
                ctx.frame_tick()
                ctx.synchronize()
                t5 = time.perf_counter()
                if k >= 2:
                    for key, dt in (("mesh_refit", t1 - t0), ("refit_vertices", t2 - t1), ("flatten_dynamic", t3 - t2), ("upload_dynamic", t4 - t3), ("total", t5 - t0)):
                        stage[key].append(dt * 1e3)
            ctx.render(1)
            # (b) the reference's way, as rounds 3-4 timed it: boxes refitted on the host, the scene flattened, whole vertex and node arrays handed over
            old = {k: [] for k in ("mesh_refit_with_boxes", "flatten", "update_geometry", "upload_dynamic", "total")}
            for k in range(ticks // 2 + 2):
                p = deformed(k + 100)
                t0 = time.perf_counter()
                mesh.refit(p)
                flat = scene.flatten()  # (asks for the nodes: the host refits its boxes here)
                t2 = time.perf_counter()
                ctx.update_geometry(flat)
                t3 = time.perf_counter()
                ctx.upload_dynamic_async(flat)
                t4 = time.perf_counter()
                ctx.frame_tick()
                ctx.synchronize()
                t5 = time.perf_counter()
                if k >= 2:
                    for key, dt in (("flatten", t2 - t0), ("update_geometry", t3 - t2), ("upload_dynamic", t4 - t3), ("total", t5 - t0)):
                        old[key].append(dt * 1e3)
            ctx.render(1)
            st = mesh.stats()
            out[name] = {"triangles": int(st["num_input_triangles"]), "triangle_references": int(st["num_triangle_refs"]), "sub_bvh_nodes": int(st["num_nodes"]),
                         "host_ms": {k: round(float(np.median(v)), 3) for k, v in stage.items() if k != "total"},
                         "ms_until_adopted": round(float(np.median(stage["total"])), 3), "ms_until_adopted_min_max": [round(min(stage["total"]), 3), round(max(stage["total"]), 3)],
                         "host_refitted_nodes_route": {"host_ms": {k: round(float(np.median(v)), 3) for k, v in old.items() if k != "total" and v},
                                                       "ms_until_adopted": round(float(np.median(old["total"])), 3),
                                                       "what": "Mesh::refit + scene flatten (boxes refitted on the host when the nodes are asked for) + pt_update_geometry: rounds 3-4's tick"}}
        finally:
            ctx.close()
    # the other branch of MeshSequence::buildBvh (src/model/mesh_sequence.cpp:81-97): a REBUILT tree per frame (fast binned builder, longest axis) --
    # a new mesh, the scene flattened, pt_upload_static + pt_upload_dynamic (everything converted and uploaded again, the render stream synchronised)
    try:
        v, f = scenes.icosphere(5)
        p0 = (v * 0.5).astype(np.float32)
        f = f.astype(np.uint32)
        mb = scenes._MeshBuilder()
        mats = scenes._room_materials()
        scenes._room(mb, mats)
        room = mb.build(mats, H.BVH_BINNED_SAH)
        cam = scenes.blob_room(W, Hh, level=2).camera
        ctx = D.Context(W, Hh, seed=1, device=device, samples_in_flight=1)
        try:
            def new_scene(k):
                p = (p0 * (1.0 + 0.1 * np.sin(k + 1.0 + 5.0 * p0[:, :1]))).astype(np.float32)
                t0 = time.perf_counter()
                mesh = H.Mesh(p, f, [mat], builder=H.BVH_BINNED_FAST)
                t1 = time.perf_counter()
                scene = H.Scene()
                scene.add_node(room)
                scene.add_node(mesh, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
                flat = scene.flatten()
                return flat, t0, t1, time.perf_counter()

            flat, _, _, _ = new_scene(0)
            ctx.upload_scene(flat, sky=None)
            ctx.set_camera(cam)
            ctx.render(1)
            # (a) round 5: pt_upload_static_async -- the rebuilt scene is converted and copied beside the one that renders, frames keep coming, the tick adopts it
            build_ms, flatten_ms, upload_ms, wait_ms, total_ms = [], [], [], [], []
            for k in range(1, 12):
                ctx.render(1, sync=False)  # a frame of the old scene is in flight while the host builds
                flat, t0, t1, t2 = new_scene(k)
                ctx.render(1, sync=False)  # ... and another one while the host converts and the copies run (rounds 5-6a enqueued it after the conversion)
                t2b = time.perf_counter()
                ctx.upload_static_async(flat)
                ctx.upload_dynamic_async(flat)
                t3 = time.perf_counter()
                ctx.frame_tick()
                ctx.render(1, sync=False)  # the first frame of the new scene
                ctx.synchronize()
                t4 = time.perf_counter()
                if k >= 3:
                    build_ms.append((t1 - t0) * 1e3), flatten_ms.append((t2 - t1) * 1e3), upload_ms.append((t3 - t2b) * 1e3), wait_ms.append((t4 - t3) * 1e3), total_ms.append((t4 - t0) * 1e3)
            out["rebuild_20k"] = {"triangles": int(len(f)), "host_ms": {"mesh_build_fast_binned": round(float(np.median(build_ms)), 3), "flatten": round(float(np.median(flatten_ms)), 3),
                                                                        "upload_static_async_and_dynamic_async": round(float(np.median(upload_ms)), 3),
                                                                        "tick_first_new_frame_and_synchronize": round(float(np.median(wait_ms)), 3)},
                                  "ms_until_adopted": round(float(np.median(total_ms)), 3), "ms_until_adopted_min_max": [round(min(total_ms), 3), round(max(total_ms), 3)],
                                  "what": "a rebuilt tree per frame: new Mesh (fast binned builder), flatten, pt_upload_static_async + pt_upload_dynamic_async, pt_frame_tick; "
                                          "three 1-spp frames rendered meanwhile (two of the old scene -- one enqueued before the build, one before the conversion -- and one "
                                          "of the new) and INCLUDED in ms_until_adopted; the render stream is never synchronised by the upload; 20 480 triangles"}
            # (b) rounds 3-4: pt_upload_static + pt_upload_dynamic (everything converted, the render stream synchronised, the dynamic state dropped)
            upload_ms, total_ms = [], []
            for k in range(9, 14):
                flat, t0, t1, t2 = new_scene(k)
                ctx.upload_scene(flat, sky=None)
                t3 = time.perf_counter()
                ctx.synchronize()
                t4 = time.perf_counter()
                ctx.render(1)
                if k >= 10:
                    upload_ms.append((t3 - t2) * 1e3), total_ms.append((t4 - t0) * 1e3)
            out["rebuild_20k"]["synchronous_route"] = {"upload_static_and_dynamic_ms": round(float(np.median(upload_ms)), 3), "ms_until_adopted": round(float(np.median(total_ms)), 3),
                                                       "what": "pt_upload_static + pt_upload_dynamic: what rounds 3-4 timed (no frame rendered meanwhile)"}
        finally:
            ctx.close()
    except Exception as e:
        out["rebuild_20k"] = {"error": str(e)[:200]}
    out["what"] = (f"medians over {ticks} ticks: Mesh::refit (positions + smooth normals; host library, through the Python binding) + pt_refit_vertices (the mesh's vertices into "
                   "pinned memory; boxes and triangle records re-made by the device on the copy stream) + flattenDynamic (lights, top level) + pt_upload_dynamic_async "
                   "(host time each), then pt_frame_tick + synchronise (ms_until_adopted = the whole tick); `host_refitted_nodes_route` = the same tick the reference's way")
    out["reference_ms_per_frame"] = "4.12 - 6.25 (refit from a binned / SBVH tree + upload, 36.5 k-triangle helicopter, RX 480; lab report Table 2) -- 26.7 - 381.6 with a rebuilt tree"
    return out


def two_level_general_times(D, scenes, W, Hh, level, device, in_flight, only=None, ways=("entered", "entered_parked", "default_flags", "copied")):
    """traceRay's GENERAL case (scene.cl:116-139 enters any 4 x 4 inverse transform, any number of instances) timed, not only tested.
    432 instances of the two 82 k-triangle meshes (35 M instanced triangles; ~2.5 GB as world-space copies: over the library's 2 GB copy budget)
    under config 4's camera, each turned about the vertical axis and scaled by three different factors; and 208 translated + uniformly scaled
    ones (more than the 95 the fold table holds).  Each scene four ways: every instance entered by the round-6 route (leaf-kind entry steps,
    nothing parked), by the parked route of rounds 2-5 (PT_FLAG_PARKED_INSTANCES), the library's default (copies while the budget lasts), and
    every instance copied to world space with the budget raised (PTAMD_BAKE_BUDGET_GB: what instancing exists to avoid -- the per-ray rate it
    reaches is the yardstick)."""
    res = {}
    for name, kw, n_inst in (("general_432", dict(nx=24, nz=18, transform="general"), 432), ("uniform_208", dict(nx=16, nz=13, transform="uniform"), 208)):
        if only and name not in only:
            continue
        crowd = scenes.instanced_crowd(W, Hh, level=level, **kw)
        what = (f"{n_inst} instances of the two {crowd.flat.instanced_triangles // n_inst}-triangle meshes = {crowd.flat.instanced_triangles} instanced triangles, config 4's "
                "camera / materials / sky; " + ("each turned about the vertical axis by a random angle and scaled by three different factors"
                                                if name.startswith("general") else "translated + uniformly scaled (configs 4 / 5's kind of transform)"))
        r = {"what": what}
        if "entered" in ways:
            r["entered"] = measure_scene(D, crowd, W, Hh, device, in_flight, flags=D.FLAG_NO_BAKED_INSTANCES, steps=2,
                                         what="every instance entered (PT_FLAG_NO_BAKED_INSTANCES): the general route of round 6, pt_trace.h LEVELS 2")
        if "entered_parked" in ways:
            r["entered_parked"] = measure_scene(D, crowd, W, Hh, device, in_flight, flags=D.FLAG_NO_BAKED_INSTANCES | D.FLAG_PARKED_INSTANCES, steps=2,
                                                what="every instance entered by the parked route of rounds 2-5 (PT_FLAG_PARKED_INSTANCES)")
        if "default_flags" in ways:
            r["default_flags"] = measure_scene(D, crowd, W, Hh, device, in_flight, flags=0, steps=2,
                                               what="the library's default: instances copied to world space while the 2 GB budget lasts, the rest entered")
        if "copied" in ways:
            os.environ["PTAMD_BAKE_BUDGET_GB"] = "16"
            try:
                r["copied"] = measure_scene(D, crowd, W, Hh, device, in_flight, flags=0, steps=2,
                                            what="every instance copied to world space (copy budget raised to 16 GB through PTAMD_BAKE_BUDGET_GB): the yardstick")
            finally:
                del os.environ["PTAMD_BAKE_BUDGET_GB"]
            for key in ("entered", "entered_parked", "default_flags"):
                if key in r:
                    r[f"{key}_over_copied"] = round(r[key]["mrays_per_s"] / r["copied"]["mrays_per_s"], 4)
                    r[f"{key}_over_copied_per_ray_kernels"] = round(r[key]["per_ray_kernels_mrays_per_s"] / r["copied"]["per_ray_kernels_mrays_per_s"], 4)
        res[name] = r
    return res


def secondary_measurements(D, H, L, scenes, bundle, args, device, in_flight):
    """The N = 1 line's secondary objects.  Every entry is guarded: a failure is reported in place and never takes the headline down."""
    W, Hh = args.width, args.height
    out = {}

    def guarded(key, fn):
        try:
            out[key] = fn()
        except Exception as e:
            out[key] = {"error": str(e)[:300]}

    guarded("two_level", lambda: {
        "every_instance_entered": measure_scene(D, bundle, W, Hh, device, in_flight, flags=D.FLAG_NO_BAKED_INSTANCES,
                                                what="PT_FLAG_NO_BAKED_INSTANCES: all 14 instances (12 meshes, ground quad, light quad) are entered "
                                                     "(scene.cl:116-139), nothing is copied to world space"),
        "single_leaf_meshes_copied": measure_scene(D, bundle, W, Hh, device, in_flight, flags=D.FLAG_TWO_LEVEL_ONLY,
                                                   what="PT_FLAG_TWO_LEVEL_ONLY: the 12 mesh instances are entered; the two quads (single-leaf meshes) "
                                                        "hang off the top level as world-space leaves"),
        "instances_copied_to_world_space": measure_scene(D, bundle, W, Hh, device, in_flight, flags=0, what="the headline's configuration, measured the same way")})
    def in_flight_curve():
        # the rate against the footprint.  Samples in flight per pixel x what stays resident for them, with the queues sized by the fractions
        # of the headline (EXT_QUEUE_FRACTION / SHADOW_QUEUE_FRACTION) and, at the headline's own count, as large as the batch (rounds 1-5: 164 B per entry).
        pts = {}
        fe, fs = args.ext_queue_fraction, args.shadow_queue_fraction
        for n in (32, 128, 256, 512):
            r = measure_scene(D, bundle, W, Hh, device, n, steps=3, queue_fractions=(fe, fs))
            pts[str(n)] = {"mrays_per_s": r["mrays_per_s"], "resident_gb": round(resident_bytes(n, W * Hh, bytes_per_entry(fe, fs)) / 1e9, 1), "batch_samples": r["batch_samples"],
                           "first_pass_ratios": r["first_pass_ratios"]}
        r = measure_scene(D, bundle, W, Hh, device, in_flight, steps=3)
        pts[f"{in_flight}_full_queues"] = {"mrays_per_s": r["mrays_per_s"], "resident_gb": round(resident_bytes(in_flight, W * Hh) / 1e9, 1), "batch_samples": r["batch_samples"]}
        return {"what": f"config 4 as in the headline, samples in flight per pixel -> Mrays/s and resident GB (queues + accumulator planes); queue fractions ext {fe} / shadow {fs} "
                        f"= {bytes_per_entry(fe, fs):.0f} B per entry; the last point: queues as large as the batch (164 B per entry, rounds 1-5)", "points": pts}
    guarded("in_flight_curve", in_flight_curve)

    guarded("two_level_general", lambda: two_level_general_times(D, scenes, W, Hh, args.level, device, in_flight))

    def dynamic():
        d = dynamic_update_times(D, H, L, scenes, bundle, W, Hh, device)
        d["refit"] = dynamic_refit_times(D, H, L, scenes, W, Hh, device)
        return d
    guarded("dynamic", dynamic)

    def configs():
        c = {}
        room = scenes.blob_room(W, Hh, level=args.level, builder=H.BVH_BINNED_SAH)
        c["config2"] = measure_scene(D, room, W, Hh, device, in_flight, what="configs[1]: 81 920-triangle mesh (binned SAH), diffuse, five-wall room + area light, 1080p")
        glass = scenes.blob_room(W, Hh, level=args.level, builder=H.BVH_SPATIAL_SPLIT, material=L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0))
        c["config3"] = measure_scene(D, glass, W, Hh, device, in_flight, what="configs[2]: the same mesh as rough glass (SBVH), NEE towards the area light, 1080p")
        W4, H4 = 2 * W, 2 * Hh
        big = scenes.instanced_grid(W4, H4, nx=4, nz=3, level=args.level, builder=H.BVH_SPATIAL_SPLIT, thin_lens=True)
        n4 = plan_in_flight(args.in_flight, 1, W4 * H4, args.max_entries)
        c["config5_one_gpu"] = measure_scene(D, big, W4, H4, device, n4, steps=2, what=f"configs[4] on ONE GPU: the headline scene at {W4}x{H4}, thin lens f/2")
        return c
    guarded("configs", configs)

    def grid_5x3():
        # SURVEY 8(d) specifies config 4 as 5 x 3 = 15 instances (1 041 765 triangles with the Stanford bunny, which does not travel to the GPU box); the headline
        # has timed 4 x 3 instances of the two 82 k-triangle substitute meshes since round 1 (985 012: 5 % under) and stays that way so that the series stays
        # comparable -- this is the same measurement at 5 x 3 (1 231 264 instanced triangles: over, not under, 1 M)
        big = scenes.instanced_grid(W, Hh, nx=5, nz=3, level=args.level, builder=H.BVH_SPATIAL_SPLIT)
        r = measure_scene(D, big, W, Hh, device, in_flight, steps=3, what="config 4 to the letter of SURVEY 8(d): 5 x 3 = 15 instances, the headline's measurement otherwise")
        r["instanced_triangles"] = int(big.flat.instanced_triangles)
        return r
    guarded("grid_5x3", grid_5x3)

    def material_order():
        m = {"what": "k_shade on scenes with five material types on one mesh (diffuse, PBR metal, PBR dielectric, rough glass, basic glass) in "
                     "the room of configs 2/3: tiles shaded in queue order (the default) against material order (PT_FLAG_MATERIAL_BINS)"}
        for pattern in ("patches", "confetti"):
            sc = scenes.mixed_material_room(W, Hh, level=args.level, pattern=pattern)
            for name, fl in (("queue_order", 0), ("material_order", D.FLAG_MATERIAL_BINS)):
                r = measure_scene(D, sc, W, Hh, device, in_flight, flags=fl, steps=2)
                m[f"{pattern}_{name}"] = {k: r[k] for k in ("mrays_per_s", "shade_ns_per_entry", "kernel_ms_per_step")}
        return m
    guarded("material_order", material_order)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3, help="batches per step (3: the driver's 20 steps time ~11 s of rendering)")
    ap.add_argument("--in-flight", type=int, default=IN_FLIGHT, help="samples in flight per pixel and per rank-share (batch = in_flight*N samples)")
    ap.add_argument("--max-entries", type=int, default=MAX_ENTRIES, help="path segments resident per rank (memory: ~190 B each)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--level", type=int, default=6, help="icosphere subdivision of the instanced mesh (6 = 81 920 tris)")
    ap.add_argument("--cpu-seconds", type=float, default=9.0, help="wall time of the CPU baseline's timed part (a warm-up pass comes on top)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-frame", action="store_true", help="skip the 1-spp 720p frame-time figure")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--frame-scene", default="", help="--mode frame: only the scenes whose name contains this (profiling one scene's frames)")
    ap.add_argument("--no-frame-after", dest="frame_after", action="store_false", help="skip the second frame-time figure (after the headline's allocations)")
    ap.add_argument("--mode", default="throughput", choices=["throughput", "frame"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: in_flight x N samples in flight per rank on 1/N of the pixels (per-rank load fixed); strong: in_flight x rounds "
                         "samples per pixel of the whole image per step whatever N is (job fixed)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the two_level / dynamic / configs objects of the N = 1 line")
    ap.add_argument("--thin-lens", action="store_true", help="configs[4]'s camera: thin lens f/2 focused on the grid centre (with --width 3840 --height 2160: config 5)")
    ap.add_argument("--grid", default="4x3", help="instances of the two meshes, columns x rows (4x3 = 985 012 instanced triangles: the headline since round 1; "
                                                  "5x3 = SURVEY 8(d)'s 15 instances, 1 231 264 with these meshes -- also reported as the `grid_5x3` object of the default line)")
    ap.add_argument("--builder", default="spatial", choices=["spatial", "binned", "fast"], help="BVH builder of the instanced meshes (diagnostic: how much the tree's quality is worth)")
    ap.add_argument("--ext-queue-fraction", type=float, default=EXT_QUEUE_FRACTION, help="pt_config.ext_queue_fraction of the headline's context (0 or 1: queues as large as the batch)")
    ap.add_argument("--shadow-queue-fraction", type=float, default=SHADOW_QUEUE_FRACTION, help="pt_config.shadow_queue_fraction of the headline's context")
    ap.add_argument("--flags", type=int, default=0, help="pt_config.flags of the render context (2 = PT_FLAG_NO_BAKED_INSTANCES: two-level traversal)")
    ap.add_argument("--dump-accum", default=None, help="rank 0 saves the (reduced) HDR accumulator as .npy (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start the ranks ourselves, as fresh child processes (this process has not touched the GPU -- torch
        # is not even imported yet -- and never will: it only waits for the children and passes rank 0's line through).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or "
                         f"without a launcher (bench.py then starts the ranks itself)")

    original_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    host_affinity = pin_to_one_l3_domain(local_rank)  # (before torch, HIP and the host library start their threads: they inherit it)
    import torch
    import torch.distributed as dist
    from ptamd import device as D, host as H, layout as L, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    if args.mode == "frame":
        if world != 1:
            raise SystemExit("--mode frame is a single-GPU measurement")
        fr = frame_times(D, H, scenes, L, local_rank, args.frames, only=args.frame_scene)
        best = min(v["ms_per_frame"] for v in fr["scenes"].values())
        print(json.dumps({"metric": "ms per 1-spp 1280x720 frame (RayTracer::rayTrace + accumulate)", "value": best, "unit": "ms", "n_gpus": 1,
                          "steps": args.frames, "warmup": 20, "ms_per_step": best, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic", "config": {"workload": fr["what"]}, "frame": fr}), flush=True)
        return

    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    W, Hh = args.width, args.height
    # The interactive frame (a secondary object: RayTracer::rayTrace, 1 spp of a 1280 x 720 frame) is measured FIRST, on the device as the process found
    # it, like `--mode frame` does.  After the headline has allocated and freed its ~190 GB of queues the same frames take ~20 % longer (0.93 instead of
    # 0.76 ms; same kernels, same launch counts: the small buffers of a frame context then come out of freed, fragmented device memory) -- a property of
    # the allocator's state, not of the path; `frame_after_headline` (--frame-after too) reports that figure as well.
    frame = frame_after = None
    if rank == 0 and world == 1 and not args.no_frame:
        try:
            frame = frame_times(D, H, scenes, L, local_rank, args.frames)
        except Exception as e:  # a secondary figure must not take the headline down
            frame = {"error": str(e)[:300]}
    gx, gz = (int(v) for v in args.grid.lower().split("x"))
    bundle = scenes.instanced_grid(W, Hh, nx=gx, nz=gz, level=args.level, thin_lens=args.thin_lens,
                                   builder={"spatial": H.BVH_SPATIAL_SPLIT, "binned": H.BVH_BINNED_SAH, "fast": H.BVH_BINNED_FAST}[args.builder])
    flat = bundle.flat
    rects = tile_rects(W, Hh, rank, world) if world > 1 else []
    owned = sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in rects) if rects else W * Hh
    in_flight = planned = plan_in_flight(args.in_flight, world, owned, args.max_entries)
    if args.scaling == "strong":  # the job is `in_flight x rounds` samples per pixel: a rank cannot keep more of them in flight than that
        in_flight = planned = min(in_flight, args.in_flight * args.rounds)
    # ... and what the card has free right now decides (512 in flight want ~200 GB; a second tenant, a context not yet freed or a
    # smaller part must cost a percent or two, not the line)
    free_b, total_b = torch.cuda.mem_get_info(local_rank)
    if args.share_gpu:
        free_b //= world
    fe, fs = args.ext_queue_fraction, args.shadow_queue_fraction
    per_entry = bytes_per_entry(fe, fs, args.thin_lens)
    in_flight = fit_in_flight(in_flight, owned, free_b, per_entry=per_entry)

    def agree(value):  # every rank must use the same batch: the smallest share decides
        if world == 1:
            return value
        t = torch.tensor([value], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    in_flight = agree(in_flight)
    # torch owns the stream and the accumulator: everything below -- render kernels, the reduce, the statistics -- is enqueued
    # on ONE explicit stream.  (torch's default stream is the legacy null stream, whose handle is 0; pt_set_stream(NULL)
    # would mean "the library's own non-blocking stream", which the collective would NOT be ordered after.)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        accum = torch.zeros(W * Hh, 4, device="cuda", dtype=torch.float32)
        fallbacks = []
        while True:
            # the context, and ONE batch through it: the queues are set up at the first render, which is where an oversized configuration is refused
            ctx, err = None, ""
            try:
                ctx = D.Context(W, Hh, seed=1, device=local_rank, samples_in_flight=in_flight, flags=args.flags, ext_queue_fraction=fe, shadow_queue_fraction=fs)
                ctx.set_stream(stream.cuda_stream)
                ctx.upload_scene(flat, sky=bundle.sky, material_textures=bundle.material_textures)
                ctx.set_camera(bundle.camera)
                if world > 1:
                    ctx.set_tiles(rects)
                ctx.set_accum_buffer(accum.data_ptr())
                ctx.render(in_flight, sync=True)
            except D.PtError as e:
                err = str(e)
                if ctx is not None:
                    ctx.close()
                ctx = None
            if agree(1 if ctx is not None else 0):
                break
            if ctx is not None:  # another rank was refused: everybody starts over with the smaller batch
                ctx.close()
            if in_flight <= 1:
                raise SystemExit(f"bench.py: no batch size fits this device: {err}")
            fallbacks.append({"samples_in_flight": in_flight, "refused": err[-240:]})
            in_flight = agree(shrink_in_flight(in_flight))
        spp_step = in_flight * args.rounds if args.scaling == "weak" else args.in_flight * args.rounds

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            ctx.render(spp_step, sync=False)
        barrier()
        free_after, _ = torch.cuda.mem_get_info(local_rank)
        ctx.clear()
        ctx.reset_stats()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.render(spp_step, sync=False)
        reduce_accumulator(dist, accum, args.backend, world)  # inside the timed region, ordered after the kernels (same stream)
        barrier()
        elapsed = time.perf_counter() - t0

        st = ctx.stats()
        (ext, shadow, gen, hits, deposits), elapsed = aggregate(
            dist, torch, [st["rays_extension"], st["rays_shadow"], st["rays_generated"], st["shade_hits"], st["deposits"]], elapsed, args.backend, world)
        total_rays = ext + shadow
        total_spp = ctx.samples_per_pixel  # per owned pixel: N x more samples per step on the image at N ranks
        image_mean = float(accum[:, :3].mean().item()) / max(total_spp, 1) if rank == 0 else 0.0
        if rank == 0 and args.dump_accum:
            np.save(args.dump_accum, accum.cpu().numpy())

        roofline = None
        if not args.no_roofline:
            roofline = measure_roofline(ctx, torch, args, W, Hh, in_flight, spp_step, world)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        pinned = os.sched_getaffinity(0) if original_affinity is not None else None
        if original_affinity is not None:
            # the oracle's threads start here and inherit this mask: every CPU of the memory node the pinned process allocated the scene on (threads on the other
            # socket read it remotely: 14-16 against 27-28 Mrays/s, alternating on one box), or the mask the process came with
            os.sched_setaffinity(0, set(host_affinity.get("node_cpus") or original_affinity))
        try:
            cpu = cpu_baseline(bundle, args.cpu_seconds, W, Hh)
        finally:
            if pinned is not None:
                os.sched_setaffinity(0, pinned)
    ctx.close()
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = secondary_measurements(D, H, L, scenes, bundle, args, local_rank, in_flight)
    if rank == 0 and world == 1 and not args.no_frame and args.frame_after:
        try:
            fa = frame_times(D, H, scenes, L, local_rank, max(args.frames // 2, 20))
            frame_after = {"scenes": {k: v["ms_per_frame"] for k, v in fa["scenes"].items()},
                           "what": "the same frames measured again AFTER the headline allocated and freed its queues (see the comment in bench.py)"}
        except Exception as e:
            frame_after = {"error": str(e)[:300]}

    if rank == 0:
        out = {
            "metric": "Mrays/s at 1080p, 1M-tri SBVH scene; 1/2/4/8-GPU scaling + %HBM roofline",
            "value": round(total_rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[{4 if args.thin_lens else 3}]: instanced ~1M-triangle grid ({gx}x{gz} instances of two {flat.instanced_triangles // (gx * gz)}-triangle "
                            f"SBVH meshes = {flat.instanced_triangles} instanced triangles, PBR metal/dielectric, procedural HDR sky + "
                            f"emissive quad), scene handed over as the reference's two-level BVH, {W}x{Hh}, {'thin lens f/2, ' if args.thin_lens else ''}4 bounces, NEE + Russian roulette, counter PRNG; "
                            + ("every instance entered at traversal (PT_FLAG_NO_BAKED_INSTANCES)" if args.flags & 2 else
                               "mesh instances entered at traversal, single-leaf meshes copied (PT_FLAG_TWO_LEVEL_ONLY)" if args.flags & 4 else
                               "instances copied to world space at upload (the library's default while they fit a 2 GB budget; the `two_level` "
                               "object times the same scene with the instances entered instead)"),
                "scene_flags": args.flags, "csrc_sha256": csrc_sha256(), "host_affinity": {k: v for k, v in host_affinity.items() if k != "node_cpus"},
                "width": W, "height": Hh, "level": args.level, "spp_per_step": spp_step, "samples_in_flight": in_flight, "batches_per_step": args.rounds,
                "samples_in_flight_planned": planned, "samples_in_flight_fallbacks": fallbacks,
                "resident_gb": round(resident_bytes(in_flight, owned, per_entry) / 1e9, 1),  # queues + accumulator planes of this rank (bytes_per_entry, 16 B per plane and pixel)
                "resident_bytes_per_entry": round(per_entry, 1), "queue_fractions": {"ext": fe, "shadow": fs},
                "batch_samples": st["batch_samples"], "first_pass_ratios": {"ext": round(st["first_pass_ext_ratio"], 4), "shadow": round(st["first_pass_shadow_ratio"], 4)},
                "probe_batches": st["probe_batches"],
                "device_memory_gb": {"total": round(total_b / 1e9, 1), "free_before": round(free_b / 1e9, 1), "free_while_rendering": round(free_after / 1e9, 1)},
                "tiles": "whole frame" if world == 1 else f"{TILE}x{TILE} tiles interleaved over ranks",
                "pixels_per_rank": owned, "paths_per_step_per_rank": owned * spp_step,
                "collective": "none" if world == 1 else f"1 x reduce(SUM) of the HDR accumulator ({args.backend}) per job, inside the timed region",
            },
            "rays": {"extension": int(ext), "shadow": int(shadow), "primary": int(gen), "shade_hits": int(hits), "deposits": int(deposits)},
            "timed_region_s": round(elapsed, 3),
            "image_mean_radiance": round(image_mean, 5),
            # the headline is a MIX: camera rays (coherent, bundles of 4 x 64) / bounce rays / shadow rays, with the rate inside the kernel that traces each class
            "rays_by_class": roofline["rays_by_class"] if roofline else None,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "frame": frame,
            "frame_after_headline": frame_after,
        }
        if secondary:
            out.update(secondary)
        if cpu:
            out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def csrc_sha256():
    """Fingerprint of the device library's sources + compiler flags (ptamd.build.csrc_fingerprint): goes into config.csrc_sha256 of every line, from where
    tools/traffic_json.py copies it into the traffic*.json made from the same session's PMC passes."""
    from ptamd import build as B
    return B.csrc_fingerprint()


def apply_traffic(k, name, tj, stale):
    """The counter half of a kernel's roofline entry from a committed traffic*.json: HBM bytes per unit always (flagged by the caller when stale), the
    issue-port figures (valu_*, wave_cycles_*) only when the counters were taken on THIS build -- they describe the instruction stream, and a changed
    kernel invalidates them."""
    if tj and name in tj.get("kernels", {}):
        k["traffic_bytes_per_unit"] = tj["kernels"][name]["bytes_per_unit"]["total"]
        k["traffic"] = round(k["traffic_bytes_per_unit"] * k["units_per_launch"])
    if tj and not stale and name in tj.get("issue", {}):
        k.update(tj["issue"][name])  # valu_issue_frac, active lanes (PMC)
    return k


def traffic_is_stale(tj, fingerprint):
    """True when the counters in `tj` were not taken on the build with this fingerprint (or do not say which build they were taken on)."""
    return bool(tj) and tj.get("csrc_sha256") != fingerprint


def newest_traffic_json(W, Hh, level, in_flight, world, flags=0):
    """HBM traffic per unit from the committed PMC passes (profiles/roundN/traffic*.json) of exactly this configuration (scene flags
    included: instances copied / entered)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "traffic*.json")), reverse=True):
        try:
            tj = json.load(open(path))
            c = tj["config"]
            if (c["width"], c["height"], c["level"], c["samples_in_flight"], c["n_gpus"], tj.get("scene_flags", 0) & 6) == (W, Hh, level, in_flight, world, flags & 6):
                tj["_path"] = os.path.relpath(path, ROOT)
                return tj
        except (OSError, KeyError, ValueError):
            continue
    return None


def measure_roofline(ctx, torch, args, W, Hh, in_flight, spp_step, world):
    """Profiled pass (HIP events around every launch, on the render stream); not part of the timed steps.  Per kernel: units
    processed, launches and device time of the pass; algorithmic bytes per unit are SURVEY.md 8(d)'s figures.  HBM traffic per
    unit and the issue-port occupancy come from the committed PMC passes of this exact configuration."""
    ctx.reset_stats()
    ctx.profile_kernels(True)
    ctx.render(spp_step, sync=True)
    ps = ctx.stats()
    ctx.profile_kernels(False)
    batches = spp_step // in_flight
    packets = ps["packet_launches"] > 0  # primary rays went through k_trace_packet
    tj = newest_traffic_json(W, Hh, args.level, in_flight, world, args.flags)
    stale = traffic_is_stale(tj, csrc_sha256())
    kernels = {}

    def add(name, what, ms, launches, units, bytes_per_unit, extra_bytes=0.0):
        if launches <= 0 or ms <= 0 or units <= 0:
            return
        gbs = (bytes_per_unit * units + extra_bytes) / (ms * 1e-3) / 1e9
        k = {"computes": what, "ms_per_step": round(ms, 3), "launches": launches, "avg_launch_ms": round(ms / launches, 4),
             "units_per_launch": int(units / launches), "algorithmic_bytes_per_unit": round(bytes_per_unit + extra_bytes / units, 2),
             "achieved": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5), "munits_per_s": round(units / ms / 1e3, 1),
             "traffic": None}
        kernels[name] = apply_traffic(k, name, tj, stale)

    primary = ps["rays_generated"] if packets else 0
    fused = packets and ps["ms_gen"] < 0.05 * batches  # the packet kernel generated (and queued) the primary rays itself: no k_gen launch
    dep_shadow, dep_shade = ps["deposits_shadow"], ps["deposits"] - ps["deposits_shadow"]
    add("k_trace<true>", "any-hit traversal of the shadow rays + deposit of the unoccluded ones", ps["ms_shadow"], 4 * batches, ps["rays_shadow"],
        BYTES_PER_SHADOW_RAY, BYTES_PER_DEPOSIT * dep_shadow)
    add("k_trace<false>", "closest-hit traversal, one ray per lane (bounce rays%s)" % ("" if packets else " and primary rays"),
        ps["ms_intersect"] - ps["ms_packet"], (3 if packets else 4) * batches, ps["rays_extension"] - primary, BYTES_PER_EXT_RAY)
    bundles = ps.get("bundle_launches", 0) > 0  # k_trace_multi: one tree walk per 4 x 64 camera rays
    add("k_trace_multi<4>" if bundles else "k_trace_packet<false>",
        ("closest-hit traversal of the primary rays, one bundle of 4 x 64 per wave (one tree walk, four rays per lane)" if bundles
         else "closest-hit traversal of the primary rays, one packet of 64 per wave")
        + ((" (generates the camera rays itself and queues direction + pixel for k_shade, which knows the eye: 16 B + the 20 B hit record written per ray, nothing read)"
            if bundles else " (generates the camera rays itself and queues them for k_shade: 32 B written per ray instead of 28 B read)") if fused else ""),
        ps["ms_packet"], batches, primary, ((16 if bundles else BYTES_PER_GEN_RAY) + 20) if fused else BYTES_PER_EXT_RAY)
    add("k_shade<false>", "shade + NEE + continuation + compaction (+ deposits of emissive hits and sky misses)", ps["ms_shade"], 4 * batches,
        ps["shade_hits"], BYTES_PER_SHADED_HIT, BYTES_PER_DEPOSIT * dep_shade)
    if not fused:
        add("k_gen", "primary rays", ps["ms_gen"], batches, ps["rays_generated"], BYTES_PER_GEN_RAY)
    dominant = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
    dk = kernels[dominant]
    achieved = dk["achieved"]
    # achievable HBM rate on this box: a float4 grid-stride copy of 2 x 2 GiB (pt_debug_copy_bandwidth, csrc/pt_bake.h: 5.5-5.6 TB/s on this pool, the best of
    # sixty shapes; rounds 1-5 timed torch's Tensor.copy_: 4.7-4.9.  MI355X_MICROARCH.md's 6.29 TB/s was not reproduced: quote `frac`, which is of the 8 TB/s peak)
    try:
        copy_gbs = ctx.copy_bandwidth(2 << 30, 5)
    except Exception:
        copy_gbs = float("nan")
    # SURVEY 8(d) formula over the whole step
    path_bytes = (48.0 * ps["rays_generated"] + 48.0 * ps["rays_extension"] + 160.0 * ps["shade_hits"] + 44.0 * ps["rays_shadow"]
                  + 24.0 * ps["deposits"])
    path_ms = ps["ms_gen"] + ps["ms_intersect"] + ps["ms_shade"] + ps["ms_shadow"]
    return {"bound": "hbm", "kernel": f"{dominant} ({dk['computes']}; largest share of device time)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": dk["traffic"],
            # the counters (traffic, valu_*, wave_cycles_*) come from the committed PMC passes named in traffic_note; true: they were taken on ANOTHER build of
            # csrc/ than the one timed here (config.csrc_sha256) -- the HBM bytes are then quoted as an indication, the issue-port figures dropped
            "traffic_stale": stale if tj else None, "traffic_build": (tj or {}).get("csrc_sha256"),
            "peak_measured_copy": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 5),
            "whole_path_achieved": round(path_bytes / (path_ms * 1e-3) / 1e9, 2),
            "traffic_note": (f"HBM bytes per launch = bytes per unit from the FETCH_SIZE / WRITE_SIZE PMC passes of this exact configuration "
                             f"({tj['_path']} states the corrections) x units per launch") if dk["traffic"] else None,
            "algorithmic_bytes_per_unit": dk["algorithmic_bytes_per_unit"], "units_per_launch": dk["units_per_launch"],
            "avg_launch_ms": dk["avg_launch_ms"], "launches": dk["launches"],
            "valu_issue_frac": dk.get("valu_issue_frac"), "issue_model": (tj or {}).get("issue_model"),
            # counters, next to the modelled fraction: rocprof's VALUBusy (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles) and what the resident
            # waves were doing (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES), from the committed PMC passes
            "valu_busy": dk.get("valu_busy"), "valu_active_lanes": dk.get("valu_active_lanes"),
            "wave_cycles_issuing": dk.get("wave_cycles_issuing"), "wave_cycles_waiting": dk.get("wave_cycles_waiting"),
            "mrays_per_s_in_kernel": dk["munits_per_s"],  # of THIS kernel (rounds 1-5 quoted the closest-hit family here: see rays_by_class of the line)
            "rays_by_class": rays_by_class(ps),
            "kernels": kernels,
            "family_ms": {"gen": round(ps["ms_gen"], 3), "intersect": round(ps["ms_intersect"], 3),
                          "shade": round(ps["ms_shade"], 3), "shadow": round(ps["ms_shadow"], 3)}}


if __name__ == "__main__":
    main()
