#!/bin/bash
# GPU box: thin-lens packets (config 5 on one GPU) and the headline, after a change to the packet kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_intersect.py tests/test_gpu_fullsize.py -m gpu -q -k "beam or packet or config5 or timed or thin or determinism" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --steps 3 --warmup 1 --rounds 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('config 4', d['value'], {k:v['ms_per_step'] for k,v in r['kernels'].items()})"
timeout -k 10 400 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "opencl-path-tracer_amd"); sys.path.insert(0, ".")
import bench
from ptamd import scenes, device as D, host as H
for W, Hh, infl in ((3840, 2160, 64), (1920, 1080, 256)):
    big = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT, thin_lens=True)
    for fl, name in ((0, "copied"), (D.FLAG_NO_BAKED_INSTANCES, "entered")):
        r = bench.measure_scene(D, big, W, Hh, 0, infl, flags=fl, steps=2)
        print(f"thin lens {W}x{Hh} {infl} in flight, instances {name}: {r['mrays_per_s']} Mrays/s", r["kernel_ms_per_step"])
PY
