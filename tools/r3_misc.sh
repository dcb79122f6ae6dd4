#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 300 python tools/packet_stats.py 2>&1 | grep -v amdgpu.ids | tee $out/packet_stats.txt
timeout -k 10 300 python -m pytest tests/test_gpu_render.py -m gpu -q -k material 2>&1 | tail -2
timeout -k 10 400 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/material_order.txt
import sys, os, json
sys.path.insert(0, "opencl-path-tracer_amd"); sys.path.insert(0, ".")
import bench
from ptamd import scenes, device as D, host as H, layout as L
W, Hh = 1920, 1080
for pattern in ("patches", "confetti"):
    sc = scenes.mixed_material_room(W, Hh, level=6, pattern=pattern)
    for name, fl in (("queue order", 0), ("material order", D.FLAG_MATERIAL_BINS)):
        r = bench.measure_scene(D, sc, W, Hh, 0, 256, flags=fl, steps=2)
        print(f"{pattern:9s} {name:15s} {r['mrays_per_s']:8.1f} Mrays/s  k_shade {r['kernel_ms_per_step']['shade']:6.2f} ms per 256-sample batch, {r['shade_ns_per_entry']*1e3:5.1f} ps per entry")
PY
