#!/bin/bash
# GPU box, round 5 session G: interactive frames -- the SMALL instantiations (both step kinds per iteration) against the voting kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5g; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_dynamic.py tests/test_gpu_mis.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
frame() { python3 -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"; }
for mode in 0 1 0 1; do
  PTAMD_SMALL_LAUNCHES=$mode timeout -k 10 300 python bench.py --mode frame > $out/frame_small$mode.json 2> $out/frame_small$mode.err || { echo "frame bench failed"; tail -3 $out/frame_small$mode.err; exit 1; }
  frame $out/frame_small$mode.json small=$mode
done
for v in "$@"; do
  PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so timeout -k 10 300 python bench.py --mode frame > $out/frame_$v.json 2> $out/frame_$v.err || { echo "frame bench $v failed"; exit 1; }
  frame $out/frame_$v.json $v
done
