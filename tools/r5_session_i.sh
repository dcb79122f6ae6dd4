#!/bin/bash
# GPU box, round 5 session I: interactive frames -- SMALL instantiations with ONE unified fetch per iteration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_dynamic.py tests/test_gpu_mis.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
frame() { python3 -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k[:8]: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"; }
for cfg in "0 0" "21 0" "21 1" "21 2" "10 0" "0 0" "21 0"; do
  set -- $cfg
  PTAMD_SMALL_LAUNCHES=$1 PTAMD_SMALL_FROM_PASS=$2 timeout -k 10 300 python bench.py --mode frame > $out/frame_$1_$2.json 2> $out/frame_$1_$2.err || { echo "frame bench failed"; tail -3 $out/frame_$1_$2.err; exit 1; }
  frame $out/frame_$1_$2.json "small=$1 from_pass=$2"
done
