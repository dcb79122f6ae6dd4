#!/bin/bash
# GPU box, round 5 session K: the team kernel (four lanes per ray) -- traversal tests through the hook, frames through it, frame times by rounds allowed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5k; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_intersect.py -m gpu -q -x -k "team" > $out/pytest_team.log 2>&1; rc=$?
tail -4 $out/pytest_team.log
[ $rc -ne 0 ] && { echo "team traversal tests failed: stopping"; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_dynamic.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -4 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
frame() { python3 -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k[:8]: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"; }
for r in 0 1 2 4 8 64 0 2; do
  PTAMD_TEAM_ROUNDS=$r timeout -k 10 300 python bench.py --mode frame > $out/frame_rounds$r.json 2> $out/frame_rounds$r.err || { echo "frame bench failed"; tail -3 $out/frame_rounds$r.err; exit 1; }
  frame $out/frame_rounds$r.json "team_rounds=$r"
done
