#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5m; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py -m gpu -q -x -k "team or interactive or golden" > $out/pytest_team.log 2>&1; rc=$?
tail -3 $out/pytest_team.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
frame() { python3 -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k[:8]: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"; }
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --mode frame > $out/frame_$tag.json 2> $out/frame_$tag.err || { echo "frame bench $tag failed"; tail -3 $out/frame_$tag.err; exit 1; }; frame $out/frame_$tag.json "$tag"; }
run off PTAMD_TEAM_ROUNDS=0
run use1 PTAMD_TEAM_USE=1
run use2 PTAMD_TEAM_USE=2
run use3 PTAMD_TEAM_USE=3
run use3_rounds3 PTAMD_TEAM_USE=3 PTAMD_TEAM_ROUNDS=3
V=$PWD/opencl-path-tracer_amd/csrc/variants
run team6_use3 PTAMD_LIB=$V/libptamd_team6.so PTAMD_TEAM_USE=3
run team7_use3 PTAMD_LIB=$V/libptamd_team7.so PTAMD_TEAM_USE=3
run team7_use3_r3 PTAMD_LIB=$V/libptamd_team7.so PTAMD_TEAM_USE=3 PTAMD_TEAM_ROUNDS=3
run off_again PTAMD_TEAM_ROUNDS=0
