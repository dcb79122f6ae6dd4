#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5o; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_dynamic.py tests/test_gpu_mis.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
frame() { python3 -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k[:8]: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"; }
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --mode frame > $out/frame_$tag.json 2> $out/frame_$tag.err || { echo "frame bench $tag failed"; tail -3 $out/frame_$tag.err; exit 1; }; frame $out/frame_$tag.json "$tag"; }
run default X=1
run rounds3 PTAMD_TEAM_ROUNDS=3
run rounds4 PTAMD_TEAM_ROUNDS=4
run off PTAMD_TEAM_ROUNDS=0
run team7 PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_team7.so
run default_again X=1
PTAMD_TEAM_USE=3 tools/r5_frame_prof2.sh 2 > $out/trace.txt 2>&1; grep -E "PTAMD|k_trace|k_shade|k_gen|k_resolve|copyBuffer|k_merge|k_end" $out/trace.txt | cut -c1-130
