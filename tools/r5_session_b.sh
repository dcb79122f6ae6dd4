#!/bin/bash
# GPU box, round 5 session B: (1) traversal + render tests with the folded instance route in, (2) two-level bench: entry nodes vs the parked route vs
# the copied scene, (3) the leaf-formation sweep, (4) occluder-leaf reuse of shadow rays.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py tests/test_gpu_dynamic.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -4 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
q="--no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1"
line() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', r['family_ms'])"; }
timeout -k 10 240 python bench.py $q > $out/bench_copied.json 2> $out/bench_copied.err || { echo "bench failed"; tail -3 $out/bench_copied.err; exit 1; }
line $out/bench_copied.json copied
timeout -k 10 240 python bench.py $q --flags 2 > $out/bench_entered_folded.json 2> $out/bench_entered_folded.err || { echo "bench --flags 2 failed"; tail -3 $out/bench_entered_folded.err; exit 1; }
line $out/bench_entered_folded.json entered_folded
timeout -k 10 240 python bench.py $q --flags 4098 > $out/bench_entered_parked.json 2> $out/bench_entered_parked.err || { echo "bench --flags 4098 failed"; tail -3 $out/bench_entered_parked.err; exit 1; }
line $out/bench_entered_parked.json entered_parked
timeout -k 10 240 python bench.py $q --flags 4 > $out/bench_meshes_folded.json 2> $out/bench_meshes_folded.err || { echo "bench --flags 4 failed"; exit 1; }
line $out/bench_meshes_folded.json meshes_entered_folded
tools/r5_leaf_sweep.sh r5b 8 4 6 "4,105,20,35,0" "8,105,20,35,0" "6,105,60,35,1" "8,105,0,25,1" 0 || exit 1
timeout -k 10 400 python tools/occluder_hist.py 256 > $out/occluder_hist.txt 2>&1; rc=$?
grep -v amdgpu.ids $out/occluder_hist.txt | tail -8
echo "[occluder_hist] rc=$rc"
