#!/bin/bash
# GPU box, round 5 session C: folded instances through per-instance root copies + LDS table (no entry step); leaves split below the reference's 3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5c; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py tests/test_gpu_dynamic.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -4 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
q="--no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1"
line() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', r['family_ms'])"; }
run() { name=$1; shift; timeout -k 10 240 python bench.py $q "$@" > $out/bench_$name.json 2> $out/bench_$name.err || { echo "bench $name failed"; tail -3 $out/bench_$name.err; exit 1; }; grep "collapse:" $out/bench_$name.err | sort | uniq -c | cut -c1-230; line $out/bench_$name.json $name; }
run copied
run entered_folded --flags 2
run entered_parked --flags 4098
run meshes_folded --flags 4
export PTAMD_COLLAPSE_REPORT=1
for m in 2 1; do
  export PTAMD_MAX_LEAF=$m
  if [ $m = 2 ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_dynamic.py -m gpu -q -x --deselect tests/test_gpu_intersect.py::test_stack_bound_of_a_root_that_shares_its_subtree_with_an_earlier_root > $out/pytest_maxleaf$m.log 2>&1; rc=$?
    tail -3 $out/pytest_maxleaf$m.log
    [ $rc -ne 0 ] && { echo "tests failed with PTAMD_MAX_LEAF=$m"; exit 1; }
  fi
  run maxleaf$m
done
unset PTAMD_MAX_LEAF
run copied_again
