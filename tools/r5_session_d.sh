#!/bin/bash
# GPU box, round 5 session D: the whole -m gpu suite on the new defaults (leaves of <= 2, folded instances, bundles entering simple instances by
# scaling their beam), then the headline the short way: copied / entered, and the inner-vs-leaf vote re-swept for the shorter leaf steps.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5d; mkdir -p $out
if [ "$1" != notests ]; then
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --durations=12 > $out/pytest.log 2>&1; rc=$?
tail -18 $out/pytest.log | cut -c1-200
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
fi
q="--no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1"
line() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernels']; print('$2', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', r['family_ms'], {n.split('<')[0]+n[-6:]: v['ms_per_step'] for n, v in k.items()})"; }
run() { name=$1; shift; timeout -k 10 240 python bench.py $q "$@" > $out/bench_$name.json 2> $out/bench_$name.err || { echo "bench $name failed"; tail -3 $out/bench_$name.err; exit 1; }; line $out/bench_$name.json $name; }
run copied
run entered_folded --flags 2
run meshes_folded --flags 4
for v in voteA voteB voteC; do
  export PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  run $v
done
unset PTAMD_LIB
run copied_again
