#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5f; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
q="--no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1"
line() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernels']; print('$2', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', {n.split('<')[0]+n[-6:]: v['ms_per_step'] for n, v in k.items()})"; }
run() { name=$1; shift; timeout -k 10 240 python bench.py $q "$@" > $out/bench_$name.json 2> $out/bench_$name.err || { echo "bench $name failed"; tail -3 $out/bench_$name.err; exit 1; }; line $out/bench_$name.json $name; }
run copied
run entered --flags 2
export PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_mtl4.so
PTAMD_LIB=$PTAMD_LIB timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py -m gpu -q -x -k "first_pass or two_level" > $out/pytest_mtl4.log 2>&1; tail -2 $out/pytest_mtl4.log
run entered_mtl4 --flags 2
run meshes_mtl4 --flags 4
unset PTAMD_LIB
run copied_again
