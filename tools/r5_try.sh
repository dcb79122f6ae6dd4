#!/bin/bash
# GPU box: headline the short way for the in-tree library and named variants, A / B / A
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5try; mkdir -p $out
q="--no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1 $PT_TRY_FLAGS"
line() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernels']; print('$2', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', {n.split('<')[0]+n[-6:]: v['ms_per_step'] for n, v in k.items()})"; }
run() { name=$1; shift; env "$@" timeout -k 10 240 python bench.py $q > $out/bench_$name.json 2> $out/bench_$name.err || { echo "bench $name failed"; tail -3 $out/bench_$name.err; exit 1; }; line $out/bench_$name.json $name; }
run base X=1
for v in "$@"; do run $v PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so; done
run base_again X=1
