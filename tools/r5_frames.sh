#!/bin/bash
# GPU box: the four interactive scenes' frame times (bench.py --mode frame), the in-tree library and named variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5frames; mkdir -p $out
run() { name=$1; shift; env "$@" timeout -k 10 240 python bench.py --mode frame --frames 200 > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', {k[:7]: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"; }
run base X=1
for v in "$@"; do run $v PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so; done
run base_again X=1
