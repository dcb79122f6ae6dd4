#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 300 python tools/dbg_field.py 2>&1 | grep -v amdgpu.ids | tail -16
for v in base nostatic nooverlap; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  PTAMD_LIB=$lib timeout -k 10 300 python bench.py --mode frame --frames 200 > $out/frame_$v.json 2> $out/frame_$v.err || exit 1
  python3 -c "
import json; d=json.load(open('$out/frame_$v.json')); print('$v', {k[:7]: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})"
done
