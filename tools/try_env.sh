#!/bin/bash
# GPU box: bench one library under several values of an environment variable: tools/try_env.sh <lib name|base> VAR v1 [v2...]
v=$1; var=$2; shift 2
lib=opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
[ "$v" = base ] && lib=opencl-path-tracer_amd/csrc/libptamd.so
for val in "$@"; do
  env PTAMD_LIB=$lib $var=$val timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $var=$val', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step  trace', r['mrays_per_s_in_kernel'], r['family_ms'])" || exit 1
done
