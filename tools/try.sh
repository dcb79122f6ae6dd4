#!/bin/bash
# GPU box: bench every variant library given (built by tools/mkvariants.sh): tools/try.sh name [name...]
for v in "$@"; do
  lib=opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  [ "$v" = base ] && lib=opencl-path-tracer_amd/csrc/libptamd.so
  PTAMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step  trace', r['mrays_per_s_in_kernel'], r['family_ms'])" || exit 1
done
