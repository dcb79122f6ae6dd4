#!/bin/bash
# GPU box: render / dynamic / full-size / multirank tests with the in-tree library, then bench.py --mode frame for base and every named variant (twice), then the
# headline for each.  tools/r3_frame_ab.sh tag variant...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_dynamic.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_intersect.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log | cut -c1-300; exit 1; }
tail -1 $out/pytest.log
for rep in 1 2; do for v in base "$@"; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so; [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  PTAMD_LIB=$lib timeout -k 10 300 python bench.py --mode frame --frames 300 > $out/frame_$v.json 2> $out/frame_$v.err || exit 1
  python3 -c "
import json
d=json.load(open('$out/frame_$v.json'))
print('$v', d['value'], {k:v['ms_per_frame'] for k,v in d['frame']['scenes'].items()})"
done; done
for v in base "$@"; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so; [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  PTAMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --steps 3 --warmup 1 --rounds 1 > $out/$v.json 2> $out/$v.err || exit 1
  python3 -c "
import json
d=json.load(open('$out/$v.json')); r=d['roofline']
print('$v', d['value'], {k:v['ms_per_step'] for k,v in r['kernels'].items()})"
done
