#!/bin/bash
# GPU box: the super-packet kernel (pt_packet_multi.h): render tests that hold packets against the per-ray kernel and the oracle first
# (short timeout), then the headline.  tools/r3_multi.sh tag [variant...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for v in base "$@"; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  PTAMD_LIB=$lib timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py tests/test_gpu_fullsize.py -m gpu -q -x -k "first_pass or determinism or random_viewpoints or timed or full_size or production" > $out/pytest_$v.log 2>&1
  rc=$?; tail -3 $out/pytest_$v.log
  if [ $rc -ne 0 ]; then echo "variant $v: tests failed (rc $rc): not benched"; [ $rc -ge 124 ] && exit $rc; continue; fi
  PTAMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --steps 3 --warmup 1 --rounds 1 > $out/$v.json 2> $out/$v.err || exit 1
  python3 - <<PY
import json
d=json.load(open("$out/$v.json")); r=d["roofline"]
print("$v", d["value"], {k:(v["ms_per_step"],v["munits_per_s"]) for k,v in r["kernels"].items()})
PY
done
