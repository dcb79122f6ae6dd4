#!/bin/bash
# GPU box: a kernel variant is BENCHED only after the intersect tests passed with it (a wrong traversal must show up as a failed test
# with a short timeout, not as a fault in a long run): tools/r3_variant_safe.sh tag variant...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  PTAMD_LIB=$lib timeout -k 10 240 python -m pytest tests/test_gpu_intersect.py -m gpu -q -x -k "golden or random_rays or edge_cases or large_leaves or deep" > $out/pytest_$v.log 2>&1
  rc=$?; tail -2 $out/pytest_$v.log
  if [ $rc -ne 0 ]; then echo "variant $v: tests failed (rc $rc): not benched"; [ $rc -ge 124 ] && exit $rc; continue; fi
  PTAMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --steps 3 --warmup 1 --rounds 1 > $out/${v}.json 2> $out/${v}.err || exit 1
  python3 - <<PY
import json
d=json.load(open("$out/${v}.json")); r=d["roofline"]
print("$v", d["value"], {k:(v["ms_per_step"],v["munits_per_s"]) for k,v in r["kernels"].items()})
PY
done
