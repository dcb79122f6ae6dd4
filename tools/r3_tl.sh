#!/bin/bash
# GPU box: the bench scene with the instances copied / entered for every named library variant:
#   tools/r3_tl.sh tag "flags.." variant...      flags: 0 copied, 2 every instance entered, 4 only the meshes entered
#   (base = the in-tree library; others from tools/mkvariants.sh).  TESTS=1 runs the traversal tests on the in-tree library first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; flagsets=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$TESTS" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_intersect.py -m gpu -q -k "two_level or unbaked or thousand or beam" > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
  [ $rc -ne 0 ] && echo "TESTS FAILED" && exit $rc
fi
for v in "$@"; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  for fl in $flagsets; do
  PTAMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --steps 3 --warmup 1 --rounds 1 --flags $fl > $out/${v}_$fl.json 2> $out/${v}_$fl.err || exit 1
  python3 - <<PY
import json
d=json.load(open("$out/${v}_$fl.json")); r=d["roofline"]
print("$v flags $fl", d["value"], r["family_ms"], {k:(v["ms_per_step"],v["munits_per_s"]) for k,v in r["kernels"].items()})
PY
  done
done
