#!/bin/bash
# GPU box: headline bench lines for scene flags 0 (instances copied to world space) and 2 (entered), no secondary objects.  tools/r3_bench_flags.sh tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
for f in 0 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --steps 3 --warmup 1 --rounds 1 --flags $f > $out/f$f.json 2> $out/f$f.err || exit 1
  python3 - <<PY
import json
d=json.load(open("$out/f$f.json")); r=d["roofline"]
print("flags $f", d["value"], {k:(v["ms_per_step"],v["munits_per_s"]) for k,v in r["kernels"].items()})
PY
done
