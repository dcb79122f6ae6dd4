#!/bin/bash
# GPU box: round-3 starting point -- baked vs two-level on the bench scene, upload times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3a; mkdir -p $out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --steps 3 --warmup 1 --rounds 1 > $out/baked.json 2> $out/baked.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --steps 3 --warmup 1 --rounds 1 --flags 2 > $out/two_level.json 2> $out/two_level.err || exit 1
timeout -k 10 300 python tools/upload_time.py > $out/upload_time.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for n in ("baked","two_level"):
    d=json.load(open(f"gpurun_out/r3a/{n}.json")); r=d["roofline"]
    print(n, d["value"], r["family_ms"], {k:(v["ms_per_step"],v["munits_per_s"]) for k,v in r["kernels"].items()})
PY
grep -v amdgpu.ids $out/upload_time.txt
