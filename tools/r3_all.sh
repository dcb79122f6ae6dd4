#!/bin/bash
# GPU box: the whole GPU suite + a quick bench: tools/r3_all.sh tag [bench args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 1700 python -m pytest tests -m gpu -q --durations=10 > $out/pytest.log 2>&1; rc=$?; tail -8 $out/pytest.log
grep -n "^FAILED\|^ERROR" $out/pytest.log | head -20
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --rounds 1 --cpu-seconds 3 "$@" > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python3 - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["family_ms"])
for k in ("two_level","dynamic","configs","material_order","frame"):
    print(k, json.dumps(d.get(k))[:1500])
PY
