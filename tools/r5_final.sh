#!/bin/bash
# GPU box, round 5: the whole -m gpu suite, then the round's committed evidence (tools/final_profiles.sh), then the C++ deform loop at both mesh sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5final; mkdir -p $out
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest.log 2>&1; rc=$?
tail -14 $out/pytest.log | cut -c1-200
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
cp gpurun_out/parity_margins.json $out/parity_margins.json
timeout -k 10 120 ./examples/deform_loop 135 135 20 > $out/deform_loop_36k.txt 2>&1; cat $out/deform_loop_36k.txt
timeout -k 10 120 ./examples/deform_loop 203 202 20 > $out/deform_loop_82k.txt 2>&1; cat $out/deform_loop_82k.txt
timeout -k 10 1100 tools/final_profiles.sh > $out/final.log 2>&1; rc=$?
tail -4 $out/final.log | cut -c1-900
echo "[final] rc=$rc"
