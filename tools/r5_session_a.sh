#!/bin/bash
# GPU box, round 5 session A: the three unmeasured axes of the per-ray kernels: (b) SDWA byte-select issue rates, (a) leaf
# formation sweep, (c) occluder-leaf reuse of shadow rays.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5a; mkdir -p $out
if [ "$1" = micro ]; then
timeout -k 10 300 tools/micro/valu_issue 2048 > $out/valu_issue_sdwa.md 2> $out/valu_issue.err || { echo "micro failed"; tail -3 $out/valu_issue.err; }
grep -E "sdwa|ubyte|v_or_b32\`|v_fma_f32\`|align" $out/valu_issue_sdwa.md | cut -c1-220
fi
tools/r5_leaf_sweep.sh r5a 0 8 4 6 "4,105,20,35,0" "8,105,20,35,0" "6,105,60,35,1" "8,105,0,25,1" 0 || exit 1
timeout -k 10 400 python tools/occluder_hist.py 256 > $out/occluder_hist.txt 2>&1; rc=$?
grep -v amdgpu.ids $out/occluder_hist.txt | tail -8
echo "[occluder_hist] rc=$rc"
