#!/bin/bash
# GPU box: the counters behind bench.py's `two_level_general` ratios -- vector instructions, vector loads, LDS instructions, kernel cycles and TA busy per
# dispatch of the per-ray kernels, for one crowd scene ENTERED (general route) and COPIED: tools/pmc_tl_general.sh <tag> [scene, default uniform_208]
tag=$1; scene=${2:-uniform_208}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for way in entered copied; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    out=gpurun_out/pmc_${tag}_${way}_$i
    rm -rf $out
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 tools/tl_general.py --only $scene --ways $way --in-flight 256 > $out.log 2>&1 || { echo "pass $way $i failed"; tail -3 $out.log; exit 1; }
    echo "== $scene $way"; python3 tools/pmc_sum.py $out | grep "k_trace<\|k_trace_multi"
    rm -rf $out
  done
done
