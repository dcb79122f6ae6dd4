#!/bin/bash
# GPU box: VALU instruction / lane counters of the traversal kernels for scene flags: tools/r5_pmc_flags.sh tag "flags" ...   (0 copied, 2 every instance entered,
# 4098 entered on the parked route).  One rocprofv3 run per counter set (--pmc with --kernel-trace only), a one-batch step each.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out/$tag
for fl in "$@"; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
    name=$(echo $set | cut -d' ' -f1)
    out=gpurun_out/$tag/flags${fl}_$name
    rm -rf $out; mkdir -p $out
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --no-secondary --rounds 1 --steps 1 --warmup 1 --flags $fl > $out.json 2> $out.log || { echo "pass failed"; tail -3 $out.log; exit 1; }
    python3 tools/pmc_sum.py $out | grep -E "k_trace" | sed "s/^/flags=$fl  /" | tee -a gpurun_out/$tag/summary.txt
    python3 -c "
import json; d=json.loads(open('$out.json').read().strip().splitlines()[-1]); print('flags=$fl rays', d['rays'], 'steps', d['steps'], 'warmup', d['warmup'])" | tee -a gpurun_out/$tag/summary.txt
  done
done
