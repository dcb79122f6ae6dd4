#!/bin/bash
# GPU box: the per-ray traversal kernels with and without the shared descent (PTAMD_DESCENT = 0 | 3): vector instructions, active-lane cycles, kernel cycles
# per launch, one rocprofv3 --pmc set per run; then kernel stats of either.  tools/r4_descent_pmc.sh tag
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for use in 0 3; do
  export PTAMD_DESCENT=$use
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
    i=$((i+1))
    out=gpurun_out/$tag/pmc_descent${use}_$i
    rm -rf $out && mkdir -p $out
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --no-secondary --rounds 1 --steps 1 --warmup 1 > $out.log 2>&1 || { echo "pass $use/$i failed"; tail -3 $out.log; exit 1; }
    python3 tools/pmc_sum.py $out | grep -E "k_trace<|k_descend" > gpurun_out/$tag/pmc_descent${use}_$i.txt
    rm -rf $out
    cat gpurun_out/$tag/pmc_descent${use}_$i.txt
  done
  out=gpurun_out/$tag/stats_descent$use
  rm -rf $out && mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --no-secondary --rounds 1 --steps 2 --warmup 1 > $out.json 2> $out.log
  cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/kernel_stats_descent$use.csv && rm -rf $out
  head -8 gpurun_out/$tag/kernel_stats_descent$use.csv
done
