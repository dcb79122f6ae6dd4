#!/bin/bash
# GPU box: the round's committed evidence -- kernel stats of the default bench, the PMC passes, the bench line; then the same kernel stats
# and PMC passes with every instance ENTERED (--flags 2, PT_FLAG_NO_BAKED_INSTANCES).  Output under gpurun_out/final/ (and final_tl/);
# copy what is quoted into profiles/roundN/ (tools/collect_profiles.sh).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU")
run_set() { # <out dir> <extra bench flags...>
  out=$1; shift
  rm -rf $out && mkdir -p $out
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-frame --no-secondary --steps 2 --warmup 1 "$@" > $out/stats_bench.json 2> $out/stats.log
  echo "$out: stats done"
  for set in "${SETS[@]}"; do
    name=$(echo $set | cut -d' ' -f1)
    # (queues as large as the batch in the counter passes: with the bench's queue fractions the library renders a 16-sample PROBE batch first, whose launches would
    # be counted as a batch of 512 by tools/traffic_json.py; the kernels and what they do per ray are the same either way)
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --no-secondary --rounds 1 --steps 1 --warmup 1 --ext-queue-fraction 0 --shadow-queue-fraction 0 "$@" > $out/pmc_$name.json 2> $out/pmc_$name.log
    python3 tools/pmc_sum.py $out/pmc_$name > $out/pmc_$name.txt
    echo "$out: pmc $name done"
  done
}
run_set gpurun_out/final
if [ -z "$PMC_ONLY" ]; then
timeout -k 10 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.log
echo "bench done"
fi
run_set gpurun_out/final_tl --flags 2
[ -z "$PMC_ONLY" ] && timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --flags 2 > gpurun_out/final_tl/bench.json 2> gpurun_out/final_tl/bench.log
# round 6: the same scene, every instance entered through the GENERAL route (pt_trace.h LEVELS 2; PTAMD_GENERAL_ROUTE=1 forces it on a scene the fold table
# would serve): kernel stats + the instruction and issue counters
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE")
export PTAMD_GENERAL_ROUTE=1
run_set gpurun_out/final_gen --flags 2
[ -z "$PMC_ONLY" ] && timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --flags 2 > gpurun_out/final_gen/bench.json 2> gpurun_out/final_gen/bench.log
unset PTAMD_GENERAL_ROUTE
# (the rank-by-rank emulation of the N = 1 .. 8 jobs is a session of its own: tools/rank_emul.py)
rm -rf gpurun_out/final*/stats/*/*_agent_info.csv
du -sh gpurun_out
[ -f gpurun_out/final/bench.json ] && tail -c 1500 gpurun_out/final/bench.json
