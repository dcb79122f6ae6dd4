#!/bin/bash
# GPU box: the round's committed evidence -- kernel stats of the default bench, the PMC traffic passes, the bench line.
# Output under gpurun_out/final/ ; copy what is quoted into profiles/roundN/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
rm -rf $out && mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-frame --steps 2 --warmup 1 > $out/stats_bench.json 2> $out/stats.log
echo "stats done"
for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --rounds 1 --steps 1 --warmup 1 > $out/pmc_$name.json 2> $out/pmc_$name.log
  python3 tools/pmc_sum.py $out/pmc_$name > $out/pmc_$name.txt
  echo "pmc $name done"
done
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.log
tail -c 3000 $out/bench.json
