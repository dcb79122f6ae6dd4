#!/bin/bash
# PMC passes over one short bench run each (GPU box): tools/pmc.sh tag "CTR CTR .." ["CTR .." ...]
# Each pass is its own rocprofv3 run (--pmc with --kernel-trace only), output under gpurun_out/pmc_<tag>_<i>.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "$@"; do
  i=$((i+1))
  out=gpurun_out/pmc_${tag}_$i
  rm -rf $out
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --no-secondary --rounds 1 --steps 1 --warmup 1 > $out.log 2>&1 || { echo "pass $i failed"; tail -3 $out.log; exit 1; }
  python3 tools/pmc_sum.py $out
done
