#!/bin/bash
# GPU box: every rank's share of the N = 1, 2, 4, 8 jobs of config 4 emulated on this GPU, weak and strong (tools/rank_emul.py), round 5's final build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5emul; mkdir -p $out
for sc in weak strong; do
  timeout -k 10 500 python tools/rank_emul.py --config 4 --scaling $sc --all-ranks --json $out/rank_emul_config4_$sc.json > $out/rank_emul_config4_$sc.txt 2>&1 || { echo "rank_emul $sc failed"; tail -5 $out/rank_emul_config4_$sc.txt; exit 1; }
  grep "^==" $out/rank_emul_config4_$sc.txt
done
