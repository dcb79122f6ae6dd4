#!/bin/bash
# GPU box, one session: the whole -m gpu suite (parity margins land in gpurun_out/parity_margins.json), then the rank-by-rank emulation of the
# N = 1, 2, 4, 8 jobs (config 4 and config 5, weak and strong).  tools/r4_session.sh tag
tag=$1
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$tag/pytest_gpu.log
cp gpurun_out/parity_margins.json gpurun_out/$tag/parity_margins.json 2>/dev/null
for cfg in 4 5; do for sc in weak strong; do
  timeout -k 10 400 python tools/rank_emul.py --config $cfg --scaling $sc --all-ranks --json gpurun_out/$tag/rank_emul_config${cfg}_$sc.json > gpurun_out/$tag/rank_emul_config${cfg}_$sc.txt 2>&1 || { echo "rank_emul $cfg $sc failed"; tail -5 gpurun_out/$tag/rank_emul_config${cfg}_$sc.txt; }
  grep "^==" gpurun_out/$tag/rank_emul_config${cfg}_$sc.txt
done; done
