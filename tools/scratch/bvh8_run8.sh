#!/bin/bash
# GPU box, inside tools/scratch/r8: 4-wide vs 8-wide per-ray kernels of the old tree -- rates, frame times, then SQ_INSTS_VALU per kernel
here=$(cd "$(dirname "$0")" && pwd)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $here
for v in 0 3; do
  export PTAMD_LIB=$here/opencl-path-tracer_amd/csrc/libptamd_bvh8_$v.so
  timeout -k 10 300 python3 measure8.py all 2>/dev/null | tail -1 | tee $out/measure_bvh8_$v.json
  rm -rf $out/pmc_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$v -- python3 measure8.py batch > $out/pmc_$v.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $out/pmc_$v > $out/pmc_bvh8_$v.txt; rm -rf $out/pmc_$v
  cat $out/pmc_bvh8_$v.txt
done
