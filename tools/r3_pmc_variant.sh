#!/bin/bash
# GPU box: VALU instruction and lane counters of the shadow-ray kernel for library variants: tools/r3_pmc_variant.sh tag variant...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
for v in "$@"; do
  lib=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_$v.so
  [ "$v" = base ] && lib=$PWD/opencl-path-tracer_amd/csrc/libptamd.so
  export PTAMD_LIB=$lib
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
    name=$(echo $set | cut -d' ' -f1)
    out=gpurun_out/$tag/${v}_$name
    rm -rf $out; mkdir -p $out
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-roofline --no-frame --no-secondary --rounds 1 --steps 1 --warmup 1 > $out.json 2> $out.log || exit 1
    python3 tools/pmc_sum.py $out | grep "k_trace<true" | sed "s/^/$v  /"
  done
done
