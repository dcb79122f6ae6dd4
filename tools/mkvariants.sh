#!/bin/bash
# Build tuning variants of libptamd.so (here, no GPU needed): tools/mkvariants.sh name "-DFLAG=.." [name flags]...
set -e
cd "$(dirname "$0")/../opencl-path-tracer_amd/csrc"
mkdir -p variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt -Xarch_device -fno-slp-vectorize -Xarch_host -msse4.1 $flags ptamd.hip -o variants/libptamd_$name.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: _ZN3ptd7k_traceILb0" | grep -E "VGPRs:|Spill|Occupancy|LDS Size" | tr '\n' ' ' | sed "s/^/$name: /"; echo ) &
done
wait
