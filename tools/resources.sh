#!/bin/bash
# here (no GPU): registers, spills, LDS and occupancy of every kernel of libptamd.so as the compiler reports them: tools/resources.sh [filter] [extra flags...]
cd "$(dirname "$0")/../opencl-path-tracer_amd/csrc"
filter=${1:-.}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" --cuda-device-only -c ptamd.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed -e 's/^.*remark: *//' -e 's/ \[-Rpass.*$//' | awk '
  /^Function Name/ { if (name != "") print name ": " line; name = $3; line = ""; next }
  /VGPRs:|TotalSGPRs|Scratch|Occupancy|VGPRs Spill|LDS Size/ { line = line $0 "  " } END { print name ": " line }' \
  | sed -e 's/_ZN3ptd[0-9]*//' -e 's/EEvNS_[0-9A-Za-z_]*E:/:/' | grep -E "$filter"
