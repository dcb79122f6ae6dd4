"""here (no GPU): where does a kernel of libptamd.so spill?  Lists its scratch loads / stores with the loop depth of their basic block.
    python tools/spills.py k_traceILb0ELi2 [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opencl-path-tracer_amd", "csrc")
name, extra = sys.argv[1], sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", *extra,
                      "-S", "--cuda-device-only", "ptamd.hip", "-o", "-"], cwd=CSRC, check=True, capture_output=True, text=True).stdout
cur, depth, block, counts, total = False, 0, "", {}, {}
for line in asm.splitlines():
    m = re.match(r"^(_ZN3ptd\S+):", line)
    if m:
        cur = name in m.group(1)
        continue
    if line.startswith(".Lfunc_end"):
        cur = False
    if not cur:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m or re.match(r"^; %bb\.\d+:", line):
        d = re.search(r"Depth=(\d+)", line)
        depth, block = (int(d.group(1)) if d else 0), (m.group(1) if m else "entry")
    m = re.match(r"\s+(\w+)", line)
    if m:
        total[depth] = total.get(depth, 0) + 1
        if m.group(1).startswith("scratch_"):
            counts[(depth, block, m.group(1))] = counts.get((depth, block, m.group(1)), 0) + 1
print("instructions per loop depth:", dict(sorted(total.items())))
for (d, b, op), n in sorted(counts.items()):
    print(f"depth {d} {b:12s} {op:24s} x{n}")
