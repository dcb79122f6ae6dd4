#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mode in "$@"; do
  out=gpurun_out/r5l/team$mode; mkdir -p $out
  export PTAMD_TEAM_ROUNDS=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --mode frame --frames 40 > $out/trace.json 2> $out/trace.err || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
print("PTAMD_TEAM_ROUNDS=$mode")
prev = None
for r in rows[-22:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:58]:58s} dur {(e-s)/1e3:8.1f} us  gap {((s-prev)/1e3 if prev else 0):7.1f} us  grid {r.get('Grid_Size_X','?')}")
    prev = e
PY
done
