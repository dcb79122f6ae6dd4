#!/bin/bash
# GPU box: kernel trace of the last frame of ONE interactive scene (tools/r5_frame_trace.sh glass [copper ...])
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for scene in "$@"; do
  out=gpurun_out/r5frame/$scene; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --mode frame --frames 40 --frame-scene $scene > $out/trace.json 2> $out/trace.err || { tail -3 $out/trace.err; exit 1; }
  python3 - <<PY
import csv, glob
f = glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
print("scene $scene", open("$out/trace.json").read().strip()[-260:])
prev = None
for r in rows[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:58]:58s} dur {(e-s)/1e3:8.1f} us  gap {((s-prev)/1e3 if prev else 0):7.1f} us  grid {r.get('Grid_Size_X','?')}")
    prev = e
PY
done
