#!/bin/bash
# GPU box: kernel trace of the 1-spp interactive frame (bench.py --mode frame)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 300 python bench.py --mode frame --frames 200 > $out/frame.json 2> $out/frame.err || exit 1
python3 -c "
import json; d=json.load(open('$out/frame.json')); print({k: (v['ms_per_frame'], v['rays_per_frame']) for k, v in d['frame']['scenes'].items()})"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --mode frame --frames 40 > $out/trace.json 2> $out/trace.err || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 2 frames of the last scene: print kernel name, duration, gap to previous
tail = rows[-40:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:60]:60s} dur {(e-s)/1e3:8.1f} us  gap {((s-prev)/1e3 if prev else 0):7.1f} us  grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')}")
    prev = e
PY
