#!/bin/bash
# GPU box: frame times of the in-tree library under different settings of an environment variable: tools/r5_frames_env.sh PTAMD_TEAM_USE 7 3 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5frames; mkdir -p $out
var=$1; shift
for v in "$@" "$1"; do
  env $var=$v timeout -k 10 240 python bench.py --mode frame --frames 200 > $out/env_$v.json 2> $out/env_$v.err || { tail -3 $out/env_$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$out/env_$v.json').read().strip().splitlines()[-1]); print('$var=$v', {k[:7]: (v['ms_per_frame'], v['team_kernel_launches_per_frame']) for k, v in d['frame']['scenes'].items()})"
done
