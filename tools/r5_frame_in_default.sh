#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5q; mkdir -p $out
timeout -k 10 400 python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $out/default_short.json 2> $out/default_short.err; python3 -c "
import json; d=json.loads(open('$out/default_short.json').read().strip().splitlines()[-1]); print('default (short)', d['value'], {k[:8]: (v['ms_per_frame'], v['team_kernel_launches_per_frame']) for k, v in d['frame']['scenes'].items()}, d['frame_after_headline'])"
