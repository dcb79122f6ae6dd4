#!/bin/bash
# GPU box: the headline the short way with the instanced meshes built by each of the host library's builders (how much is the tree's quality worth?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5builder; mkdir -p $out
for b in spatial binned fast; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1 --builder $b > $out/$b.json 2> $out/$b.err || { tail -3 $out/$b.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$out/$b.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$b', d['value'], 'Mrays/s', {n.split('<')[0]+n[-6:]: v['ms_per_step'] for n, v in k.items()})"
done
