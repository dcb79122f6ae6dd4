#!/bin/bash
# here, after tools/final_profiles.sh ran on the GPU box: copy the evidence into profiles/<round>/ under a tag: tools/collect_profiles.sh r6f round6
set -e
cd "$(dirname "$0")/.."
tag=$1; dst=profiles/${2:-round6}; mkdir -p $dst
for pair in "final:$tag" "final_tl:${tag}_two_level" "final_gen:${tag}_general_route"; do
  src=gpurun_out/${pair%%:*}; t=${pair##*:}
  [ -d $src ] || continue
  cp $(ls -t $src/stats/*/*kernel_stats.csv | head -1) $dst/${t}_kernel_stats_bench_steps2.csv
  for f in $src/pmc_*.txt; do cp $f $dst/${t}_$(basename $f); done
  python3 -c "
import json,sys
l=[x for x in open('$src/bench.json').read().splitlines() if x.startswith('{')][-1]
json.dump(json.loads(l), open('$dst/${t}_bench.json','w'), indent=1)"
done
[ -f gpurun_out/final/rank_emul.txt ] && grep -v amdgpu.ids gpurun_out/final/rank_emul.txt > $dst/${tag}_rank_emul.txt
python3 tools/isa_mix.py $dst/isa_mix.json > $dst/isa_mix.txt
python3 tools/traffic_json.py gpurun_out/final $dst/traffic.json $tag > /dev/null
[ -d gpurun_out/final_tl ] && python3 tools/traffic_json.py gpurun_out/final_tl $dst/traffic_two_level.json ${tag}_two_level > /dev/null
# (the general route's passes are quoted from their pmc_*.txt: bench.py looks for traffic*.json by scene flags, and flags 2 is the folded route's file)
ls $dst
