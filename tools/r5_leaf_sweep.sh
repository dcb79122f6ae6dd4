#!/bin/bash
# GPU box (round 5): leaf formation inside the 4-wide collapse, swept at run time through PTAMD_LEAF_FORMATION
# ("cap[,inner,leaf0,tri,alpha]", csrc/ptamd.hip collapseKids).  For every setting: the traversal tests first (a wrong tree must show up as a
# failed test under a short timeout, not as a fault in a long run), then the headline measured the short way.
#   tools/r5_leaf_sweep.sh tag "setting" ["setting" ...]        (setting 0 = the leaves are given: rounds 1-4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export PTAMD_COLLAPSE_REPORT=1
first=1
for s in "$@"; do
  name=$(echo "$s" | tr ',.' '__')
  export PTAMD_LEAF_FORMATION="$s"
  if [ "$s" != 0 ] && [ $first = 1 ]; then
    first=0
    # (the stack-bound test asserts the depth of a 100-level chain whose last levels a merged leaf now swallows: deselected here)
    timeout -k 10 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_dynamic.py -m gpu -q -x --deselect tests/test_gpu_intersect.py::test_stack_bound_of_a_root_that_shares_its_subtree_with_an_earlier_root > $out/pytest_$name.log 2>&1; rc=$?
    tail -3 $out/pytest_$name.log
    if [ $rc -ne 0 ]; then echo "tests failed with PTAMD_LEAF_FORMATION=$s: stopping"; exit 1; fi
  fi
  timeout -k 10 240 python bench.py --no-cpu-baseline --no-frame --no-secondary --rounds 1 --steps 4 --warmup 1 > $out/bench_$name.json 2> $out/bench_$name.err; rc=$?
  if [ $rc -ne 0 ]; then echo "bench failed ($rc) with $s"; tail -3 $out/bench_$name.err; [ $rc -eq 124 ] && exit 124; continue; fi
  grep "collapse:" $out/bench_$name.err | sort | uniq -c | cut -c1-260
  python3 -c "
import json; d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernels']
print('LEAF_FORMATION=$s', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', r['family_ms'], 'any-hit', k['k_trace<true>']['avg_launch_ms'], 'closest', k['k_trace<false>']['avg_launch_ms'])"
done
