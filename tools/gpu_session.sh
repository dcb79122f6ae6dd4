#!/bin/bash
# One GPU-box session (run through gpurun): tools/gpu_session.sh <tag> [steps...]
#   steps: micro  = the VALU issue micro-benchmark            -> gpurun_out/<tag>/valu_issue.md
#          tests  = pytest -m gpu                              -> gpurun_out/<tag>/pytest.log
#          bench  = bench.py (default flags)                   -> gpurun_out/<tag>/bench.json
#          quick  = bench.py without the CPU / frame legs      -> gpurun_out/<tag>/bench_quick.json
#          stats  = tools/trace_stats.py (needs variants/libptamd_stats.so)
#          try:a,b = tools/try.sh a b (variant libraries)
#          "cmd:NAME=command line" = any command, output -> gpurun_out/<tag>/NAME.txt (STEP_TIMEOUT seconds, default 600)
# A step that is killed by its timeout ends the session (no further GPU work after a hang).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out
for step in "$@"; do
  case $step in
    micro) timeout -k 10 240 tools/micro/valu_issue 4096 > $out/valu_issue.md 2> $out/valu_issue.err; rc=$? ;;
    micro2) timeout -k 10 200 tools/micro/valu_issue 4096 49 > $out/valu_issue_f16.md 2> $out/valu_issue_f16.err; rc=$?; cat $out/valu_issue_f16.md | cut -c1-200 ;;
    tests) timeout -k 10 1500 python -m pytest tests -m gpu -q -x --durations=15 > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log ;;
    testsall) timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1; rc=$?; tail -15 $out/pytest.log ;;
    bench) timeout -k 10 420 python bench.py > $out/bench.json 2> $out/bench.err; rc=$?; head -c 700 $out/bench.json; echo; tail -3 $out/bench.err ;;
    frame) timeout -k 10 300 python bench.py --mode frame > $out/frame.json 2> $out/frame.err; rc=$?; python3 -c "
import json; d=json.load(open('$out/frame.json')); print({k: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})" ;;
    vframe:*) PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_${step#vframe:}.so timeout -k 10 300 python bench.py --mode frame > $out/frame_${step#vframe:}.json 2> $out/frame_${step#vframe:}.err; rc=$?; python3 -c "
import json; d=json.load(open('$out/frame_${step#vframe:}.json')); print('${step#vframe:}', {k: v['ms_per_frame'] for k, v in d['frame']['scenes'].items()})" ;;
    quick) timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --steps 4 --warmup 1 --rounds 1 > $out/bench_quick.json 2> $out/bench_quick.err; rc=$?; head -c 400 $out/bench_quick.json; echo ;;
    pytest:*) timeout -k 10 900 python -m pytest ${step#pytest:} -m gpu -q -x > $out/pytest_sel.log 2>&1; rc=$?; tail -8 $out/pytest_sel.log ;;
    vtest:*) PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_${step#vtest:}.so timeout -k 10 900 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py tests/test_gpu_fullsize.py -m gpu -q -x > $out/vtest_${step#vtest:}.log 2>&1; rc=$?; tail -5 $out/vtest_${step#vtest:}.log ;;
    inflight512) timeout -k 10 400 python bench.py --no-cpu-baseline --no-frame --steps 3 --warmup 1 --rounds 2 --in-flight 512 --max-entries 1100000000 > $out/bench_if512.json 2> $out/bench_if512.err; rc=$?; head -c 300 $out/bench_if512.json; echo; tail -2 $out/bench_if512.err ;;
    quick2) timeout -k 10 300 python bench.py --no-cpu-baseline --no-frame --steps 3 --warmup 1 --rounds 2 > $out/bench_quick2.json 2> $out/bench_quick2.err; rc=$?; head -c 300 $out/bench_quick2.json; echo ;;
    share2:*) # two ranks on this one GPU, persistent grids of <n> (per-ray) and <m> (packet) blocks per CU each: share2:n,m
      IFS=, read tb pb <<< "${step#share2:}"
      PTAMD_TRACE_BLOCKS_PER_CU=$tb PTAMD_PACKET_BLOCKS_PER_CU=$pb timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-gpu --backend gloo --no-cpu-baseline --no-frame --steps 3 --warmup 1 --rounds 2 --in-flight 128 > $out/share2_${tb}_${pb}.json 2> $out/share2_${tb}_${pb}.err; rc=$?; head -c 300 $out/share2_${tb}_${pb}.json; echo; tail -2 $out/share2_${tb}_${pb}.err ;;
    sorttest) timeout -k 10 400 python tools/sort_probe.py > $out/sort_test.txt 2>&1; rc=$?; grep -v amdgpu.ids $out/sort_test.txt | tail -12 ;;
    py:*) timeout -k 10 400 python ${step#py:} > $out/py_$(basename ${step#py:} .py).txt 2>&1; rc=$?; grep -v amdgpu.ids $out/py_$(basename ${step#py:} .py).txt | tail -20 ;;
    final) timeout -k 10 1100 tools/final_profiles.sh > $out/final.log 2>&1; rc=$?; tail -5 $out/final.log | cut -c1-600 ;;
    stats:*) PTAMD_LIB=$PWD/opencl-path-tracer_amd/csrc/variants/libptamd_${step#stats:}.so timeout -k 10 300 python tools/trace_stats.py > $out/trace_${step#stats:}.txt 2>&1; rc=$?; grep -v amdgpu.ids $out/trace_${step#stats:}.txt | cut -c1-330 | tail -32 ;;
    stats) timeout -k 10 300 python tools/trace_stats.py > $out/trace_stats.txt 2>&1; rc=$?; tail -30 $out/trace_stats.txt ;;
    try:*) timeout -k 10 600 tools/try.sh $(echo ${step#try:} | tr , ' ') > $out/try.txt 2>&1; rc=$?; cat $out/try.txt ;;
    cmd:*) # cmd:NAME=command line (quote the whole step): output -> gpurun_out/<tag>/NAME.txt, the last 25 lines echoed
      spec=${step#cmd:}; name=${spec%%=*}; line=${spec#*=}
      timeout -k 10 ${STEP_TIMEOUT:-600} bash -c "$line" > $out/$name.txt 2>&1; rc=$?; grep -v amdgpu.ids $out/$name.txt | tail -${STEP_TAIL:-25} | cut -c1-400 ;;
    *) echo "unknown step $step"; rc=0 ;;
  esac
  echo "[$step] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $step timed out: stopping"; exit $rc; fi
done
