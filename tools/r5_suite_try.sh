#!/bin/bash
# GPU box: the whole -m gpu suite, then the headline the short way (tools/r5_try.sh with the variants named on the command line)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5suite; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -6 $out/pytest.log | cut -c1-200
[ $rc -ne 0 ] && { echo "tests failed: stopping"; exit 1; }
bash tools/r5_try.sh "$@"
