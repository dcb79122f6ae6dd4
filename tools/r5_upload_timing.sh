#!/bin/bash
# GPU box: where the host time of a rebuilt scene per frame goes (PTAMD_UPLOAD_TIMING=1: one stderr line per upload stage, csrc/ptamd.hip StageTimer)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5upload; mkdir -p $out
PTAMD_UPLOAD_TIMING=1 timeout -k 10 300 python tools/rebuild_timing.py > $out/rebuild.txt 2> $out/rebuild.err || { tail -5 $out/rebuild.err; exit 1; }
cat $out/rebuild.txt; grep "ptamd\]" $out/rebuild.err | tail -12
