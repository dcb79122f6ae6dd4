#!/bin/bash
# GPU box: (1) kernel trace of the 1-spp interactive frame; (2) how coherent the first instance of a packet is on the two-level scene
# (k_descend stops at the first instance reference: its followers there = rays of a packet that enter ONE instance together)
tag=$1; mkdir -p gpurun_out/$tag
bash tools/r3_frame_prof.sh $tag > gpurun_out/$tag/frame_trace.txt 2>&1; tail -45 gpurun_out/$tag/frame_trace.txt
timeout -k 10 300 python tools/descend_stats.py 2 64 > gpurun_out/$tag/descend_stats_entered.txt 2>&1; grep -A3 "descent 3" gpurun_out/$tag/descend_stats_entered.txt
