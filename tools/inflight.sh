#!/bin/bash
# GPU box: whole-job rate vs samples in flight
for f in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --in-flight $f --steps 3 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in_flight', d['config']['samples_in_flight'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')" || exit 1
done
