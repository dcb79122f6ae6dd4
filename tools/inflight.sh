for f in 16 32 64; do python bench.py --no-cpu-baseline --no-roofline --in-flight $f --rounds 1 --steps 4 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in_flight', d['config']['samples_in_flight'], 'spp/step', d['config']['spp_per_step'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"; done
