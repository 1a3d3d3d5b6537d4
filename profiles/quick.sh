#!/bin/bash
# quick A/B on the GPU box: bench line (phases only) + work counters of the diagnostic build
ARGS="--secondary= --cpu-sample 0 --writeout-frames 0 --end-to-end-frames 0 --steps 20 --warmup 3"
python3 bench.py $ARGS "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in d['phase_ms'].items()}, 'parity', d.get('parity', {}).get('ok'))"
