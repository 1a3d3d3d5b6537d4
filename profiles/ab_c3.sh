#!/bin/bash
for L in "$@"; do
  n=$(basename $L .so)
  TRGL_LIB=$PWD/$L python3 bench.py --workload c3 --frames-in-flight 1 --end-to-end-frames 0 --writeout-frames 0 2>gpurun_out/ab_c3_$n.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['phase_ms'].items()}, d['parity']['ok'])"
done
