for w in c3 c2; do for sl in 0 8 16 24 48 96 200 1000000; do
  if [ $sl = 0 ]; then unset TRGL_SPLIT_LEN; else export TRGL_SPLIT_LEN=$sl; fi
  python3 bench.py --workload $w --frames-in-flight 1 --end-to-end-frames 0 --cpu-sample 0 --writeout-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w split_len=$sl', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['phase_ms'].items()})"
done; done
