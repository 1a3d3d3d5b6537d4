#!/bin/bash
# A/B of library builds on the default bench workload: bash profiles/ab_bench.sh <tag> lib1.so lib2.so ...
# (each build is checked against the committed full-size digests of the reference's frame by bench.py's parity gate)
TAG=$1; shift
mkdir -p gpurun_out
for L in "$@"; do
  n=$(basename $L .so)
  TRGL_LIB=$PWD/$L python3 bench.py --secondary= --cpu-sample 0 --end-to-end-frames 0 --writeout-frames 0 --frames-in-flight 1 > gpurun_out/ab_${TAG}_$n.json 2> gpurun_out/ab_${TAG}_$n.err
  python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_${TAG}_$n.json").read().strip().splitlines()[-1])
    print("$n", "ms_per_step %.3f" % d["ms_per_step"], {k: round(v, 3) for k, v in d["phase_ms"].items()}, "parity", d["parity"].get("ok"), d["parity"].get("golden"))
except Exception as e:
    print("$n", "FAILED", e, open("gpurun_out/ab_${TAG}_$n.err").read()[-800:])
PY
done
