#!/bin/bash
# Kernel trace of the default bench workload only (per-kernel averages): bash profiles/trace_only.sh <tag>
set -e
OUT=gpurun_out/trace_${1:-x}
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --secondary= --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0 --frames-in-flight 1 > $OUT/bench_trace.log 2>&1
python3 - <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
        acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:60]:60s} calls {len(v):4d}  avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}  total {sum(v)/1e3:8.3f} ms  {100*sum(v)/tot:5.1f} %")
PY
