set -e
OUT=gpurun_out/trace_vs; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/vertex_stage_probe.py > $OUT/log 2>&1
grep faces $OUT/log
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("k_vertex_stage","k_setup"):
            if k in n: acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in acc.items(): print(n, [round(x,1) for x in v])
PY
