#!/bin/bash
# Detailed SQ counters of the raster kernel (two passes of 8 SQ counters each), default bench workload.
set -e
OUT=gpurun_out/pmc_raster_${1:-x}
rm -rf $OUT
mkdir -p $OUT; export TMPDIR=/tmp
ARGS="bench.py --secondary= --steps 3 --warmup 1 --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_IFETCH --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_WAVES SQ_LEVEL_WAVES SQ_INST_CYCLES_SALU --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_raster" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = sorted(v)[1:] if len(v) > 1 else v      # drop the tiny clear-only launch
    print(f"{k:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
