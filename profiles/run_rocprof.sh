#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_rocprof.sh r01
# 1) kernel trace + stats of the default bench workload, 2) separate PMC passes (FETCH_SIZE / WRITE_SIZE / LDS).
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --secondary= --steps 3 --warmup 1 --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0"
# the kernel trace runs the DEFAULT step count (20 + 3 warm-up), so that its per-kernel averages are those of the bench line
# (a 4-frame run is over before the clocks have settled: k_raster 2.59 instead of 2.43 ms)
TRACE_ARGS="bench.py --secondary= --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $TRACE_ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_pmc_sq.log 2>&1
# 3) the write-out-only frame (clear + flush without triangles) at 4096x4096: k_raster's duration = W*H*11 B leaving the chip
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_writeout -- python3 profiles/writeout_probe.py 4096 > $OUT/writeout.log 2>&1
python3 profiles/summarize_rocprof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
