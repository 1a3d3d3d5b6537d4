set -e
OUT=gpurun_out/trace_c2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload c2 --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0 --frames-in-flight 1 > $OUT/log 2>&1
