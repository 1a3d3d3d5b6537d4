set -e
OUT=gpurun_out/trace_post; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/postprocess_probe.py > $OUT/log 2>&1
tail -3 $OUT/log
