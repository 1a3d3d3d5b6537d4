set -e
OUT=gpurun_out/pmc_rw; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="bench.py --secondary= --steps 3 --warmup 1 --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/w.log 2>&1
python3 profiles/summarize_rocprof.py $OUT | grep -E "k_raster|HBM traffic"
python3 bench.py --secondary= --cpu-sample 0 --writeout-frames 0 --no-parity --end-to-end-frames 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phase_ms'])"
