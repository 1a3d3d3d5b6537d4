# diagnostic: where k_raster's time goes (outputs are wrong by construction when ABLATE != 0)
for a in 0 1 2 3; do TRGL_RASTER_VARIANT=0 TRGL_RASTER_ABLATE=$a timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate',$a, 'raster ms', round(d['phase_ms']['raster'],3))"; done
