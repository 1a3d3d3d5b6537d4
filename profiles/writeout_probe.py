"""Frame write-out alone: trgl_clear + trgl_flush with no triangles = k_raster's tile-in (clear values) and tile-out on
every tile, W*H*11 bytes leaving the chip once.  BASELINE north_star: >= 50 % of the HBM roofline on framebuffer + z
writes at 4096x4096."""
import sys
sys.path.insert(0, '.')
import torch
from tinyrenderder_amd.api import Context, PHASE_RASTER_KERNEL, PHASE_TOTAL
for W in ([int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]):
    H = W
    for bpp in (3, 4):
        ctx = Context(W, H, bpp); ctx.set_profiling(True)
        for it in range(25):
            if it == 5: ctx.reset_phase_ms()
            ctx.clear(); ctx.flush()
        ms, n = ctx.phase_ms()
        b = W * H * (8 + bpp)
        print(f"{W}x{H} bpp={bpp}: k_raster {ms[PHASE_RASTER_KERNEL]/n*1e3:.1f} us  flush total {ms[PHASE_TOTAL]/n*1e3:.1f} us  "
              f"{b/1e6:.1f} MB -> {b/(ms[PHASE_RASTER_KERNEL]/n*1e-3)/1e9:.0f} GB/s = {b/(ms[PHASE_RASTER_KERNEL]/n*1e-3)/8e12*100:.1f} % of 8 TB/s")
        ctx.close()
