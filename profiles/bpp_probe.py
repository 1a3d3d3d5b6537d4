"""k_raster time on the C4 frame with an RGB (3 B/px) and an RGBA (4 B/px) framebuffer."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT, PHASE_RASTER_KERNEL
W = H = 4096; N = 10_000_000
clip, col = scenes.random_triangles(N, W, H)
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()
for bpp in (3, 4):
    ctx = Context(W, H, bpp); ctx.set_profiling(True)
    for it in range(8):
        if it == 3: ctx.reset_phase_ms()
        ctx.clear(); ctx.draw(FLAT, dclip, colors=dcol, device=True); ctx.flush()
    ms, n = ctx.phase_ms()
    print(f"bpp={bpp}: k_raster {ms[PHASE_RASTER_KERNEL]/n:.3f} ms  total {ms[3]/n:.3f} ms  {ctx.stats_line()}")
    ctx.close()
