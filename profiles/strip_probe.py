"""Per-rank frame time of the C4 workload when the context owns 1/G of the rows (what one rank of a G-GPU run does,
without the RCCL gather): bounds the strong-scaling curve from one GPU."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT
W = H = 4096; N = 10_000_000
clip, col = scenes.random_triangles(N, W, H)
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()
for G in (1, 2, 4, 8):
    ctx = Context(W, H, 3); ctx.set_strip(0, H // G); ctx.set_profiling(True)
    for it in range(8):
        if it == 3: ctx.reset_phase_ms()
        ctx.clear(); ctx.draw(FLAT, dclip, colors=dcol, device=True); ctx.flush()
    ms, n = ctx.phase_ms()
    print(f"G={G}: setup {ms[0]/n:.3f}  bin {ms[1]/n:.3f}  raster {ms[2]/n:.3f}  total {ms[3]/n:.3f} ms  pairs {ctx.last_flush_info()['pairs']}")
    ctx.close()
