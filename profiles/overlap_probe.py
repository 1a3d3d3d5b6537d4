"""Do the HBM-bound phases (setup, binning) overlap with the VALU-bound raster kernel when they run on different
streams?  Two contexts render the C4 frame from two host threads (own stream each; ctypes drops the GIL); compare the
aggregate frame rate with one context alone."""
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT
W = H = 4096; N = 10_000_000
clip, col = scenes.random_triangles(N, W, H)
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()
torch.cuda.synchronize()
def run(ctx, frames):
    for _ in range(frames):
        ctx.clear(); ctx.draw(FLAT, dclip, colors=dcol, device=True); ctx.flush()
    ctx.sync()
a = Context(W, H, 3); b = Context(W, H, 3)
run(a, 3); run(b, 3)
F = 20
t0 = time.perf_counter(); run(a, F); t1 = time.perf_counter() - t0
print(f"one context: {t1 / F * 1e3:.3f} ms per frame")
ta = threading.Thread(target=run, args=(a, F)); tb = threading.Thread(target=run, args=(b, F))
t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); t2 = time.perf_counter() - t0
print(f"two contexts concurrently: {t2 / (2 * F) * 1e3:.3f} ms per frame aggregate ({t1 / F / (t2 / (2 * F)):.2f}x)")
