"""Do two frames in flight overlap on one GPU?  Two contexts, each on its own HIP stream, render the C4 frame alternately; the
HBM-bound phases (k_setup, binning) of one can run under the issue-bound k_raster of the other if the hardware lets them
co-reside.  Prints frames/s for one context alone and for the two together."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT

W = H = 4096
clip, col = scenes.random_triangles(10_000_000, W, H)
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()


def run(nctx, frames):
    ctxs = [Context(W, H, 3) for _ in range(nctx)]
    for c in ctxs:
        c.set_stream(0, use_own=True)
    def frame_begin(c):
        c.clear(); c.draw(FLAT, dclip, colors=dcol, device=True); c.flush_begin()
    for c in ctxs:      # warm
        for _ in range(3):
            frame_begin(c); c.flush_end()
        c.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = []
    for f in range(frames):
        c = ctxs[f % nctx]
        if len(pending) == nctx:
            pending.pop(0).flush_end()
        frame_begin(c); pending.append(c)
    for c in pending:
        c.flush_end()
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fb = [c.read_framebuffer() for c in ctxs]
    same = all(np.array_equal(fb[0], x) for x in fb[1:])
    for c in ctxs:
        c.close()
    print(f"{nctx} context(s): {frames} frames in {dt*1e3:.1f} ms = {dt*1e3/frames:.3f} ms/frame, frames identical: {same}", flush=True)


run(1, 20)
run(2, 40)
run(3, 60)
