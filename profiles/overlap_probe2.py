"""Can the HBM-bound first half of a flush (k_setup + binning) run UNDER the issue-bound k_raster of the frame before it?
Two contexts on two streams render the C4 frame alternately with the order enforced by events: raster(f) on one stream starts when
raster(f-1) on the other has ended, and the first half of frame f+1 is queued behind raster(f-1) on that other stream - so every raster
has exactly one first half to share the machine with, and two rasters never run together (which is all that two free-running contexts
achieve, profiles/overlap_probe.py).  Prints ms per frame for the serial loop and for the enforced overlap."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT

W = H = 4096
clip, col = scenes.random_triangles(10_000_000, W, H)
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()


def frame_begin(c):
    c.clear(); c.draw(FLAT, dclip, colors=dcol, device=True); c.flush_begin()


def serial(frames):
    c = Context(W, H, 3)
    for _ in range(3):
        frame_begin(c); c.flush_end()
    c.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame_begin(c); c.flush_end()
    c.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fb = c.read_framebuffer(); c.close()
    print(f"serial: {dt*1e3/frames:.3f} ms/frame", flush=True)
    return fb


def overlapped(frames, prio=(0, 0)):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    st = [torch.cuda.Stream(priority=p) for p in prio]
    ctxs = [Context(W, H, 3) for _ in range(2)]
    for c, s in zip(ctxs, st):
        c.set_stream(s.cuda_stream)
        for _ in range(3):
            frame_begin(c); c.flush_end()
        c.sync()
    torch.cuda.synchronize()
    ev = None
    t0 = time.perf_counter()
    frame_begin(ctxs[0])
    for f in range(frames):
        X, Y = ctxs[f % 2], ctxs[(f + 1) % 2]
        sX = st[f % 2]
        if f + 1 < frames:
            frame_begin(Y)                  # first half of frame f+1, behind raster(f-1) on Y's stream
        if ev is not None:
            sX.wait_event(ev)               # raster(f) starts when raster(f-1) has ended
        X.flush_end()
        ev = torch.cuda.Event(); ev.record(sX)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fbs = [c.read_framebuffer() for c in ctxs]
    for c in ctxs:
        c.close()
    print(f"overlap (stream priorities {prio}): {dt*1e3/frames:.3f} ms/frame", flush=True)
    return fbs


ref = serial(20)
for prio in ((0, 0), (-1, -1)):
    fbs = overlapped(40, prio)
    print("   frames identical to the serial one:", all(np.array_equal(ref, x) for x in fbs), flush=True)
