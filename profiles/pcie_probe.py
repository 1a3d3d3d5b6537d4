"""PCIe-inclusive frame time: the C4 triangle stream handed over as HOST arrays (trgl_draw TRGL_MEM_HOST copies them
before returning), plus read-back of the finished framebuffer.  Never the bench `value`; quoted in DESIGN.md."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT
W = H = 4096; N = 10_000_000
clip, col = scenes.random_triangles(N, W, H)
pin_clip = torch.from_numpy(clip).pin_memory().numpy(); pin_col = torch.from_numpy(col.view(np.int32)).pin_memory().numpy().view(np.uint32)
ctx = Context(W, H, 3)
for name, a, b in (("pageable host arrays", clip, col), ("pinned host arrays", pin_clip, pin_col)):
    for it in range(4):
        if it == 1: t0 = time.perf_counter()
        ctx.clear(); ctx.draw(FLAT, a, colors=b); ctx.flush(); ctx.sync()
    dt = (time.perf_counter() - t0) / 3
    t1 = time.perf_counter(); fb = ctx.read_framebuffer(); t2 = time.perf_counter()
    print(f"{name}: {dt*1e3:.1f} ms per frame incl. H2D of {clip.nbytes/1e9:.2f} GB -> {N/dt/1e6:.0f} Mtri/s; framebuffer D2H {1e3*(t2-t1):.1f} ms")
ctx.close()
