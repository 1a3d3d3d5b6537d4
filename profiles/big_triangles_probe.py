"""Stress outside the benchmark's regime: large triangles (64..512 px) at 4096x4096, ~100 tiles per triangle, tile lists of
several thousand entries.  Checks determinism and split-submission invariance at full size and the oracle on a prefix."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT
from oracle import orc
W = H = 4096; N = 1_000_000
clip, col = scenes.random_triangles(N, W, H, seed=77, rmin=64, rmax=512)
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()
res = []
for parts in (1, 1, 3):
    with Context(W, H, 3) as ctx:
        ctx.set_profiling(True)
        edges = [N * i // parts for i in range(parts + 1)]
        t0 = time.perf_counter()
        for a, b in zip(edges[:-1], edges[1:]):
            ctx.draw(FLAT, dclip[a:b], colors=dcol[a:b], device=True); ctx.flush()
        ctx.sync(); dt = time.perf_counter() - t0
        ms, n = ctx.phase_ms()
        res.append((scenes.digest(ctx.read_framebuffer()), scenes.digest(ctx.read_zbuffer()), ctx.stats()))
        print(f"parts={parts}: {dt*1e3:.1f} ms, phases/flush {[round(m/n,3) for m in ms]}, pairs(last flush) {ctx.last_flush_info()['pairs']}, {ctx.stats_line()}")
assert res[0] == res[1] == res[2], "non-deterministic or split-dependent"
M = 4000
with Context(W, H, 3) as ctx:
    ctx.draw(FLAT, clip[:M], colors=col[:M]); fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
o = orc.Oracle(W, H, 3); o.draw(orc.FLAT, clip[:M], colors=col[:M])
assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and np.array_equal(fb, o.fb) and st == o.stats
print("ok: deterministic, split-invariant, prefix equals oracle")
