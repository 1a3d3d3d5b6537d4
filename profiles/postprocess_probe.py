"""Kernel times of the N4 post-process (z image, SSAO, composite) on the head stand-in at 4096x4096: run under
rocprofv3 --kernel-trace --stats (profiles/run via gpurun); prints wall time of trgl_postprocess incl. the D2H copies."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, PHONG, make_uniforms
W = H = 4096
hd = scenes.head_standin(7, W, H)
d, n, s = scenes.procedural_textures(1024)
u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
with Context(W, H, 3) as ctx:
    for slot, t in ((0, d), (1, n), (2, s)): ctx.upload_texture(slot, t)
    ctx.draw(PHONG, hd["clip"], varyings=hd["varyings"], uniforms=u); ctx.sync()
    for it in range(3):
        t0 = time.perf_counter(); out = ctx.postprocess(); dt = time.perf_counter() - t0
        print(f"postprocess (3 images to host): {dt*1e3:.1f} ms")
