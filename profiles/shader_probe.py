import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT, PHONG, EYE, GOURAUD, make_uniforms
W = H = 4096
hd = scenes.head_standin(7, W, H)
d_, n_, s_ = scenes.procedural_textures(1024)
N = hd["clip"].shape[0]
dclip = torch.from_numpy(hd["clip"]).cuda(); dvary = torch.from_numpy(hd["varyings"]).cuda()
dcol = torch.from_numpy((np.arange(N, dtype=np.uint32) | 0xFF000000).view(np.int32)).cuda()
dint = torch.from_numpy(np.ascontiguousarray(hd["varyings"][:, :3])).cuda()
def run(name, kind, vary, col, u, tex):
    ctx = Context(W, H, 3)
    for k, t in tex.items(): ctx.upload_texture(k, t)
    ctx.set_profiling(True)
    for it in range(7):
        if it == 2: ctx.reset_phase_ms()
        ctx.clear(); ctx.draw(kind, dclip, varyings=vary, colors=col, uniforms=u, device=True); ctx.flush()
    ms, n = ctx.phase_ms(); print(name, 'raster ms', round(ms[2]/n, 3), 'total', round(ms[3]/n, 3), ctx.last_flush_info()); ctx.close()
base = (hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0)
run('flat', FLAT, None, dcol, None, {})
run('gouraud', GOURAUD, dint, dcol, None, {})
run('phong no maps', PHONG, dvary, None, make_uniforms(*base, -1, -1, -1), {})
run('phong diffuse', PHONG, dvary, None, make_uniforms(*base, 0, -1, -1), {0: d_})
run('phong all maps', PHONG, dvary, None, make_uniforms(*base, 0, 1, 2), {0: d_, 1: n_, 2: s_})
run('eye', EYE, dvary, None, make_uniforms(*base, 0, -1, 2), {0: d_, 2: s_})
