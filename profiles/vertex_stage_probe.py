"""k_vertex_stage on the head stand-in (327680 faces, indexed): run under rocprofv3 --kernel-trace (profiles/trace via gpurun)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, PHONG, make_uniforms
import test_next_rows as T
W = H = 4096
hd, verts, idx = T._indexed_head(7, W, H)
d, n, s = scenes.procedural_textures(1024)
u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
dv = torch.from_numpy(verts).cuda(); di = torch.from_numpy(idx.view(np.int32)).cuda()
with Context(W, H, 3) as ctx:
    for slot, t in ((0, d), (1, n), (2, s)): ctx.upload_texture(slot, t)
    for it in range(5):
        ctx.clear(); ctx.draw_indexed(PHONG, u, hd["projection"], dv, di, device=True); ctx.flush(); ctx.sync()
    print("faces", idx.shape[0], "vertices", verts.shape, ctx.stats_line())
