import sys
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT
W = H = 4096
hd = scenes.head_standin(7, W, H)
N = hd["clip"].shape[0]
dclip = torch.from_numpy(hd["clip"]).cuda()
dcol = torch.from_numpy((np.arange(N, dtype=np.uint32) | 0xFF000000).view(np.int32)).cuda()
ctx = Context(W, H, 3)
for it in range(5):
    ctx.clear(); ctx.draw(FLAT, dclip, colors=dcol, device=True); ctx.flush()
ctx.sync()
# empty frames for comparison
for it in range(3):
    ctx.clear(); ctx.flush()
ctx.sync(); ctx.close()
