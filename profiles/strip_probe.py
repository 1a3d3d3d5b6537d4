"""Per-rank frame time when the context owns 1/G of the rows, on ONE GPU (what a rank of a G-GPU run does, without the RCCL
gather): bounds the strong-scaling curve and shows the load imbalance of each partitioning.
  strips: rank r owns rows [r H/G, (r+1) H/G)                       (trgl_set_strip)
  bands : bands of 128 rows dealt round-robin to the G ranks          (trgl_set_interleave)
Workloads: c4 (10 M random triangles, uniform), c2 / c3 (PHONG head stand-in at 2048 / 4096: the mesh sits in the middle rows).
Prints per (workload, partitioning, G): the per-rank totals, their max (= the frame time a G-GPU run is bound by) and mean."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT, PHONG, make_uniforms

BAND = 128


def workload(name):
    if name == "c4":
        W = H = 4096
        clip, col = scenes.random_triangles(10_000_000, W, H)
        return W, H, FLAT, torch.from_numpy(clip).cuda(), None, torch.from_numpy(col.view(np.int32)).cuda(), None, {}
    W = H = 2048 if name == "c2" else 4096
    hd = scenes.head_standin(7, W, H)
    d, n, s = scenes.procedural_textures(1024)
    tex = {0: d} if name == "c2" else {0: d, 1: n, 2: s}
    slots = (0, -1, -1) if name == "c2" else (0, 1, 2)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, *slots)
    return W, H, PHONG, torch.from_numpy(hd["clip"]).cuda(), torch.from_numpy(hd["varyings"]).cuda(), None, u, tex


for name in (sys.argv[1:] or ["c4", "c2", "c3"]):
    W, H, kind, dclip, dvary, dcol, u, tex = workload(name)
    for part in ("strips", "bands"):
        for G in (1, 2, 4, 8):
            if part == "bands" and G == 1:
                continue
            per_rank = []
            for r in range(G):
                ctx = Context(W, H, 3)
                for k, t in tex.items():
                    ctx.upload_texture(k, t)
                if part == "strips":
                    ctx.set_strip(r * H // G, (r + 1) * H // G)
                else:
                    ctx.set_interleave(BAND, r, G)
                ctx.set_profiling(True)
                frames = 6 if name == "c4" else 12
                for it in range(frames + 2):
                    if it == 2:
                        ctx.reset_phase_ms()
                    ctx.clear(); ctx.draw(kind, dclip, varyings=dvary, colors=dcol, uniforms=u, device=True); ctx.flush()
                ms, n = ctx.phase_ms()
                per_rank.append((ms[0] / n, ms[1] / n, ms[2] / n, ms[3] / n))
                ctx.close()
            tot = [p[3] for p in per_rank]
            worst = per_rank[int(np.argmax(tot))]
            print(f"{name} {part:6s} G={G}: max {max(tot):.3f} ms (setup {worst[0]:.3f} bin {worst[1]:.3f} raster {worst[2]:.3f})  mean {np.mean(tot):.3f}  "
                  f"per rank [{' '.join(f'{t:.3f}' for t in tot)}]", flush=True)
