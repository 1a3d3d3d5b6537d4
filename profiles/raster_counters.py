"""k_raster work counters on the C4 frame (diagnostic build: make -C tinyrenderder_amd/csrc ../libtrgl_dbg.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from tinyrenderder_amd import scenes
W = H = 4096; N = 10_000_000
clip, col = scenes.random_triangles(N, W, H)
L = C.CDLL('tinyrenderder_amd/libtrgl_dbg.so')
h = C.c_void_p(); assert L.trgl_create(0, W, H, 3, C.byref(h)) == 0
dclip = torch.from_numpy(clip).cuda(); dcol = torch.from_numpy(col.view(np.int32)).cuda()
L.trgl_draw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
assert L.trgl_draw(h, 0, None, dclip.data_ptr(), None, dcol.data_ptr(), N, 1) == 0
out = (C.c_ulonglong * 16)(); L.trgl_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
assert L.trgl_debug_counters(h, out) == 0
names = ["candidates (list entries whose bbox reaches the wave's block)", "survivors of the cull = visits", "visits", "lanes of visits inside (bbox n block)",
         "visits with a covered lane (deferred)", "covered lanes that survive the depth plane (deferred fragments)", "resolves", "lanes resolved",
         "fragments written"]
for n, v in zip(names, out): print(f"{n:82s} {v:>12d}  per triangle {v / N:.3f}")
print(f"{'list entries that are no triangle of the flush (must be 0)':82s} {out[9]:>12d}")
tot = sum(out[k] for k in (10, 11, 12, 13, 15))
for k, n in ((10, "startup + left-over"), (11, "list steps"), (12, "cull rounds"), (13, "visits (with resolves)"), (14, "... of which resolves"), (15, "block out")):
    print(f"{'wave cycles: ' + n:82s} {out[k]:>14d}  {100.0 * out[k] / max(tot, 1):5.1f} %")
print(f"{'lanes per resolve':82s} {out[7] / max(out[6], 1):>12.2f}")
print(f"{'covered lanes per covered visit':82s} {out[5] / max(out[4], 1):>12.2f}")
