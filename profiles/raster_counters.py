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
names = ["triangles scanned (slots read)", "blocks visited", "blocks with a covered pixel inside the bbox (before the depth-plane test)",
         "lanes of visited blocks inside the bbox", "list entries (pairs)", "visited blocks without a covered pixel",
         "blocks dropped by the per-lane masks", "... of which wrote a pixel (must be 0)",
         "covered lanes (before the depth-plane test)", "... that write their pixel", "blocks with covered pixels that write nothing",
         "... all of whose covered pixels the per-pixel depth-plane test kills (no divisions)", "lanes the depth-plane test would wrongly kill (must be 0)",
         "visited blocks that end at the depth-plane test (no lane inside the bbox survives it)"
         ]
for n, v in zip(names, out): print(f"{n:82s} {v:>12d}  per triangle {v / N:.3f}")
# the diagnostic build counts what the depth-plane test WOULD skip and still runs it; the production kernel:
print(f"{'production kernel: blocks that run the three divisions':82s} {out[2] - out[11]:>12d}  per triangle {(out[2] - out[11]) / N:.3f}")
print(f"{'production kernel: visited blocks that end before the divisions':82s} {out[1] - out[2] + out[11]:>12d}  per triangle {(out[1] - out[2] + out[11]) / N:.3f}")
