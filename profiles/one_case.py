"""Run golden cases in one process, in the order given (debugging aid).  With TRGL_LIB = the diagnostic build, prints its
'list entry is no triangle' counter after every case."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from tests import cases
from tinyrenderder_amd.api import Context
dbg = 'dbg' in os.environ.get('TRGL_LIB', '')
for name in sys.argv[1:]:
    c = cases.CASES[name]()
    print("running", name, flush=True)
    with Context(c["width"], c["height"], c["bpp"]) as ctx:
        ctx.set_viewport(c["viewport"]); ctx.clear(c["clear"], c["zclear"])
        for slot, t in c["textures"].items():
            ctx.upload_texture(slot, t)
        for kind, u, clip, vary, col in c["draws"]:
            ctx.draw(kind, clip, vary, col, u)
        fb = ctx.read_framebuffer(); line = ctx.stats_line()
        if dbg:
            out = (C.c_ulonglong * 16)(); ctx.L.trgl_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
            assert ctx.L.trgl_debug_counters(ctx.h, out) == 0
            print("   bad list entries:", out[9], " candidates", out[0], "visits", out[2])
            if out[10]:
                print("   first bad candidate: list position", out[11] >> 32, "triangle", out[11] & 0xffffffff, " tile slice [", out[12] >> 32, ",", out[12] & 0xffffffff,
                      ") tile", out[13] >> 32, "block", out[13] & 0xffffffff)
    print("ran", name, line.strip(), flush=True)
