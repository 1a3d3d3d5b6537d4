"""One-off parity check of flat frames at assorted sizes and pixel formats (1 - 2 M triangles each, perspective w) against the C restatement:
1920x1080 RGBA, 3000x2000, 4097x1025 (odd width, partial tiles), 640x4800 gray, 8192x512.  Prints equal / MISMATCH per frame."""
import sys; sys.path.insert(0,'.')
import numpy as np, time
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT
from oracle import orc
for (W,H,bpp,n,rmax) in [(1920,1080,4,1_500_000,24),(3000,2000,3,2_000_000,16),(4097,1025,3,1_000_000,40),(640,4800,1,800_000,30),(8192,512,3,1_000_000,12)]:
    clip,col=scenes.random_triangles(n,W,H,seed=W+H,rmin=1,rmax=rmax,perspective_w=True)
    with Context(W,H,bpp) as ctx:
        t=time.time(); ctx.draw(FLAT,clip,colors=col); fb,z,st=ctx.read_framebuffer(),ctx.read_zbuffer(),ctx.stats(); tg=time.time()-t
    o=orc.Oracle(W,H,bpp); t=time.time(); o.draw(orc.FLAT,clip,colors=col); to=time.time()-t
    ok=np.array_equal(fb,o.fb) and np.array_equal(z.view(np.uint64),o.z.view(np.uint64)) and st==o.stats
    print(W,H,bpp,n,'equal' if ok else 'MISMATCH', f'gpu(host arrays) {tg:.2f}s oracle {to:.1f}s', flush=True)
