"""Named parity scenes shared by the golden-fixture generator, the oracle tests and the GPU tests.

Each builder returns a dict: width, height, bpp, viewport (4x4), draws = [(kind, uniforms|None, clip,
varyings|None, colors|None)], textures = {slot: array}, clear (4 bytes), zclear.
Inputs come from tinyrenderder_amd.scenes (bit-reproducible everywhere).
"""
import numpy as np

from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import FLAT, GOURAUD, PHONG, EYE, CHECKER, make_uniforms

DEFAULT_CLEAR = (0, 0, 0, 255)


def _case(w, h, draws, bpp=3, viewport=None, textures=None, clear=DEFAULT_CLEAR, zclear=np.inf):
    return dict(width=w, height=h, bpp=bpp, viewport=scenes.init_viewport(0, 0, w, h) if viewport is None else viewport,
                draws=draws, textures=textures or {}, clear=tuple(clear), zclear=zclear)


def flat_small_64():
    clip, col = scenes.random_triangles(300, 64, 64, seed=11, rmin=2, rmax=32)
    return _case(64, 64, [(FLAT, None, clip, None, col)])


def flat_800():          # BASELINE config 0 shape: 800x800, flat, CPU-runnable
    clip, col = scenes.random_triangles(100_000, 800, 800, seed=12, rmin=1, rmax=16)
    return _case(800, 800, [(FLAT, None, clip, None, col)])


def flat_persp_512():
    clip, col = scenes.random_triangles(20_000, 512, 512, seed=13, rmin=2, rmax=64, perspective_w=True)
    return _case(512, 512, [(FLAT, None, clip, None, col)])


def flat_big_tris_512():
    clip, col = scenes.random_triangles(400, 512, 512, seed=14, rmin=64, rmax=512)
    return _case(512, 512, [(FLAT, None, clip, None, col)])


def edge_256():
    clip, col = scenes.edge_case_triangles(256, 256)
    return _case(256, 256, [(FLAT, None, clip, None, col)])


def grid_256():
    clip, col = scenes.shared_edge_grid(8, 8, 256, 256)
    return _case(256, 256, [(FLAT, None, clip, None, col)])


def grid_fine_128():
    clip, col = scenes.shared_edge_grid(32, 32, 128, 128, z_slope=0.0)   # all z equal: every shared pixel is a tie
    return _case(128, 128, [(FLAT, None, clip, None, col)])


def gouraud_256_rgba():
    clip, col = scenes.random_triangles(5000, 256, 256, seed=15, rmin=2, rmax=64, perspective_w=True)
    inten = scenes.SplitMix64(5).uniform(5000 * 3, -0.2, 1.3).reshape(5000, 3)
    return _case(256, 256, [(GOURAUD, None, clip, inten, col)], bpp=4)


def _head(level, w, h, tex):
    hd = scenes.head_standin(level, w, h)
    d, n, s = scenes.procedural_textures(tex)
    return hd, {0: d, 1: n, 2: s}


def phong_512():
    hd, tx = _head(4, 512, 512, 256)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
    return _case(512, 512, [(PHONG, u, hd["clip"], hd["varyings"], None)], textures=tx)


def phong_nomaps_256():
    hd, _ = _head(3, 256, 256, 64)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 0.5, -1, -1, -1)
    return _case(256, 256, [(PHONG, u, hd["clip"], hd["varyings"], None)])


def eye_256():
    hd, tx = _head(3, 256, 256, 128)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, -1, 2)
    return _case(256, 256, [(EYE, u, hd["clip"], hd["varyings"], None)], textures=tx)


def multi_draw_320x200():
    """main.cpp's frame shape: background PHONG (strength 0.5), PHONG head, EYE on top, flat overlay."""
    w, h = 320, 200
    hd, tx = _head(3, w, h, 128)
    big = scenes.head_standin(2, w, h, seed=99, distance=1.6)
    u_bg = make_uniforms(big["model_view"], big["key"], big["fill"], big["rim"], 0.5, 0, 1, -1)
    u_hd = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
    u_ey = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, -1, -1)
    small = scenes.head_standin(2, w, h, seed=5, distance=4.0)
    fclip, fcol = scenes.random_triangles(500, w, h, seed=17, rmin=2, rmax=24)
    return _case(w, h, [(PHONG, u_bg, big["clip"], big["varyings"], None),
                        (PHONG, u_hd, hd["clip"], hd["varyings"], None),
                        (EYE, u_ey, small["clip"], small["varyings"], None),
                        (FLAT, None, fclip, None, fcol)], textures=tx, clear=(30, 20, 10, 255))


def odd_dims_101x67():
    clip, col = scenes.random_triangles(2000, 101, 67, seed=18, rmin=1, rmax=24)
    return _case(101, 67, [(FLAT, None, clip, None, col)])


def gray_bpp1_96x64():
    clip, col = scenes.random_triangles(1500, 96, 64, seed=19, rmin=1, rmax=24)
    return _case(96, 64, [(FLAT, None, clip, None, col)], bpp=1)


def viewport_offset_256x160():
    clip, col = scenes.random_triangles(3000, 200, 120, seed=20, rmin=1, rmax=32)
    return _case(256, 160, [(FLAT, None, clip, None, col)], viewport=scenes.init_viewport(16, 8, 200, 120))


def zclear_finite_128():
    clip, col = scenes.random_triangles(3000, 128, 128, seed=21, rmin=2, rmax=32)
    return _case(128, 128, [(FLAT, None, clip, None, col)], bpp=4, clear=(7, 8, 9, 10), zclear=0.25)


def huge_depths_128():
    """NDC depths up to 1.7e308 on vertices 1 and 2 (vertex 0 stays inside [-1, 1], so our_gl.cpp:103-106 keeps the
    triangle): the interpolated depth (our_gl.cpp:156-158) overflows to +-inf or lands on huge finite values, and
    our_gl.cpp:160 drops the non-finite ones.  The screen coordinates stay ordinary ('well scaled')."""
    clip, col = scenes.random_triangles(600, 128, 128, seed=23, rmin=4, rmax=48)
    clip = clip.copy()
    clip[0::5, 6] = 1.7e308;   clip[0::5, 10] = 1.7e308       # sum overflows to +inf
    clip[1::5, 6] = -1.7e308;  clip[1::5, 10] = -1.7e308      # ... to -inf (would win every z-test if it were written)
    clip[2::5, 6] = 1.7e308;   clip[2::5, 10] = -1.7e308      # huge cancelling terms: finite or not depending on the pixel
    clip[3::5, 6] = -1e300                                     # huge but finite: wins where covered
    return _case(128, 128, [(FLAT, None, clip, None, col)])


def checker_256():
    """The kind that discards (our_gl.cpp:187-188): fragments whose perspective-correct barycentrics fall on odd checker cells
    write nothing - no depth, no colour, no counter - so what lies behind them shows through and later fragments are tested against
    the depth they left alone.  Perspective w makes the perspective-correct barycentrics differ from the screen-space ones."""
    clip, col = scenes.random_triangles(4000, 256, 256, seed=31, rmin=4, rmax=48, perspective_w=True)
    return _case(256, 256, [(CHECKER, make_uniforms(cells=6), clip, None, col)])


def checker_mixed_200x120():
    """Discarding and ordinary draws in one flush (per-fragment kinds): flat below, checker in the middle, Gouraud on top."""
    w, h = 200, 120
    c0, k0 = scenes.random_triangles(800, w, h, seed=32, rmin=4, rmax=40)
    c1, k1 = scenes.random_triangles(1500, w, h, seed=33, rmin=4, rmax=40, perspective_w=True)
    c2, k2 = scenes.random_triangles(600, w, h, seed=34, rmin=2, rmax=24, perspective_w=True)
    v2 = scenes.SplitMix64(35).uniform(600 * 3, 0.1, 1.2).reshape(600, 3)
    return _case(w, h, [(FLAT, None, c0, None, k0), (CHECKER, make_uniforms(cells=3), c1, None, k1), (GOURAUD, None, c2, v2, k2)], bpp=4)


def empty_scene_64():
    return _case(64, 64, [])


CASES = {f.__name__: f for f in (
    flat_small_64, flat_800, flat_persp_512, flat_big_tris_512, edge_256, grid_256, grid_fine_128, gouraud_256_rgba,
    phong_512, phong_nomaps_256, eye_256, multi_draw_320x200, odd_dims_101x67, gray_bpp1_96x64,
    viewport_offset_256x160, zclear_finite_128, huge_depths_128, empty_scene_64, checker_256, checker_mixed_200x120)}

# cases whose full buffers are stored in tests/golden/ (small enough to commit)
FULL_BUFFER_CASES = ("flat_small_64", "odd_dims_101x67", "gray_bpp1_96x64")


def run_oracle(case, strip=None):
    """Render a case with the CPU oracle; returns (fb, z, stats tuple)."""
    from oracle import orc
    o = orc.Oracle(case["width"], case["height"], case["bpp"], viewport=case["viewport"], clear_bgra=case["clear"],
                   z_clear=case["zclear"], strip=strip)
    for slot, t in case["textures"].items():
        o.upload_texture(slot, t)
    for kind, u, clip, vary, col in case["draws"]:
        ou = None
        if u is not None:
            ou = orc.Uniforms.from_buffer_copy(bytes(u))
        o.draw(kind, clip, vary, col, ou)
    return o.fb, o.z, o.stats


def run_gpu(case, strip=None, split=None):
    """Render a case through the C ABI on the GPU; split=k submits each draw in k flushes."""
    from tinyrenderder_amd.api import Context
    with Context(case["width"], case["height"], case["bpp"]) as ctx:
        ctx.set_viewport(case["viewport"])
        ctx.clear(case["clear"], case["zclear"])
        if strip is not None:
            ctx.set_strip(*strip)
        for slot, t in case["textures"].items():
            ctx.upload_texture(slot, t)
        for kind, u, clip, vary, col in case["draws"]:
            n = clip.shape[0]
            parts = 1 if not split else split
            edges = [n * i // parts for i in range(parts + 1)]
            for a, b in zip(edges[:-1], edges[1:]):
                if a == b:
                    continue
                ctx.draw(kind, clip[a:b], None if vary is None else vary[a:b], None if col is None else col[a:b], u)
                if split:
                    ctx.flush()
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
        line = ctx.stats_line()
    return fb, z, st, line


# ---- BASELINE configs[3] / [4] at their stated sizes (C4 = 10 M random triangles at 4096^2,
# C5 = the C4-style scene at 8192^2, N = 10 M).  Too slow for the scalar oracle inside a test: the reference's own
# rasterize() rendered them once (tests/golden/make_golden_fullsize.py) and tests/golden/golden_fullsize.json holds the
# digests of its framebuffer bytes, z-buffer bits and its print_render_stats() line.
def c4_4096_10m():
    clip, col = scenes.random_triangles(10_000_000, 4096, 4096)          # = bench.py's default workload
    return _case(4096, 4096, [(FLAT, None, clip, None, col)])


def c5_8192_10m():
    clip, col = scenes.random_triangles(10_000_000, 8192, 8192, seed=0x5EED0005, rmin=2, rmax=40)
    return _case(8192, 8192, [(FLAT, None, clip, None, col)])


FULLSIZE_CASES = {f.__name__: f for f in (c4_4096_10m, c5_8192_10m)}
