"""GPU parity tests: the HIP path through the C ABI vs the golden vectors (reference output) and the CPU oracle.

Bar: z-buffer bit-identical, framebuffer byte-identical, stats tuple identical.  The only
tolerance is for EYE (std::pow with exponent 8 vs device pow): at most 1 LSB per colour byte on at most 0.1 % of
pixels; z and stats stay exact.
"""
import json
import os

import numpy as np
import pytest

import cases
from oracle import orc
from tinyrenderder_amd import scenes
from tinyrenderder_amd.api import Context, FLAT

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))
GOLDEN_FULL = json.load(open(os.path.join(HERE, "golden", "golden_fullsize.json")))
POW_CASES = {"eye_256", "multi_draw_320x200"}      # contain EYE fragments


def _assert_fb(fb, ref, name):
    if name in POW_CASES and not np.array_equal(fb, ref):
        d = np.abs(fb.astype(np.int16) - ref.astype(np.int16))
        assert d.max() <= 1, f"{name}: colour differs by more than 1 LSB"
        assert (d.max(axis=-1) > 0).mean() <= 1e-3, f"{name}: more than 0.1 % of pixels differ"
    else:
        bad = np.argwhere(fb != ref)
        assert bad.size == 0, f"{name}: {len(bad)} framebuffer bytes differ, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_gpu_matches_reference_golden_and_oracle(name):
    case = cases.CASES[name]()
    g = GOLDEN[name]
    fb, z, st, line = cases.run_gpu(case)
    ofb, oz, ost = cases.run_oracle(case)
    bad = np.argwhere(z.view(np.uint64) != oz.view(np.uint64))
    assert bad.size == 0, f"{name}: {len(bad)} z values differ, first at {bad[:5].tolist()}"
    _assert_fb(fb, ofb, name)
    assert st == ost
    assert line == g["stats"]
    assert scenes.digest(z) == g["z"]
    if name not in POW_CASES:
        assert scenes.digest(fb) == g["fb"]


@pytest.mark.parametrize("name", ["flat_persp_512", "multi_draw_320x200", "grid_fine_128"])
def test_split_submission_is_invisible(name):
    """Submitting the same triangles over several flushes (read-modify-write of tiles) changes nothing."""
    case = cases.CASES[name]()
    fb, z, st, _ = cases.run_gpu(case)
    fb2, z2, st2, _ = cases.run_gpu(case, split=3)
    assert np.array_equal(z.view(np.uint64), z2.view(np.uint64))
    assert np.array_equal(fb, fb2)
    assert st == st2


@pytest.mark.parametrize("cut", [200, 256, 31])
def test_strips_compose(cut):
    """Multi-GPU shard, run sequentially on one GPU: two strip contexts give the rows of the whole image."""
    case = cases.CASES["flat_persp_512"]()
    fb, z, st, _ = cases.run_gpu(case)
    h = case["height"]
    fb0, z0, s0, _ = cases.run_gpu(case, strip=(0, cut))
    fb1, z1, s1, _ = cases.run_gpu(case, strip=(cut, h))
    assert np.array_equal(np.concatenate([fb0[:cut], fb1[cut:]]), fb)
    assert np.array_equal(np.concatenate([z0[:cut], z1[cut:]]).view(np.uint64), z.view(np.uint64))
    assert s0[1] + s1[1] == st[1] and min(s0[6], s1[6]) == st[6] and max(s0[7], s1[7]) == st[7]
    assert s0[2:6] == st[2:6]
    # and the strip itself matches the oracle restricted to the same rows
    ofb, oz, os_ = cases.run_oracle(case, strip=(0, cut))
    assert np.array_equal(fb0[:cut], ofb[:cut]) and s0 == os_


def test_c4_prefix_4096_vs_oracle():
    """BASELINE config 3 at full resolution, 300k-triangle prefix (the oracle needs ~1.5 s for it)."""
    W = H = 4096
    clip, col = scenes.random_triangles(300_000, W, H)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats


def test_c4_full_size_properties():
    """10 M triangles at 4096x4096 (BASELINE config 3): too slow for the scalar oracle, so check the
    size-independent properties: determinism, split-submission invariance, and the counters' invariants."""
    import torch
    W = H = 4096
    N = 10_000_000
    clip, col = scenes.random_triangles(N, W, H)
    dclip = torch.from_numpy(clip).cuda()
    dcol = torch.from_numpy(col.view(np.int32)).cuda()
    res = []
    for parts in (1, 1, 4):
        with Context(W, H, 3) as ctx:
            edges = [N * i // parts for i in range(parts + 1)]
            for a, b in zip(edges[:-1], edges[1:]):
                ctx.draw(FLAT, dclip[a:b], colors=dcol[a:b], device=True)
                ctx.flush()
            res.append((scenes.digest(ctx.read_framebuffer()), scenes.digest(ctx.read_zbuffer()), ctx.stats()))
    assert res[0] == res[1], "two identical runs differ"
    assert res[0] == res[2], "4 flushes differ from 1 flush"
    st = res[0][2]
    assert st[0] == N and st[2:6] == (0, 0, W - 1, H - 1) and st[1] > 0 and -1.0 <= st[6] < st[7] <= 1.0


def test_c4_full_size_equals_the_reference_frame():
    """BASELINE configs[3] at its stated size — the frame bench.py times: all 10 M triangles at 4096x4096 against the
    frame the reference's own rasterize() rendered (tests/golden/golden_fullsize.json: sha256 of its framebuffer bytes and
    z-buffer bits, its print_render_stats() line)."""
    import torch
    g = GOLDEN_FULL["c4_4096_10m"]
    case = cases.FULLSIZE_CASES["c4_4096_10m"]()
    _, _, clip, _, col = case["draws"][0]
    dclip = torch.from_numpy(clip).cuda()
    dcol = torch.from_numpy(col.view(np.int32)).cuda()
    with Context(4096, 4096, 3) as ctx:
        ctx.draw(FLAT, dclip, colors=dcol, device=True)
        fb, z, line = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats_line()
    assert line == g["stats"]
    assert scenes.digest(z) == g["z"]
    assert scenes.digest(fb) == g["fb"]


@pytest.mark.parametrize("seed", range(6))
def test_block_masks_on_adversarial_shapes(seed):
    """Shapes chosen against the conservative per-block tests of k_raster (edge functions at block corners, depth-plane
    minimum vs the block's stored maximum): needles with aspect ratios up to 1e7, triangles of area 1e-9..1e-4 px^2 (u.z
    next to the 1e-12 limit), depth planes as steep as [-1, 1] across a fraction of a pixel, a vertex thousands of pixels
    off screen, all on top of dense overdraw so that the stored maxima are low and the masks actually drop blocks.  Any
    block dropped wrongly shows up as a missing fragment: z bits, colours and counters must equal the oracle's."""
    rng = scenes.SplitMix64(7000 + seed)
    W, H = 160, 96
    n = 12000
    clip, col = scenes.random_triangles(n, W, H, seed=7100 + seed, rmin=2, rmax=40)
    clip = clip.copy()
    u = rng.uniform(n * 4).reshape(n, 4)
    ndc_px = 2.0 / W
    for i in range(0, n, 3):                                   # every third triangle is replaced by an adversarial one
        kind = (i // 3) % 4
        cx, cy = clip[i, 0], clip[i, 1]
        ang = 6.283185307179586 * u[i, 0]
        dx, dy = np.cos(ang), np.sin(ang)
        if kind == 0:                                          # needle: long axis 5..60 px, width 1e-6..1e-2 px
            L = (5 + 55 * u[i, 1]) * ndc_px; wdt = 10.0 ** (-6 + 4 * u[i, 2]) * ndc_px
            p = [(cx - L * dx, cy - L * dy), (cx + L * dx, cy + L * dy), (cx - wdt * dy, cy + wdt * dx)]
        elif kind == 1:                                        # tiny: edges of 1e-5..1e-2 px
            e = 10.0 ** (-5 + 3 * u[i, 1]) * ndc_px
            p = [(cx, cy), (cx + e * dx, cy + e * dy), (cx - e * dy, cy + e * dx)]
        elif kind == 2:                                        # ordinary footprint, depth plane as steep as it gets
            r = (1 + 6 * u[i, 1]) * ndc_px
            p = [(cx + r * np.cos(ang + a), cy + r * np.sin(ang + a)) for a in (0.0, 2.1, 4.2)]
        else:                                                  # one vertex far off screen
            r = (2 + 10 * u[i, 1]) * ndc_px; far = 50.0 + 5000.0 * u[i, 2]
            p = [(cx, cy), (cx + r * dx, cy + r * dy), (cx - far * dy, cy + far * dx)]
        # counter-clockwise so that the back-face test keeps it
        (x0, y0), (x1, y1), (x2, y2) = p
        if (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0) < 0:
            p[1], p[2] = p[2], p[1]
        zs = [-1.0, 1.0, -1.0 + 2.0 * u[i, 3]] if kind == 2 else [2.0 * u[i, 3] - 1.0, 2.0 * u[i, 1] - 1.0, 2.0 * u[i, 2] - 1.0]
        for v in range(3):
            clip[i, 4 * v + 0], clip[i, 4 * v + 1], clip[i, 4 * v + 2], clip[i, 4 * v + 3] = p[v][0], p[v][1], zs[v], 1.0
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats


@pytest.mark.parametrize("seed", range(6))
def test_depth_plane_early_test_on_extreme_depths(seed):
    """k_raster skips the three divisions of a block when a plane c0 + u.y g1 + u.x g2 (a lower bound of every covered pixel's
    computed z, 2^-40 max|z_i| below the exact plane) is not below the stored depth.  Depths chosen against it, on dense
    overdraw: vertices at +-1e0..1e300 next to one inside [-1,1] (slopes overflow: the test must switch itself off), depths of
    1e-300..1e-320 (subnormal planes), planes one ulp from flat, exactly flat duplicates (ties), perspective w, and a stored
    z-buffer with NaN, -inf, +inf and finite patches.  A pixel killed wrongly is a missing fragment: bits must equal the oracle's."""
    rng = scenes.SplitMix64(8000 + seed)
    W, H = 192, 128
    n = 15000
    clip, col = scenes.random_triangles(n, W, H, seed=8100 + seed, rmin=2, rmax=30, perspective_w=bool(seed & 1))
    clip = clip.copy()
    u = rng.uniform(n * 4).reshape(n, 4)
    for i in range(n):
        kind = i % 8
        w = clip[i, [3, 7, 11]]
        z = clip[i, [2, 6, 10]] / w                        # NDC depths of the generator, inside [-1, 1]
        if kind == 0:                                      # two huge vertices, every sign
            e1, e2 = 300.0 * u[i, 0], 300.0 * u[i, 1]
            z = np.array([z[0], (1 if u[i, 2] < 0.5 else -1) * 10.0 ** e1, (1 if u[i, 3] < 0.5 else -1) * 10.0 ** e2])
        elif kind == 1:                                    # tiny, down to subnormals
            z = z * 10.0 ** (-300.0 - 20.0 * u[i, 0])
        elif kind == 2:                                    # one ulp from flat
            z = np.array([z[0], np.nextafter(z[0], 2.0), z[0] if u[i, 0] < 0.5 else np.nextafter(z[0], -2.0)])
        elif kind == 3:                                    # exactly flat, few distinct values: ties between triangles
            z = np.full(3, np.round(z[0] * 8) / 8)
        elif kind == 4:                                    # one vertex beyond the far plane, one beyond the near plane
            z = np.array([z[0], 1.0 + 50.0 * u[i, 0], -1.0 - 50.0 * u[i, 1]])
        clip[i, [2, 6, 10]] = z * w
    # stored depths the flush starts from: finite band, NaN, -inf, +inf columns
    z0 = np.full((H, W), np.inf)
    z0[:, 0:40] = 0.25; z0[:, 40:70] = np.nan; z0[:, 70:100] = -np.inf; z0[:, 100:130] = -0.5; z0[10:20, :] = 1e-310
    with Context(W, H, 3) as ctx:
        ctx.clear((1, 2, 3, 255))
        ctx.write_zbuffer(z0)
        half = n // 2
        ctx.draw(FLAT, clip[:half], colors=col[:half]); ctx.flush()
        ctx.draw(FLAT, clip[half:], colors=col[half:])
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3, clear_bgra=(1, 2, 3, 255))
    o.z[:] = z0
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats
    assert st[1] > 50_000


@pytest.mark.parametrize("bpp", [1, 4])
def test_phong_and_eye_on_gray_and_rgba_framebuffers(bpp):
    """TGAImage::set copies bytespp bytes of the returned colour (tgaimage.cpp:32-39): k_shade must do the same on a
    1-byte and a 4-byte framebuffer (the golden PHONG / EYE cases are RGB)."""
    from tinyrenderder_amd.api import PHONG, EYE, make_uniforms
    w, h = 200, 136
    hd = scenes.head_standin(3, w, h)
    d, n, sp = scenes.procedural_textures(64)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
    for kind in (PHONG, EYE):
        case = cases._case(w, h, [(kind, u, hd["clip"], hd["varyings"], None)], bpp=bpp, textures={0: d, 1: n, 2: sp}, clear=(30, 20, 10, 200))
        ofb, oz, ost = cases.run_oracle(case)
        fb, z, st, _ = cases.run_gpu(case)
        assert np.array_equal(z.view(np.uint64), oz.view(np.uint64)) and st == ost
        if kind == PHONG:
            assert np.array_equal(fb, ofb)
        else:                       # EYE: pow(x, 8) may differ in the last ulp -> at most 1 LSB on at most 0.1 % of the bytes
            diff = np.abs(fb.astype(np.int16) - ofb.astype(np.int16))
            assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3


def test_flush_in_two_halves():
    """trgl_flush_begin (setup + binning) / trgl_flush_end (raster) give what trgl_flush gives; any other entry point
    called in between completes the begun flush first (here: a second draw, then a read-back)."""
    case = cases.CASES["flat_persp_512"]()
    kind, u, clip, vary, col = case["draws"][0]
    ofb, oz, ost = cases.run_oracle(case)
    half = clip.shape[0] // 2
    with Context(case["width"], case["height"], case["bpp"]) as ctx:
        ctx.draw(kind, clip[:half], colors=col[:half])
        ctx.flush_begin()
        ctx.flush_begin()                                   # a second begin is a no-op
        ctx.flush_end()
        ctx.flush_end()                                     # so is a second end
        ctx.draw(kind, clip[half:], colors=col[half:])
        ctx.flush_begin()
        fb = ctx.read_framebuffer()                         # completes the begun flush
        z, st = ctx.read_zbuffer(), ctx.stats()
    assert np.array_equal(fb, ofb) and np.array_equal(z.view(np.uint64), oz.view(np.uint64)) and st == ost
    with Context(case["width"], case["height"], case["bpp"]) as ctx:
        ctx.draw(kind, clip[:half], colors=col[:half])
        ctx.flush_begin()
        ctx.draw(kind, clip[half:], colors=col[half:])      # completes the begun flush, then queues
        assert np.array_equal(ctx.read_framebuffer(), ofb)


def test_two_phong_draws_in_one_flush_and_strips():
    """The deferred PHONG path (k_raster visibility + k_shade) with TWO PHONG draws of different uniforms / varyings in one
    flush (k_shade serves one draw at a time inside a wave), whole frame and cut into strips of odd heights (row bands,
    strip-clipped tiles), against the oracle."""
    from tinyrenderder_amd.api import PHONG, make_uniforms
    w, h = 320, 200
    hd = scenes.head_standin(3, w, h)
    big = scenes.head_standin(2, w, h, seed=99, distance=1.6)
    d, n, sp = scenes.procedural_textures(128)
    u_bg = make_uniforms(big["model_view"], big["key"], big["fill"], big["rim"], 0.5, 0, 1, -1)
    u_hd = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
    case = cases._case(w, h, [(PHONG, u_bg, big["clip"], big["varyings"], None), (PHONG, u_hd, hd["clip"], hd["varyings"], None)],
                       textures={0: d, 1: n, 2: sp}, clear=(30, 20, 10, 255))
    ofb, oz, ost = cases.run_oracle(case)
    fb, z, st, _ = cases.run_gpu(case)
    assert np.array_equal(z.view(np.uint64), oz.view(np.uint64)) and np.array_equal(fb, ofb) and st == ost
    for y0, y1 in ((0, 77), (77, 131), (131, h)):
        sfb, sz, _, _ = cases.run_gpu(case, strip=(y0, y1))
        assert np.array_equal(sfb[y0:y1], ofb[y0:y1]) and np.array_equal(sz[y0:y1].view(np.uint64), oz[y0:y1].view(np.uint64))


def test_c5_8192_eight_strips_compose():
    """BASELINE configs[4] at its stated size on one GPU: 10 M triangles on an 8192x8192 frame, cut into the 8 horizontal
    strips the 8 ranks of `bench.py --gpus 8` own and rendered one after the other by strip contexts.  The unsharded frame
    equals the frame the reference's own rasterize() rendered (golden_fullsize.json), and the strips equal its rows,
    depths, summed fragment counts and z range.  (What RCCL then does with the strips is a plain all-gather.)"""
    import torch
    g = GOLDEN_FULL["c5_8192_10m"]
    case = cases.FULLSIZE_CASES["c5_8192_10m"]()
    W = H = 8192
    G = 8
    _, _, clip, _, col = case["draws"][0]
    dclip = torch.from_numpy(clip).cuda()
    dcol = torch.from_numpy(col.view(np.int32)).cuda()
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, dclip, colors=dcol, device=True)
        fb, z, st, line = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats(), ctx.stats_line()
    assert line == g["stats"]
    assert scenes.digest(z) == g["z"]
    assert scenes.digest(fb) == g["fb"]
    frags, zmin, zmax = 0, np.inf, -np.inf
    for r in range(G):
        y0, y1 = H * r // G, H * (r + 1) // G
        with Context(W, H, 3) as ctx:
            ctx.set_strip(y0, y1)
            ctx.draw(FLAT, dclip, colors=dcol, device=True)
            sfb, sz, sst = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
        assert np.array_equal(sfb[y0:y1], fb[y0:y1]), f"strip {r}: colours"
        assert np.array_equal(sz[y0:y1].view(np.uint64), z[y0:y1].view(np.uint64)), f"strip {r}: depths"
        assert sst[0] == st[0] and sst[2:6] == st[2:6]
        frags += sst[1]; zmin = min(zmin, sst[6]); zmax = max(zmax, sst[7])
    assert (frags, zmin, zmax) == (st[1], st[6], st[7])


def test_readback_roundtrip_and_zbuffer_restore():
    """main.cpp:700,730 copies the z-buffer before the eyes pass and restores it afterwards."""
    case = cases.CASES["flat_small_64"]()
    kind, u, clip, vary, col = case["draws"][0]
    with Context(64, 64, 3) as ctx:
        ctx.draw(kind, clip[:150], colors=col[:150])
        z_before = ctx.read_zbuffer()
        ctx.draw(kind, clip[150:], colors=col[150:])
        fb_after = ctx.read_framebuffer()
        ctx.write_zbuffer(z_before)
        assert np.array_equal(ctx.read_zbuffer().view(np.uint64), z_before.view(np.uint64))
        assert np.array_equal(ctx.read_framebuffer(), fb_after)
    ofb, _, _ = cases.run_oracle(case)
    assert np.array_equal(fb_after, ofb)


def test_clear_between_frames_and_error_paths():
    from tinyrenderder_amd.api import TrglError, PHONG
    clip, col = scenes.random_triangles(500, 96, 64, seed=3, rmin=2, rmax=20)
    with Context(96, 64, 4) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        a = ctx.read_framebuffer()
        ctx.clear((1, 2, 3, 4), 0.0)          # nothing passes z < 0 ... except negative z
        ctx.draw(FLAT, clip, colors=col)
        b, zb = ctx.read_framebuffer(), ctx.read_zbuffer()
        o = orc.Oracle(96, 64, 4, clear_bgra=(1, 2, 3, 4), z_clear=0.0)
        o.draw(orc.FLAT, clip, colors=col)
        assert np.array_equal(b, o.fb) and np.array_equal(zb.view(np.uint64), o.z.view(np.uint64))
        assert not np.array_equal(a, b)
        with pytest.raises(TrglError):
            ctx.draw(PHONG, clip, varyings=np.zeros((500, 24)))   # PHONG without uniforms
    with pytest.raises(TrglError):
        Context(0, 10, 3)


def test_exact_division_shortcuts_selftest():
    """k_raster divides by the per-triangle constant u.z with FMAs and decides coverage from signs; both must be
    bit-identical to IEEE division.  2e9 random + adversarial operand pairs against the hardware divide."""
    with Context(64, 64, 3) as ctx:
        assert ctx.selftest_division(2_000_000_000, seed=12345) == 0


def test_badly_scaled_triangles_take_the_literal_path():
    """Vertices at 1e250 px (legal: the bbox clamp keeps the triangle) overflow the edge products; such triangles
    are not 'well scaled' and must follow the reference's literal arithmetic (inf/NaN semantics included)."""
    W = H = 128
    clip, col = scenes.random_triangles(400, W, H, seed=31, rmin=4, rmax=64)
    clip = clip.copy()
    clip[::7, 0] = -1e250          # one vertex absurdly far left
    clip[3::11, 5] = -1e200        # or far below
    clip[5::13, 0:2] *= 1e-300     # or collapsing towards the origin (tiny edge deltas)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats


def test_literal_and_well_scaled_flushes_alternate_on_one_context():
    """k_setup counts the triangles of a flush that are not well scaled; the host launches the raster kernel without the literal
    path only when there are none.  A flush with such triangles, then one without, then one with again on the same context (the
    count is per flush): every frame must equal the oracle's."""
    W, H = 160, 128
    n = 4000
    clip, col = scenes.random_triangles(n, W, H, seed=515, rmin=2, rmax=30)
    bad = clip.copy()
    bad[::7, [0, 1, 4, 5, 8, 9]] *= 1e-260                    # edge deltas below 2^-250: not well scaled, still drawn by the reference's arithmetic
    bad[3::11, 0] = 1e250                                      # a vertex beyond 2^200
    with Context(W, H, 3) as ctx:
        o = orc.Oracle(W, H, 3)
        for k, batch in enumerate((bad, clip, bad, clip)):
            part = batch[k * 1000:(k + 1) * 1000]; pc = col[k * 1000:(k + 1) * 1000]
            ctx.draw(FLAT, part, colors=pc); ctx.flush()
            o.draw(orc.FLAT, part, colors=pc)
            assert np.array_equal(ctx.read_zbuffer().view(np.uint64), o.z.view(np.uint64)), k
            assert np.array_equal(ctx.read_framebuffer(), o.fb), k
        assert ctx.stats() == o.stats


def test_bench_rccl_strip_gather_path_single_rank():
    """bench.py's N>1 code path (strip context + in-place RCCL all-gather on the context's own framebuffer memory,
    one stream shared with torch) rehearsed with a 1-rank process group: a 1-GPU box cannot host two RCCL ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                        "--triangles", "300000", "--size", "1024", "--cpu-sample", "0", "--secondary="], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["rccl_ranks"] == 1 and len(d["rank_phase_ms"]) == 1 and d["rank_phase_ms"][0]["raster_kernel"] > 0     # what the driver's SCALE record can be checked against


@pytest.mark.parametrize("cfg", ["c2_2048_phong_diffuse", "c3_4096_phong_diffuse_normal_spec"])
def test_baseline_phong_configs_full_size(cfg):
    """BASELINE configs[1] and [2] at full resolution: PHONG on the 327 680-triangle head stand-in (african_head.obj
    is absent from the reference tree) with 1024x1024 procedural maps, against the CPU oracle: z bit-identical,
    colour byte-identical (PHONG's pow exponent is always 1.0, so no tolerance is needed), stats identical."""
    from tinyrenderder_amd.api import PHONG, make_uniforms
    size = 2048 if cfg.startswith("c2") else 4096
    hd = scenes.head_standin(7, size, size)
    d, n, s = scenes.procedural_textures(1024)
    slots = (0, -1, -1) if cfg.startswith("c2") else (0, 1, 2)
    tex = {0: d} if cfg.startswith("c2") else {0: d, 1: n, 2: s}
    args = (hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0) + slots
    with Context(size, size, 3) as ctx:
        for k, t in tex.items():
            ctx.upload_texture(k, t)
        ctx.draw(PHONG, hd["clip"], hd["varyings"], uniforms=make_uniforms(*args))
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(size, size, 3)
    for k, t in tex.items():
        o.upload_texture(k, t)
    o.draw(orc.PHONG, hd["clip"], hd["varyings"], uniforms=orc.make_uniforms(*args))
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats
    assert st[1] > 100_000          # the head covers a good part of the screen


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_flat_scenes_against_oracle(seed):
    """Randomised scenes for the depth shortcuts (hierarchical Z, early z, sign coverage, FMA division): dense
    overdraw, perspective w, depths partly outside [-1,1], exact ties and duplicates, finite z clears, several
    flushes onto the same tiles, odd image sizes, strips."""
    rng = scenes.SplitMix64(1000 + seed)
    u = rng.uniform(16)
    W = int(64 + u[0] * 300); H = int(64 + u[1] * 300)
    if seed % 3 == 0:
        W, H = (W // 4) * 4, (H // 2) * 2
    n = int(2000 + u[2] * 30000)
    clip, col = scenes.random_triangles(n, W, H, seed=2000 + seed, rmin=1 + 3 * u[3], rmax=8 + 120 * u[4],
                                        perspective_w=bool(seed & 1))
    clip = clip.copy()
    zs = 0.2 + 2.0 * u[5]                                  # stretch depths so some vertices leave [-1,1]
    for v in range(3):
        clip[:, 4 * v + 2] *= zs
    k = n // 7
    clip[k:2 * k] = clip[0:k]                              # exact duplicates: ties must keep the earlier triangle
    clip[3 * k:4 * k, [2, 6, 10]] = np.round(clip[3 * k:4 * k, [2, 6, 10]] * 4) / 4 * clip[3 * k:4 * k, [3, 7, 11]]   # few distinct depths
    zclear = np.inf if seed % 4 else float(0.3 * (u[6] - 0.5))
    strip = None if seed % 5 else (H // 3, H - H // 5)
    parts = 1 + seed % 3
    with Context(W, H, 3) as ctx:
        ctx.clear((9, 8, 7, 255), zclear)
        if strip:
            ctx.set_strip(*strip)
        edges = [n * i // parts for i in range(parts + 1)]
        for a, b in zip(edges[:-1], edges[1:]):
            ctx.draw(FLAT, clip[a:b], colors=col[a:b]); ctx.flush()
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3, clear_bgra=(9, 8, 7, 255), z_clear=zclear, strip=strip)
    o.draw(orc.FLAT, clip, colors=col)
    rows = slice(None) if strip is None else slice(*strip)
    assert np.array_equal(z[rows].view(np.uint64), o.z[rows].view(np.uint64))
    assert np.array_equal(fb[rows], o.fb[rows])
    assert st == o.stats


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_gouraud_scenes_against_oracle(seed):
    """The GOURAUD kernel (three waves per SIMD, 10-chunk slots, perspective-correct barycentrics, colours from the varyings riding in
    the batch registers) on dense overdraw with perspective w, extreme depths and two flushes: bits equal the oracle's."""
    from tinyrenderder_amd.api import GOURAUD
    rng = scenes.SplitMix64(9100 + seed)
    u = rng.uniform(8)
    W, H = int(96 + 200 * u[0]), int(96 + 160 * u[1])
    n = int(4000 + 16000 * u[2])
    bpp = (3, 4, 1, 3)[seed % 4]
    clip, col = scenes.random_triangles(n, W, H, seed=9200 + seed, rmin=2, rmax=20 + 60 * u[3], perspective_w=True)
    clip = clip.copy()
    inten = scenes.SplitMix64(9300 + seed).uniform(n * 3, -0.3, 1.4).reshape(n, 3)
    clip[::5, [2, 6, 10]] *= 1.0 + 3.0 * u[4]                  # some depths beyond the clip range
    clip[1::9, 6] = clip[1::9, 7] * 1e6                          # a far vertex: steep depth planes
    half = n // 2
    with Context(W, H, bpp) as ctx:
        ctx.draw(GOURAUD, clip[:half], varyings=inten[:half], colors=col[:half]); ctx.flush()
        ctx.draw(GOURAUD, clip[half:], varyings=inten[half:], colors=col[half:])
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, bpp)
    o.draw(orc.GOURAUD, clip, inten, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_phong_soup_against_oracle(seed):
    """PHONG on a triangle soup with dense overdraw (the head stand-in is a closed surface: two layers at most): the deferred
    path keeps, per pixel, the LAST triangle that passed the z-test, and k_shade recomputes its barycentrics.  Random varyings
    (uv outside [0,1), unnormalised normals, degenerate tangent frames), perspective w, depth ties from duplicated triangles,
    two flushes, strips on half of the seeds."""
    from tinyrenderder_amd.api import PHONG, make_uniforms
    rng = scenes.SplitMix64(9500 + seed)
    u8 = rng.uniform(8)
    W, H = int(96 + 160 * u8[0]), int(96 + 160 * u8[1])
    n = int(3000 + 9000 * u8[2])
    clip, _ = scenes.random_triangles(n, W, H, seed=9600 + seed, rmin=2, rmax=20 + 50 * u8[3], perspective_w=True)
    clip = clip.copy()
    vr = scenes.SplitMix64(9700 + seed)
    uv = vr.uniform(n * 6, -0.5, 1.5).reshape(n, 6)
    pos = vr.uniform(n * 9, -2.0, 2.0).reshape(n, 9)
    nrm = vr.uniform(n * 9, -1.0, 1.0).reshape(n, 9)
    nrm[::17] = 0.0                                             # zero normals: normalized() returns them unchanged
    uv[::13, 2:4] = uv[::13, 0:2]                               # degenerate uv triangles: the tangent frame falls back
    vary = np.ascontiguousarray(np.concatenate([uv, pos, nrm], 1))
    k = n // 6
    clip[k:2 * k] = clip[0:k]; vary[k:2 * k] = vary[0:k][::-1]  # same geometry, other attributes: the earlier triangle must win the tie
    d, nm, sp = scenes.procedural_textures(64)
    hd = scenes.head_standin(1, W, H)
    uni = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 0.7, 0, 1, 2)
    strip = None if seed % 2 else (H // 4, H - H // 3)
    half = n // 2
    with Context(W, H, 3) as ctx:
        for slot, t in ((0, d), (1, nm), (2, sp)):
            ctx.upload_texture(slot, t)
        if strip:
            ctx.set_strip(*strip)
        ctx.draw(PHONG, clip[:half], varyings=vary[:half], uniforms=uni); ctx.flush()
        ctx.draw(PHONG, clip[half:], varyings=vary[half:], uniforms=uni)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3, strip=strip)
    for slot, t in ((0, d), (1, nm), (2, sp)):
        o.upload_texture(slot, t)
    o.draw(orc.PHONG, clip, vary, uniforms=orc.Uniforms.from_buffer_copy(bytes(uni)))
    rows = slice(None) if strip is None else slice(*strip)
    assert np.array_equal(z[rows].view(np.uint64), o.z[rows].view(np.uint64))
    assert np.array_equal(fb[rows], o.fb[rows])
    assert st == o.stats


def test_mixed_flush_with_gouraud_runs_the_any_kernel():
    """One flush holding GOURAUD, PHONG, FLAT and EYE draws: k_raster<ANY> serves it, and its GOURAUD branch reads the
    varyings and base colours through the draw descriptor (d.vary + local * K), not from the batch registers."""
    from tinyrenderder_amd.api import GOURAUD, PHONG, EYE, make_uniforms
    w, h = 288, 176
    hd = scenes.head_standin(3, w, h)
    d, n, sp = scenes.procedural_textures(64)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
    gclip, gcol = scenes.random_triangles(3000, w, h, seed=71, rmin=2, rmax=48, perspective_w=True)
    ginten = scenes.SplitMix64(72).uniform(3000 * 3, -0.2, 1.3).reshape(3000, 3)
    fclip, fcol = scenes.random_triangles(800, w, h, seed=73, rmin=2, rmax=24)
    small = scenes.head_standin(2, w, h, seed=5, distance=4.0)
    case = cases._case(w, h, [(GOURAUD, None, gclip[:1500], ginten[:1500], gcol[:1500]), (PHONG, u, hd["clip"], hd["varyings"], None),
                              (GOURAUD, None, gclip[1500:], ginten[1500:], gcol[1500:]), (FLAT, None, fclip, None, fcol),
                              (EYE, u, small["clip"], small["varyings"], None)],
                       textures={0: d, 1: n, 2: sp}, clear=(3, 2, 1, 255))
    ofb, oz, ost = cases.run_oracle(case)
    for bpp in (3,):
        fb, z, st, _ = cases.run_gpu(case)
        assert np.array_equal(z.view(np.uint64), oz.view(np.uint64)) and st == ost
        diff = np.abs(fb.astype(np.int16) - ofb.astype(np.int16))           # EYE: pow(x, 8), at most 1 LSB on at most 0.1 % of the bytes
        assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3
    # the same frame without the EYE draw is byte-exact
    case["draws"] = case["draws"][:4]
    ofb, oz, ost = cases.run_oracle(case)
    fb, z, st, _ = cases.run_gpu(case)
    assert np.array_equal(fb, ofb) and np.array_equal(z.view(np.uint64), oz.view(np.uint64)) and st == ost


def test_more_draws_than_descriptors_between_flushes():
    """70 small trgl_draw calls from host memory without a flush in between: the 65th finds the 64 draw descriptors of a
    flush taken and flushes on its own (trgl_api.cpp, trgl_draw), while the staged host arrays of the draw in progress stay
    alive.  The frame equals the oracle's, which sees one rasterize() loop."""
    W, H = 200, 120
    clip, col = scenes.random_triangles(70 * 37, W, H, seed=81, rmin=2, rmax=30, perspective_w=True)
    with Context(W, H, 3) as ctx:
        for i in range(70):
            ctx.draw(FLAT, clip[37 * i: 37 * (i + 1)], colors=col[37 * i: 37 * (i + 1)])
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(fb, o.fb) and np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and st == o.stats


@pytest.mark.parametrize("log2n", [24, 25])
def test_one_draw_of_more_than_2_to_24_triangles(log2n):
    """A record addresses its triangle as (draw, 24-bit index), so trgl_draw splits a submission of 2^24 + 40 000 triangles into
    two descriptors whose clip / colour pointers are offset.  The last 40 000 triangles (the second descriptor) are the
    visible ones: wrong offsets would show as wrong colours or positions.  Host arrays (1.6 GB of clip data).
    2^25 + 40 000: k_raster addresses a record as base + (index << 7) with a 32-bit offset, so trgl_draw also starts a new FLUSH
    before the 2^25-th triangle of one (TRGL_FLUSH_MAX_TRIS); the frame must not show where."""
    W, H = 256, 256
    n_small, n_vis = (1 << log2n) + 3000, 37_000
    vis, vcol = scenes.random_triangles(n_vis + 3000, W, H, seed=91, rmin=2, rmax=30)
    clip = np.empty((n_small + n_vis, 12))
    col = (np.arange(n_small + n_vis, dtype=np.uint64) * np.uint64(2654435761) & np.uint64(0xFFFFFF)).astype(np.uint32) | np.uint32(0xFF000000)
    # the first 2^24 + 3000 triangles are the first 3000 of `vis` shrunk around their first vertex to 1/1000 of a pixel (almost never
    # cover a pixel centre, cheap for the oracle) and repeated; the rest are ordinary
    base = vis[:3000].copy()
    for v in (1, 2):
        base[:, 4 * v: 4 * v + 2] = base[:, 0:2] + (base[:, 4 * v: 4 * v + 2] - base[:, 0:2]) * 1e-3
    reps = -(-n_small // 3000)
    clip[:n_small] = np.tile(base, (reps, 1))[:n_small]
    clip[n_small:] = vis[3000:]
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert st == o.stats and st[0] == n_small + n_vis
    assert np.array_equal(fb, o.fb) and np.array_equal(z.view(np.uint64), o.z.view(np.uint64))


def test_pair_buffers_grow_behind_an_optimistic_launch():
    """The binning kernels are queued for the capacity the pair buffers already have (first flush: 2 pairs per triangle) and do
    nothing when the flush needs more; trgl_flush_end then grows the buffers and queues them again.  100 full-screen triangles
    at 512x512 are 25 600 pairs against a first guess of 4 296; a second, larger frame grows them once more."""
    W = H = 512
    clip, col = scenes.random_triangles(100, W, H, seed=95, rmin=600, rmax=1200)
    big, bcol = scenes.random_triangles(300, W, H, seed=96, rmin=600, rmax=1200)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
        assert ctx.last_flush_info()["pairs"] > 2 * 100 + 4096
        o = orc.Oracle(W, H, 3)
        o.draw(orc.FLAT, clip, colors=col)
        assert np.array_equal(fb, o.fb) and np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and st == o.stats
        ctx.draw(FLAT, big, colors=bcol)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
        o.draw(orc.FLAT, big, colors=bcol)
        assert np.array_equal(fb, o.fb) and np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and st == o.stats


def test_frames_beyond_65536_tiles_use_wide_keys():
    """The binning sorts 16-bit tile indices while the frame has at most 65536 tiles (8192x8192 is exactly that) and 32-bit
    ones beyond: 8224x8224 = 257x257 tiles, with triangles in the last tile rows (indices above 65535 would wrap)."""
    W = H = 8224
    n = 3000
    clip, col = scenes.random_triangles(n, W, H, seed=4242, rmin=4, rmax=60)
    clip = clip.copy()
    clip[: n // 2, [1, 5, 9]] = clip[: n // 2, [1, 5, 9]] * 0.02 + 0.975 * clip[: n // 2, [3, 7, 11]]     # half of them into the top rows (NDC y near +1)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert st == o.stats
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    ys = np.nonzero(np.isfinite(z).any(axis=1))[0]
    assert ys.max() >= 8192 or ys.min() < 32          # the frame's extreme tile rows were hit (whichever way y is flipped)


def test_more_than_2_to_32_pairs_is_refused_not_wrapped():
    """70 000 full-screen triangles at 8192x8192 are 4.6e9 (tile, triangle) pairs: the pair count is accumulated in 64 bits
    on the device (k_chunk_spine), every binning kernel sees it exceed the buffers and does nothing, and the flush returns
    TRGL_E_UNSUPPORTED instead of drawing from wrapped offsets.  The context stays usable."""
    from tinyrenderder_amd.api import TrglError
    W = H = 8192
    n = 70_000
    clip = np.empty((n, 12))
    clip[:] = [-3.0, -3.0, 0.0, 1.0, 3.0, -3.0, 0.0, 1.0, 0.0, 3.0, 0.0, 1.0]      # covers the whole screen, counter-clockwise
    col = np.full(n, 0xFF00FF00, np.uint32)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        with pytest.raises(TrglError):
            ctx.flush()
        small, scol = scenes.random_triangles(2000, W, H, seed=97, rmin=4, rmax=40)
        ctx.clear()
        ctx.draw(FLAT, small, colors=scol)
        fb = ctx.read_framebuffer()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, small, colors=scol)
    assert np.array_equal(fb, o.fb)


def test_strip_loop_two_contexts_interleaved_frames():
    """bench.py's N > 1 frame loop (shard.StripLoop: submit -> flush_begin -> wait for the previous gather -> flush_end -> start
    the gather, in place into an alias of the context's own framebuffer) driven by TWO strip contexts on one GPU over three
    different frames.  The collective is replaced by what it does to one rank's memory: the peer's strip lands in place, and
    it lands late - while the next frame's setup and binning are already queued - exactly when StripLoop waits for it.
    Checked after every frame: the rank's own rows hold the new frame, the rows it does not own still hold the peer's
    PREVIOUS frame (a pending clear must not touch them), and after the last gather both ranks hold the whole last frame."""
    import torch
    from tinyrenderder_amd import shard
    W, H, cut = 256, 192, 96
    frames = [scenes.random_triangles(6000, W, H, seed=300 + k, rmin=2, rmax=40, perspective_w=bool(k & 1)) for k in range(3)]
    want = []
    for clip, col in frames:
        o = orc.Oracle(W, H, 3, clear_bgra=(5 * len(want) + 1, 2, 3, 255))
        o.draw(orc.FLAT, clip, colors=col)
        want.append(o.fb.copy())
    stream = torch.cuda.current_stream().cuda_stream
    ctxs = [Context(W, H, 3), Context(W, H, 3)]
    rows = [(0, cut), (cut, H)]
    full = []
    for c, (y0, y1) in zip(ctxs, rows):
        c.set_stream(stream)
        c.set_strip(y0, y1)
        full.append(shard.framebuffer_tensor(c))

    class LateCopy:                       # the peer's strip, joined in place when the loop waits for the gather
        def __init__(self, me):
            self.me = me

        def wait(self):
            y0, y1 = rows[1 - self.me]
            full[self.me][y0 * W * 3: y1 * W * 3].copy_(full[1 - self.me][y0 * W * 3: y1 * W * 3])

    loops = [shard.StripLoop(ctxs[r], (lambda r=r: LateCopy(r))) for r in range(2)]
    try:
        for k, (clip, col) in enumerate(frames):
            dclip = torch.from_numpy(clip).cuda()
            dcol = torch.from_numpy(col.view(np.int32)).cuda()

            def submit(c, k=k, dclip=dclip, dcol=dcol):
                c.clear((5 * k + 1, 2, 3, 255))
                c.draw(FLAT, dclip, colors=dcol, device=True)
            for r in range(2):
                loops[r].step(submit)
            torch.cuda.synchronize()
            for r in range(2):
                got = full[r].cpu().numpy().reshape(H, W, 3)
                y0, y1 = rows[r]
                p0, p1 = rows[1 - r]
                assert np.array_equal(got[y0:y1], want[k][y0:y1]), f"frame {k} rank {r}: own rows"
                if k:       # what the late copy delivered: rank 0 stepped first and saw the peer's frame k-1, rank 1 saw rank 0's frame k
                    assert np.array_equal(got[p0:p1], want[k - 1 + r][p0:p1]), f"frame {k} rank {r}: rows of the peer were touched"
        for r in range(2):
            loops[r].finish()
        torch.cuda.synchronize()
        for r in range(2):
            assert np.array_equal(full[r].cpu().numpy().reshape(H, W, 3), want[-1]), f"rank {r}: gathered last frame"
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("i", range(5))
def test_device_sampler_equals_reference_sample2d(i):
    """The device samplers' texel fetch (tex_fetch in kernels_raster.hip, through trgl_selftest_sampler) against the
    fixtures taken from the reference's compiled IShader::sample2D + TGAImage::get: NaN / inf / negative / huge uv (the
    x86 cvttsd2si result INT_MIN clamps to texel 0), every texel boundary +- 1 ulp, 1 / 3 / 4 bytes per pixel."""
    g = np.load(os.path.join(HERE, "golden", "sampler_golden.npz"))
    tex, uv, want = g[f"tex{i}"], g[f"uv{i}"].view(np.float64), g[f"out{i}"]
    with Context(32, 32, 3) as ctx:
        ctx.upload_texture(3, tex)
        got = ctx.selftest_sampler(3, uv)
        empty = ctx.selftest_sampler(5, uv[:4])
    bad = np.argwhere((got != want).any(axis=1))
    assert bad.size == 0, f"{len(bad)} samples differ, first uv {uv[bad[0, 0]]}: {got[bad[0, 0]]} vs {want[bad[0, 0]]}"
    assert (empty == np.array([255, 255, 255, 255, 4], np.uint8)).all()


@pytest.mark.parametrize("name,world,band", [("flat_persp_512", 4, 32), ("flat_persp_512", 2, 128), ("phong_512", 4, 64), ("multi_draw_320x200", 1, 32)])
def test_interleaved_bands_compose(name, world, band):
    """trgl_set_interleave: bands of `band` rows dealt round-robin to `world` contexts (the load-balanced alternative to one
    strip per rank).  Every context's own bands equal the rows of the unsharded frame (colours and depths; PHONG goes through
    the visibility buffer + k_shade), the fragment counts add up and the z range is the min / max over ranks."""
    from tinyrenderder_amd import shard
    case = cases.CASES[name]()
    W, H = case["width"], case["height"]
    fb, z, st, _ = cases.run_gpu(case)
    frags, zmin, zmax = 0, np.inf, -np.inf
    for rank in range(world):
        with Context(W, H, case["bpp"]) as ctx:
            ctx.set_viewport(case["viewport"]); ctx.clear(case["clear"], case["zclear"])
            ctx.set_interleave(band, rank, world)
            for slot, t in case["textures"].items():
                ctx.upload_texture(slot, t)
            for kind, u, clip, vary, col in case["draws"]:
                ctx.draw(kind, clip, vary, col, u)
            rfb, rz, rst = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
        rows = shard.band_rows_of(H, world, rank, band) if world > 1 else [(0, H)]
        for y0, y1 in rows:
            if name in POW_CASES:
                _assert_fb(rfb[y0:y1], fb[y0:y1], name)
            else:
                assert np.array_equal(rfb[y0:y1], fb[y0:y1]), f"rank {rank}: colours of rows {y0}..{y1}"
            assert np.array_equal(rz[y0:y1].view(np.uint64), z[y0:y1].view(np.uint64)), f"rank {rank}: depths of rows {y0}..{y1}"
        assert rst[0] == st[0] and rst[2:6] == st[2:6]
        frags += rst[1]; zmin = min(zmin, rst[6]); zmax = max(zmax, rst[7])
    assert (frags, zmin, zmax) == (st[1], st[6], st[7])


@pytest.mark.parametrize("partition", ["strips", "bands"])
def test_bench_two_ranks_on_one_gpu_over_gloo(partition):
    """bench.py's N = 2 code path run for real as two processes (torch.distributed.run), both on this box's one GPU, with gloo
    as the transport because RCCL refuses two ranks on one device: strip / band contexts, shard.StripLoop with the asynchronous
    in-place gather into an alias of each context's framebuffer, the barrier + max-over-ranks timing, and the parity gate on
    rank 0's GATHERED framebuffer against the reference's full-size digest (4096^2, 10 M triangles).  The throughput it prints is
    meaningless (two processes share a GPU and the strips travel through host memory)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    port = 29600 + (os.getpid() % 300) + (7 if partition == "bands" else 0)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--partition", partition,
                        "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["parity"]["checked"] and d["parity"]["ok"], d["parity"]


@pytest.mark.parametrize("partition", ["strips", "bands"])
def test_c_abi_gather_single_rank_communicator(partition):
    """trgl_gather (the RCCL all-gather of the north star, behind the C ABI) with the communicator trgl_rccl_comm_create makes:
    on a one-GPU box that is a 1-rank communicator - the in-place all-gathers run through RCCL for real, over one strip /
    the bands of one rank - and the frame still equals the oracle's.  (Two ranks need two GPUs: tests/test_multi_rank_cpu.py
    covers the composition over gloo, examples/demo_multi.cpp is the C++ caller.)"""
    from tinyrenderder_amd import api
    case = cases.CASES["flat_persp_512"]()
    ofb, oz, ost = cases.run_oracle(case)
    comm = api.rccl_comm_create(api.rccl_unique_id(), 0, 1, 0)
    try:
        with api.Context(case["width"], case["height"], case["bpp"]) as ctx:
            ctx.set_viewport(case["viewport"]); ctx.clear(case["clear"], case["zclear"])
            if partition == "bands":
                ctx.set_interleave(64, 0, 1)
            for kind, u, clip, vary, col in case["draws"]:
                ctx.draw(kind, clip, vary, col, u)
            ctx.gather(comm, 0, 1, with_z=True)
            fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    finally:
        api.rccl_comm_destroy(comm)
    assert np.array_equal(z.view(np.uint64), oz.view(np.uint64)) and np.array_equal(fb, ofb) and st == ost


def test_interleaved_rank_without_rows_draws_nothing_and_stays_usable():
    """trgl_set_interleave with more ranks x band rows than the image has rows: a rank whose bands lie beyond the image owns no tile
    row.  Its flushes must complete (no launch over an empty grid, no stuck draw list), count nothing, and the context must keep
    working when it is given rows again."""
    W, H = 96, 64
    clip, col = scenes.random_triangles(500, W, H, seed=3, rmin=2, rmax=24)
    with Context(W, H, 3) as ctx:
        ctx.set_interleave(32, 3, 4)                 # bands of 32 rows, 4 ranks: ranks 2 and 3 own nothing of 64 rows
        ctx.draw(FLAT, clip, colors=col)
        st = ctx.stats()
        assert st[0] == 500 and st[1] == 0
        ctx.draw(FLAT, clip, colors=col)             # a second flush on the same context
        assert ctx.stats()[1] == 0
        ctx.set_strip(0, H)
        ctx.reset_stats(); ctx.clear()
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(fb, o.fb) and np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and st == o.stats


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_depth_bound_in_the_pair_on_large_triangles(seed):
    """A flush that holds triangles of 64 pixels and more lets k_raster's list steps drop entries by the 7-bit depth bound that rides in
    the pair's triangle word (TRGL_VAL_ZQ): 4000 triangles of 20 - 300 px on 320 x 200, depth complexity in the hundreds, so that
    nearly every later entry is occluded - and has to be dropped ONLY then.  A third of them get a vertex depth outside [-1, 1]
    (bound 0 = says nothing), every fifth is nearly flat in depth (bound as tight as the 1/64 grid allows), all on top of small
    triangles that share the tiles.  z bits, colours and counters must equal the oracle's."""
    W, H = 320, 200
    big, bcol = scenes.random_triangles(4000, W, H, seed=8100 + seed, rmin=20, rmax=300)
    small, scol = scenes.random_triangles(6000, W, H, seed=8200 + seed, rmin=1, rmax=12)
    rng = scenes.SplitMix64(8300 + seed)
    u = rng.uniform(4000 * 2).reshape(4000, 2)
    big = big.copy()
    for i in range(4000):
        if i % 3 == 0:
            big[i, 4 * (i % 3) + 2] = -1.0 - 3.0 * u[i, 0]           # one vertex in front of the near plane
        if i % 5 == 0:
            z0 = 2.0 * u[i, 1] - 1.0
            for v in range(3):
                big[i, 4 * v + 2] = z0 + 1e-4 * v
    clip = np.concatenate([small[:3000], big[:2000], small[3000:], big[2000:]])
    col = np.concatenate([scol[:3000], bcol[:2000], scol[3000:], bcol[2000:]])
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
        # the same scene in two flushes, the second starting from the depths of the first
        ctx.clear()
        ctx.draw(FLAT, clip[:5000], colors=col[:5000]); ctx.flush()
        ctx.draw(FLAT, clip[5000:], colors=col[5000:])
        fb2, z2 = ctx.read_framebuffer(), ctx.read_zbuffer()
    o = orc.Oracle(W, H, 3)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats
    assert np.array_equal(z2.view(np.uint64), o.z.view(np.uint64)) and np.array_equal(fb2, o.fb)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_fuzz_discarding_scenes_against_oracle(seed):
    """The kind that discards (CHECKER, our_gl.cpp:187-188) under the conditions of the flat fuzz: dense overdraw, perspective w (the
    predicate reads the PERSPECTIVE-CORRECT barycentrics), 1 - 40 cells, large and small triangles in one flush (the depth bound in
    the pair must not drop an entry whose fragments behind a discarded one still count), a flat draw underneath in the same flush
    on odd seeds, two flushes on the others.  A discarded fragment writes nothing and counts nowhere."""
    from tinyrenderder_amd.api import CHECKER, make_uniforms
    rng = scenes.SplitMix64(9000 + seed)
    u = rng.uniform(8)
    W = int(80 + u[0] * 240); H = int(80 + u[1] * 200)
    n = int(3000 + u[2] * 12000)
    cells = int(1 + u[3] * 40)
    clip, col = scenes.random_triangles(n, W, H, seed=9100 + seed, rmin=1 + 4 * u[4], rmax=10 + 150 * u[5], perspective_w=True)
    base, bcol = scenes.random_triangles(n // 3, W, H, seed=9200 + seed, rmin=3, rmax=60)
    uni = make_uniforms(cells=cells)
    with Context(W, H, 3) as ctx:
        if seed & 1:
            ctx.draw(FLAT, base, colors=bcol)
            ctx.draw(CHECKER, clip, colors=col, uniforms=uni)
        else:
            ctx.draw(CHECKER, clip[: n // 2], colors=col[: n // 2], uniforms=uni); ctx.flush()
            ctx.draw(CHECKER, clip[n // 2:], colors=col[n // 2:], uniforms=uni)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    o = orc.Oracle(W, H, 3)
    if seed & 1:
        o.draw(orc.FLAT, base, colors=bcol)
    o.draw(orc.CHECKER, clip, colors=col, uniforms=orc.make_uniforms(cells=cells))
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    assert np.array_equal(fb, o.fb)
    assert st == o.stats
