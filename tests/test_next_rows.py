"""SURVEY.md §8(f) rows built after the hot path: N1 (vertex stage on the device) and N4 (z-buffer image, SSAO,
composite on the resident buffers).  CPU part: the C restatement against an independent numpy evaluation;
GPU part: the HIP kernels against the restatement, bit / byte exact."""
import numpy as np
import pytest

import cases
from oracle import orc
from tinyrenderder_amd import scenes


def _indexed_head(level, w, h):
    hd = scenes.head_standin(level, w, h)
    pos, nrm, uv = hd["positions"].reshape(-1, 3), hd["normals"].reshape(-1, 3), hd["uvs"].reshape(-1, 2)
    # the reference's Vertex (model.h:14-20): position, normal, texcoord, tangent, bitangent = 14 doubles; share
    # vertices between faces through a shuffled index buffer so the gather is exercised
    verts = np.concatenate([pos, nrm, uv, np.zeros((pos.shape[0], 6))], 1)
    perm = np.argsort(scenes.SplitMix64(9).u64(pos.shape[0]), kind="stable")
    inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
    return hd, np.ascontiguousarray(verts[perm]), inv.astype(np.uint32).reshape(-1, 3)


def test_oracle_vertex_stage_matches_numpy_vertex_stage():
    hd, verts, idx = _indexed_head(3, 320, 200)
    clip, vary = orc.vertex_stage(hd["model_view"], hd["projection"], verts, idx)
    assert np.array_equal(clip.view(np.uint64), hd["clip"].view(np.uint64))
    assert np.array_equal(vary.view(np.uint64), hd["varyings"].view(np.uint64))


def test_oracle_postprocess_against_numpy():
    """Independent numpy evaluation of main.cpp:269-311 and 768-783 (the SSAO loop is checked on the GPU side)."""
    fb, z, _ = cases.run_oracle(cases.CASES["multi_draw_320x200"]())
    fin = np.isfinite(z)
    lo, hi = min(1e9, z[fin].min()), max(-1e9, z[fin].max())
    if hi - lo < 1e-7:
        hi = lo + 1e-7
    val = np.full(z.shape, 255, np.uint8)
    val[fin] = (255.0 * (1.0 - (z[fin] - lo) / (hi - lo))).astype(np.uint8)
    assert np.array_equal(orc.zbuffer_image(z), np.repeat(val[..., None], 3, -1))
    ao = orc.ssao(z)
    expect = np.minimum(255.0, fb.astype(np.float64) * (ao[..., :1] / 255.0)).astype(np.uint8)
    assert np.array_equal(orc.composite(fb, ao), expect)
    assert ao.min() < 255 and ao.max() == 255          # some occlusion, some open sky


@pytest.mark.gpu
def test_gpu_draw_indexed_equals_host_vertex_stage():
    from tinyrenderder_amd.api import Context, PHONG, EYE, make_uniforms
    W, H = 640, 480
    hd, verts, idx = _indexed_head(5, W, H)
    d, n, s = scenes.procedural_textures(256)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 0.8, 0, 1, 2)
    res = []
    for indexed in (False, True):
        with Context(W, H, 3) as ctx:
            for k, t in enumerate((d, n, s)):
                ctx.upload_texture(k, t)
            if indexed:
                ctx.draw_indexed(PHONG, u, hd["projection"], verts, idx)
                ctx.draw_indexed(EYE, u, hd["projection"], verts, idx[::4])
            else:
                ctx.draw(PHONG, hd["clip"], hd["varyings"], uniforms=u)
                ctx.draw(EYE, hd["clip"][::4], hd["varyings"][::4], uniforms=u)
            res.append((ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()))
    assert np.array_equal(res[0][1].view(np.uint64), res[1][1].view(np.uint64))
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][2] == res[1][2]
    o = orc.Oracle(W, H, 3)
    for k, t in enumerate((d, n, s)):
        o.upload_texture(k, t)
    ou = orc.make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 0.8, 0, 1, 2)
    o.draw(orc.PHONG, hd["clip"], hd["varyings"], uniforms=ou)
    o.draw(orc.EYE, hd["clip"][::4], hd["varyings"][::4], uniforms=ou)
    assert np.array_equal(res[1][1].view(np.uint64), o.z.view(np.uint64)) and res[1][2] == o.stats


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["multi_draw_320x200", "flat_persp_512", "odd_dims_101x67", "empty_scene_64"])
def test_gpu_postprocess_matches_restatement(name):
    from tinyrenderder_amd.api import Context
    case = cases.CASES[name]()
    if case["bpp"] != 3:
        pytest.skip("composite is defined on the RGB framebuffer")
    with Context(case["width"], case["height"], 3) as ctx:
        ctx.set_viewport(case["viewport"]); ctx.clear(case["clear"], case["zclear"])
        for slot, t in case["textures"].items():
            ctx.upload_texture(slot, t)
        for kind, u, clip, vary, col in case["draws"]:
            ctx.draw(kind, clip, vary, col, u)
        fb, z = ctx.read_framebuffer(), ctx.read_zbuffer()
        out = ctx.postprocess()
    assert np.array_equal(out["zbuffer_image"], orc.zbuffer_image(z))
    ao = orc.ssao(z)
    assert np.array_equal(out["ao"], ao)
    assert np.array_equal(out["final"], orc.composite(fb, ao))


@pytest.mark.gpu
def test_gpu_postprocess_odd_size():
    """W * H not a multiple of 4 (33 x 31): the three output images sit at 16-byte boundaries of the device buffer, so the dword
    stores of the z-image and composite kernels stay aligned whatever the size."""
    from tinyrenderder_amd.api import Context, FLAT
    W, H = 33, 31
    clip, col = scenes.random_triangles(400, W, H, seed=77, rmin=2, rmax=16)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z = ctx.read_framebuffer(), ctx.read_zbuffer()
        out = ctx.postprocess()
    ao = orc.ssao(z)
    assert np.array_equal(out["zbuffer_image"], orc.zbuffer_image(z))
    assert np.array_equal(out["ao"], ao) and np.array_equal(out["final"], orc.composite(fb, ao))


@pytest.mark.gpu
def test_gpu_postprocess_4096():
    """Full-size post-process on the C4 frame (1 M-triangle prefix): SSAO reads 64 depths per pixel from LDS tiles."""
    from tinyrenderder_amd.api import Context, FLAT
    W = H = 4096
    clip, col = scenes.random_triangles(1_000_000, W, H)
    with Context(W, H, 3) as ctx:
        ctx.draw(FLAT, clip, colors=col)
        fb, z = ctx.read_framebuffer(), ctx.read_zbuffer()
        out = ctx.postprocess()
    rows = slice(1000, 1200)                                   # the scalar restatement needs ~1 us per sample: check a band
    ao_band = orc.ssao(z)[rows] if False else None
    zi = orc.zbuffer_image(z)
    assert np.array_equal(out["zbuffer_image"], zi)
    # SSAO of a band, computed on a padded crop (radius 16) so the crop's interior equals the full image's
    y0, y1 = 984, 1216
    ao_crop = orc.ssao(np.ascontiguousarray(z[y0:y1]))
    assert np.array_equal(out["ao"][1000:1200], ao_crop[16:216])
    assert np.array_equal(out["final"], orc.composite(fb, out["ao"]))


def _write_obj(path, level=2, with_normals=True):
    """An OBJ of the icosphere with shared positions, quads mixed in, negative indices and v/vt/vn forms."""
    tris = scenes.icosphere(level)                                   # [F,3,3]
    pts = tris.reshape(-1, 3)
    uniq, inv = np.unique(np.round(pts, 12), axis=0, return_inverse=True)
    inv = inv.reshape(-1, 3)
    uv = np.stack([uniq[:, 0] * 0.25 + 0.5, uniq[:, 1] * 0.25 + 0.5], 1)
    with open(path, "w") as f:
        f.write("# generated\n")
        for p in uniq: f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for t in uv: f.write("vt %.9g %.9g\n" % tuple(t))
        if with_normals:
            for p in uniq: f.write("vn %.9g %.9g %.9g\n" % tuple(p))
        n = len(uniq)
        for k, (a, b, c) in enumerate(inv):
            if with_normals:
                if k % 3 == 0:
                    f.write(f"f {a+1}/{a+1}/{a+1} {b+1}/{b+1}/{b+1} {c+1}/{c+1}/{c+1}\n")
                elif k % 3 == 1:
                    f.write(f"f {a-n}/{a-n}/{a-n} {b-n}/{b-n}/{b-n} {c-n}/{c-n}/{c-n}\n")      # relative indices
                else:
                    f.write(f"f {a+1}/{a+1}/{a+1} {b+1}/{b+1}/{b+1} {c+1}/{c+1}/{c+1} {c+1}/{c+1}/{c+1}\n")   # a (degenerate) quad: 2 fan triangles
            else:
                f.write(f"f {a+1}/{a+1} {b+1}/{b+1} {c+1}/{c+1}\n")
    return uniq, uv, inv


def test_obj_loader_layout_flipuv_fan_and_normals(tmp_path):
    from tinyrenderder_amd import api
    path = str(tmp_path / "ico.obj")
    uniq, uv, inv = _write_obj(path, 2, with_normals=True)
    verts, idx = api.load_obj(path)
    f32 = lambda a: np.asarray(a, np.float64).astype(np.float32)
    # one vertex per distinct triple, in order of first use
    order = []
    for a in inv.reshape(-1):
        if a not in order: order.append(a)
    assert verts.shape == (len(order), 14)
    assert np.array_equal(verts[:, 0:3], f32(np.array([[float("%.9g" % c) for c in uniq[a]] for a in order])).astype(np.float64))
    assert np.array_equal(verts[:, 7], (np.float32(1.0) - f32([float("%.9g" % uv[a, 1]) for a in order])).astype(np.float64))   # FlipUVs in float
    assert np.all(verts[:, 8:] == 0)
    nquads = len(range(2, len(inv), 3))
    assert idx.shape[0] == len(inv) + nquads                         # each quad became two fan triangles
    remap = {a: k for k, a in enumerate(order)}
    assert np.array_equal(idx[0], [remap[a] for a in inv[0]])
    # without vn: the reference's own fallback, model.cpp:269-316 (area-weighted smooth normals, normalised)
    path2 = str(tmp_path / "ico_nonormals.obj")
    _write_obj(path2, 2, with_normals=False)
    v2, i2 = api.load_obj(path2)
    ln = np.sqrt((v2[:, 3:6] ** 2).sum(1))
    assert np.allclose(ln, 1.0, atol=1e-12)
    assert (np.einsum("ij,ij->i", v2[:, 3:6], v2[:, 0:3]) > 0.9).all()      # on a sphere they point outwards
    with pytest.raises(api.TrglError):
        api.load_obj(str(tmp_path / "missing.obj"))


@pytest.mark.gpu
def test_obj_to_screen_through_device_vertex_stage(tmp_path):
    """OBJ file -> trgl_obj_load -> trgl_draw_indexed (vertex stage + PHONG on the GPU) vs the oracle on the same arrays."""
    from tinyrenderder_amd import api
    from tinyrenderder_amd.api import Context, PHONG, make_uniforms
    path = str(tmp_path / "ico.obj")
    _write_obj(path, 4, with_normals=True)
    verts, idx = api.load_obj(path)
    W, H = 512, 384
    hd = scenes.head_standin(1, W, H)                                # only for its camera and lights
    d, n, s = scenes.procedural_textures(128)
    u = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2)
    with Context(W, H, 3) as ctx:
        for k, t in enumerate((d, n, s)):
            ctx.upload_texture(k, t)
        ctx.draw_indexed(PHONG, u, hd["projection"], verts, idx)
        fb, z, st = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats()
    clip, vary = orc.vertex_stage(hd["model_view"], hd["projection"], verts, idx)
    o = orc.Oracle(W, H, 3)
    for k, t in enumerate((d, n, s)):
        o.upload_texture(k, t)
    o.draw(orc.PHONG, clip, vary, uniforms=orc.make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, 0, 1, 2))
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and np.array_equal(fb, o.fb) and st == o.stats
    assert st[1] > 10_000


def _c_round(v):
    """C round(): half away from zero, exactly (np.rint is half-to-even; floor(v + 0.5) is wrong just below a half)."""
    r = np.trunc(v)
    return np.where(np.abs(v - r) >= 0.5, r + np.sign(v), r)


def _numpy_ssao(z, ndir, steps, radius, threshold, intensity):
    """main.cpp:317-362,757-763 for arbitrary parameters, vectorised over the pixels (fp64, the reference's operation order)."""
    import math
    h, w = z.shape
    ys, xs = np.mgrid[0:h, 0:w]
    occluded = np.zeros((h, w), np.int64); total = np.zeros((h, w), np.int64)
    live = np.isfinite(z)
    for d in range(ndir):
        angle = 2.0 * 3.14159265358979323846 * d / ndir
        dx, dy = math.cos(angle), math.sin(angle)
        for step in range(1, steps + 1):
            r = float(step) / steps * radius
            sx = _c_round(xs + dx * r).astype(np.int64); sy = _c_round(ys + dy * r).astype(np.int64)
            inside = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h) & live
            sd = z[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]
            fin = np.isfinite(sd)
            with np.errstate(invalid="ignore"):
                occ = inside & fin & (sd < z - threshold)
            total += inside; occluded += occ
    ao = np.ones((h, w))
    m = live & (total != 0)
    ao[m] = 1.0 - (occluded[m].astype(np.float64) / total[m].astype(np.float64)) * intensity
    val = (255.0 * ao).astype(np.uint8)
    return np.repeat(val[..., None], 3, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("params", [(8, 8, 16.0, 1e-3, 0.35), (5, 3, 7.5, 0.02, 0.8), (16, 20, 12.0, 1e-3, 0.5), (7, 6, 40.0, 1e-3, 0.35),
                                    (3, 1, 0.4, 0.0, 1.0)])
def test_gpu_ssao_paths_against_numpy(params):
    """k_ssao takes integer sample offsets when they are exact for a block, counts interior blocks without range tests, leaves
    background blocks early, reads the z-buffer directly for radii beyond its 16-pixel halo and falls back to the literal
    arithmetic for more than 256 samples: every path against a plain numpy evaluation, on a z-buffer with NaN, -inf, +inf,
    an empty region (background blocks), image edges that cut blocks (150 x 97) and exact ties at the threshold."""
    from tinyrenderder_amd.api import Context, SsaoParams
    ndir, steps, radius, thr, inten = params
    W, H = 150, 97
    rng = scenes.SplitMix64(4100 + ndir * 31 + steps)
    z = rng.uniform(W * H, -1.0, 1.0).reshape(H, W)
    z = np.round(z * 64) / 64                                   # few distinct depths: differences exactly at the threshold occur
    z[:, 100:] = np.inf                                         # background (whole blocks without a finite depth)
    z[10:14, 20:40] = np.nan; z[30:33, 5:60] = -np.inf; z[50:70:3, 0:90:4] = np.inf
    sp = SsaoParams(ndir, steps, radius, thr, inten)
    with Context(W, H, 3) as ctx:
        ctx.write_zbuffer(z)
        got = ctx.postprocess(zbuffer_image=False, ao=True, final=False, params=sp)["ao"]
    want = _numpy_ssao(z, ndir, steps, radius, thr, inten)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{len(bad)} differing bytes, first at {bad[:5].tolist()}"
    if params == (8, 8, 16.0, 1e-3, 0.35):
        assert np.array_equal(got, orc.ssao(z))                 # the reference's own parameters: the C restatement agrees too
