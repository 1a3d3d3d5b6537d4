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
