"""CPU tests of the oracle (oracle/trgl_oracle.c) against the golden vectors in tests/golden/, which were
produced by the reference's own rasterize() (tests/golden/make_golden.py).  No GPU, no reference tree needed."""
import json
import os

import numpy as np
import pytest

import cases
from oracle import orc
from tinyrenderder_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def _input_digest(case):
    parts = [np.asarray(case["viewport"], np.float64)]
    for kind, u, clip, vary, col in case["draws"]:
        parts.append(np.frombuffer(bytes(u), np.uint8) if u is not None else np.zeros(1, np.uint8))
        parts += [clip] + ([vary] if vary is not None else []) + ([col] if col is not None else [])
    for slot in sorted(case["textures"]):
        parts.append(case["textures"][slot])
    return scenes.digest(np.concatenate([np.ascontiguousarray(p).view(np.uint8).ravel() for p in parts]))


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_matches_reference_golden(name):
    case = cases.CASES[name]()
    g = GOLDEN[name]
    assert _input_digest(case) == g["inputs"], "scene generator drifted: inputs differ from the ones the reference saw"
    fb, z, st = cases.run_oracle(case)
    assert orc.format_stats_line(st) == g["stats"]
    assert scenes.digest(z) == g["z"], "z-buffer bits differ from the reference"
    assert scenes.digest(fb) == g["fb"], "framebuffer bytes differ from the reference"


@pytest.mark.parametrize("name", cases.FULL_BUFFER_CASES)
def test_oracle_full_buffers(name):
    ref = np.load(os.path.join(HERE, "golden", name + ".npz"))
    fb, z, _ = cases.run_oracle(cases.CASES[name]())
    assert np.array_equal(fb, ref["fb"])
    assert np.array_equal(z.view(np.uint64), ref["z"].view(np.uint64))


def test_oracle_strips_compose():
    """Rows are independent: rendering two strips separately equals the whole (basis of the multi-GPU shard)."""
    case = cases.CASES["flat_persp_512"]()
    fb, z, st = cases.run_oracle(case)
    h = case["height"]
    cut = 200                                  # not a multiple of the tile size on purpose
    fb0, z0, s0 = cases.run_oracle(case, strip=(0, cut))
    fb1, z1, s1 = cases.run_oracle(case, strip=(cut, h))
    assert np.array_equal(np.concatenate([fb0[:cut], fb1[cut:]]), fb)
    assert np.array_equal(np.concatenate([z0[:cut], z1[cut:]]).view(np.uint64), z.view(np.uint64))
    assert s0[1] + s1[1] == st[1]                                  # fragments_drawn adds up
    assert min(s0[6], s1[6]) == st[6] and max(s0[7], s1[7]) == st[7]
    assert s0[2:6] == st[2:6] and s1[2:6] == st[2:6]               # bbox stats come from triangles, not pixels


def test_x86_cast_edge_cases_are_in_the_golden_scene():
    """edge_256 holds a vertex whose screen coordinate overflows int: the reference (cvttsd2si) drops it."""
    clip, _ = scenes.edge_case_triangles(256, 256)
    assert abs(clip[25, 0] / clip[25, 3]) * 128 > 2 ** 31


@pytest.mark.ref
@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (reference tree absent)")
def test_value_ops_match_reference_geometry_h(tmp_path):
    """normalized(), mat*vec4, varying interpolation and TGAColor*float of the restatement vs the real
    geometry.h / tgaimage.h (through oracle/_ref/ref_harness vecops)."""
    import ctypes as C
    import struct
    import subprocess
    rng = scenes.SplitMix64(77)
    n = 2000
    data = rng.uniform(n * 35, -3.0, 3.0).reshape(n, 35)
    data[0, 0:3] = 0.0                                    # normalized(0) returns v
    data[:, 34] = rng.uniform(n, -0.5, 1.5)
    packed = (rng.u64(n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    p_in, p_out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(p_in, "wb") as f:
        f.write(struct.pack("<ii", n, 0))
        for i in range(n):
            f.write(data[i].tobytes()); f.write(struct.pack("<II", int(packed[i]), 0))
    subprocess.run([orc.REF_HARNESS, "vecops", str(p_in), str(p_out)], check=True)
    raw = np.frombuffer(open(p_out, "rb").read(), np.uint8).reshape(n, 80)
    ref_d = raw[:, :72].copy().view(np.float64).reshape(n, 9)
    ref_c = raw[:, 72:76]
    L = orc.lib()
    for i in range(n):
        v, nn, M = data[i, 0:3].copy(), data[i, 3:6].copy(), data[i, 6:22].copy()
        v0, v1, v2, b = (data[i, 22 + 3 * k: 25 + 3 * k].copy() for k in range(4))
        o = np.empty(9)
        L.orc_normalized3(v.ctypes.data, o[0:3].ctypes.data)
        tmp = np.empty(3); L.orc_mat4_mul_dir(M.ctypes.data, nn.ctypes.data, tmp.ctypes.data); o[3:6] = tmp
        tmp2 = np.empty(3); L.orc_interp(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, b.ctypes.data, 3, tmp2.ctypes.data); o[6:9] = tmp2
        assert np.array_equal(o.view(np.uint64), ref_d[i].view(np.uint64)), i
        out = (C.c_uint8 * 4)()
        inten = np.array([data[i, 34]] * 3)
        bary = (C.c_double * 3)(1.0, 0.0, 0.0)
        L.orc_fragment(orc.GOURAUD, None, None, inten.ctypes.data, int(packed[i]), bary, out)
        assert bytes(out) == bytes(ref_c[i]), i


@pytest.mark.ref
@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (reference tree absent)")
def test_golden_is_reproducible_from_reference():
    """Re-run the reference on one case and compare with the committed fixture."""
    case = cases.CASES["flat_persp_512"]()
    fb, z, line = orc.run_reference(case["width"], case["height"], case["bpp"], case["viewport"],
                                    [(k, None, cl, v, co) for k, u, cl, v, co in case["draws"]])
    g = GOLDEN["flat_persp_512"]
    assert (scenes.digest(fb), scenes.digest(z), line) == (g["fb"], g["z"], g["stats"])


# ---- samplers: index math pinned by the reference's compiled IShader::sample2D (our_gl.h:38-44) ----------------------------------
SAMPLER_GOLDEN = np.load(os.path.join(HERE, "golden", "sampler_golden.npz"))


@pytest.mark.parametrize("i", range(5))
def test_sampler_restatement_equals_reference_sample2d(i):
    """orc tex_fetch (model.cpp:415-425 restated) vs the fixtures tests/golden/make_sampler_golden.py took from the compiled
    reference: NaN, +-inf, negatives, exactly 1.0, 1e300, values beyond INT_MAX after scaling, every texel boundary +- 1 ulp,
    textures of 1 / 3 / 4 bytes per pixel, 1x1 and odd sizes."""
    tex, uv, want = SAMPLER_GOLDEN[f"tex{i}"], SAMPLER_GOLDEN[f"uv{i}"].view(np.float64), SAMPLER_GOLDEN[f"out{i}"]
    got = orc.tex_fetch(tex, uv)
    bad = np.argwhere((got != want).any(axis=1))
    assert bad.size == 0, f"{len(bad)} samples differ, first uv {uv[bad[0, 0]]}"


@pytest.mark.ref
@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (reference tree absent)")
def test_sampler_fixture_is_what_the_reference_returns_now():
    tex, uv, want = SAMPLER_GOLDEN["tex1"], SAMPLER_GOLDEN["uv1"].view(np.float64), SAMPLER_GOLDEN["out1"]
    assert np.array_equal(orc.run_reference_sample2d(tex, uv), want)


@pytest.mark.ref
@pytest.mark.skipif(not os.path.exists("/root/reference/geometry.h"), reason="reference tree absent")
def test_shim_compiles_against_the_reference_headers():
    """a maintainer keeps the reference's own geometry.h / tgaimage.h and includes the shim with
    TRGL_GEOMETRY_HEADER / TRGL_IMAGE_HEADER.  tests/host/shim_against_reference_headers.cpp is such a caller (face loop,
    zbuffer copy / restore through the proxy, gl_draw_model, gl_flush); it must compile with the reference's headers where
    they lie (syntax check only: nothing of the reference is copied or linked)."""
    import subprocess
    root = os.path.dirname(HERE)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I/root/reference", "-I" + os.path.join(root, "tinyrenderder_amd", "shim"),
                        os.path.join(HERE, "host", "shim_against_reference_headers.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
