"""The main.cpp-shaped C++ demo (examples/demo_main.cpp) over the shim headers, against the CPU oracle replaying
the same sequence of passes (head PHONG, z-buffer save, EYE pass, z-buffer restore, flat overlay)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import orc
from tinyrenderder_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "examples", "demo_main")


def _write_model(path, W, H, bpp, hd, strength, textures, fclip, fcol):
    with open(path, "wb") as f:
        nf = hd["positions"].shape[0]
        f.write(b"TRGMDL01")
        f.write(struct.pack("<4i", W, H, bpp, nf))
        f.write(np.asarray(hd["model_view"], np.float64).tobytes())
        f.write(np.asarray(hd["projection"], np.float64).tobytes())
        for k in ("key", "fill", "rim"):
            f.write(np.asarray(hd["world_lights"][k], np.float64).tobytes())
        f.write(struct.pack("<d", strength))
        for k in ("positions", "normals", "uvs"):
            f.write(np.ascontiguousarray(hd[k], np.float64).tobytes())
        for t in textures:
            t = np.ascontiguousarray(t, np.uint8)
            f.write(struct.pack("<4i", t.shape[1], t.shape[0], t.shape[2], 0))
            b = t.tobytes()
            f.write(b + b"\0" * ((8 - len(b) % 8) % 8))
        f.write(struct.pack("<ii", fclip.shape[0], 0))
        f.write(fclip.tobytes())
        f.write(fcol.tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["model", "loop"])
def test_main_cpp_shaped_demo_matches_oracle(tmp_path, mode):
    """mode "model": the head pass is ONE gl_draw_model() call (vertex stage on the device, trgl_draw_indexed); mode "loop": the
    reference's per-face vertex() + rasterize() loop.  In both, `zbuffer` is read and assigned exactly as main.cpp:700,730 do
    (the proxy completes the batched eye triangles BEFORE the saved depths replace them: the eyes must show in the colours
    and not in the depths)."""
    assert os.path.exists(DEMO), "examples/demo_main not built: run __graft_entry__.build()"
    W, H, bpp = 320, 240, 3
    hd = scenes.head_standin(3, W, H)
    d, n, s = scenes.procedural_textures(128)
    fclip, fcol = scenes.random_triangles(300, W, H, seed=8, rmin=2, rmax=20)
    model, out, tga = tmp_path / "model.bin", tmp_path / "out.bin", tmp_path / "out.tga"
    _write_model(model, W, H, bpp, hd, 0.75, (d, n, s), fclip, fcol)
    post = str(tmp_path / "post")
    r = subprocess.run([DEMO, str(model), str(out), str(tga), post] + (["loop"] if mode == "loop" else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    fb = np.frombuffer(raw, np.uint8, W * H * bpp).reshape(H, W, bpp)
    z = np.frombuffer(raw, np.float64, W * H, W * H * bpp).reshape(H, W)
    line = raw[W * H * bpp + W * H * 8:].decode().strip()

    o = orc.Oracle(W, H, bpp)
    for slot, t in enumerate((d, n, s)):
        o.upload_texture(slot, t)
    args = (hd["model_view"], hd["key"], hd["fill"], hd["rim"])
    o.draw(orc.PHONG, hd["clip"], hd["varyings"], uniforms=orc.make_uniforms(*args, 0.75, 0, 1, 2))
    z_before = o.z.copy()
    o.draw(orc.EYE, hd["clip"][::3], hd["varyings"][::3], uniforms=orc.make_uniforms(*args, 1.0, 0, -1, 2))
    o.z[...] = z_before                                   # main.cpp:730
    o.draw(orc.FLAT, fclip, colors=fcol)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64))
    d8 = np.abs(fb.astype(np.int16) - o.fb.astype(np.int16))
    assert d8.max() <= 1 and (d8.max(axis=-1) > 0).mean() <= 1e-3      # EYE pass: pow tolerance (see test_gpu_parity)
    assert line == orc.format_stats_line(o.stats)
    assert r.stderr.strip().endswith(line)                # print_render_stats() wrote the same line to stderr
    assert open(tga, "rb").read() == orc.tga_encode(fb)   # the written file is what the reference writer would emit
    # main.cpp:751-785: zbuffer.tga / ao.tga / final.tga from the device post-process, byte for byte
    ao = orc.ssao(z)
    assert open(post + "_zbuffer.tga", "rb").read() == orc.tga_encode(orc.zbuffer_image(z))
    assert open(post + "_ao.tga", "rb").read() == orc.tga_encode(ao)
    assert open(post + "_final.tga", "rb").read() == orc.tga_encode(orc.composite(fb, ao))


@pytest.mark.gpu
def test_shim_reports_c_abi_errors_instead_of_aborting():
    """examples/shim_errors.cpp: a flush beyond 2^32 triangle-tile pairs comes back as gl_flush() == false with
    gl_last_error() == TRGL_E_UNSUPPORTED; the process lives and the next frame is drawn."""
    exe = os.path.join(ROOT, "examples", "shim_errors")
    assert os.path.exists(exe), "examples/shim_errors not built: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "flush 1: failed, code -5" in r.stdout and "2^32" in r.stdout
    assert "flush 2: ok, centre pixel 200 100 50" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strips", "bands"])
def test_cpp_multi_gpu_host_single_rank(tmp_path, mode):
    """examples/demo_multi.cpp - one process per GPU over the C ABI, RCCL bootstrap and trgl_gather included - run as a
    world of ONE on the one-GPU box: the gathered frame and this rank's stats equal the oracle's."""
    exe = os.path.join(ROOT, "examples", "demo_multi")
    assert os.path.exists(exe), "examples/demo_multi not built: run __graft_entry__.build()"
    W, H, bpp = 256, 192, 3
    clip, col = scenes.random_triangles(3000, W, H, seed=41, rmin=2, rmax=40, perspective_w=True)
    scene, out = tmp_path / "scene.bin", tmp_path / "out.bin"
    with open(scene, "wb") as f:
        f.write(struct.pack("<4i", W, H, bpp, clip.shape[0])); f.write(clip.tobytes()); f.write(col.tobytes())
    r = subprocess.run([exe, str(scene), str(out), "0", "1", str(tmp_path / "id")] + (["bands"] if mode == "bands" else []),
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, TRGL_DEVICE="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(str(out) + ".0", "rb").read()
    fb = np.frombuffer(raw, np.uint8, W * H * bpp).reshape(H, W, bpp)
    z = np.frombuffer(raw, np.float64, W * H, W * H * bpp).reshape(H, W)
    o = orc.Oracle(W, H, bpp)
    o.draw(orc.FLAT, clip, colors=col)
    assert np.array_equal(z.view(np.uint64), o.z.view(np.uint64)) and np.array_equal(fb, o.fb)
    assert raw[W * H * bpp + W * H * 8:].decode().strip() == orc.format_stats_line(o.stats)
