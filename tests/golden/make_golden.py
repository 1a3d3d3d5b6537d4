"""Generate tests/golden/golden.json (+ small full buffers) by running the REFERENCE ITSELF.

Runs oracle/_ref/ref_harness — the reference's own our_gl.cpp + tgaimage.cpp compiled in place from
/root/reference (see oracle/Makefile) — on every scene in tests/cases.py and records, per case:
sha256 of the inputs, sha256 of the framebuffer bytes and of the z-buffer bit patterns, and the
print_render_stats() line.  Only runs in the build container (the reference tree is not on the GPU
box); the committed outputs are data, not reference source.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from oracle import orc  # noqa: E402
from tinyrenderder_amd import scenes  # noqa: E402


def input_digest(case):
    parts = [np.asarray(case["viewport"], np.float64)]
    for kind, u, clip, vary, col in case["draws"]:
        parts.append(np.frombuffer(bytes(u), np.uint8) if u is not None else np.zeros(1, np.uint8))
        parts += [clip] + ([vary] if vary is not None else []) + ([col] if col is not None else [])
    for slot in sorted(case["textures"]):
        parts.append(case["textures"][slot])
    return scenes.digest(np.concatenate([np.ascontiguousarray(p).view(np.uint8).ravel() for p in parts]))


def main():
    assert orc.ref_available(), "oracle/_ref/ref_harness missing: run `make -C oracle` where /root/reference exists"
    out = {}
    for name, build in cases.CASES.items():
        c = build()
        draws = [(k, None if u is None else orc.Uniforms.from_buffer_copy(bytes(u)), cl, v, co) for k, u, cl, v, co in c["draws"]]
        fb, z, line = orc.run_reference(c["width"], c["height"], c["bpp"], c["viewport"], draws, c["textures"], c["clear"], c["zclear"])
        out[name] = dict(inputs=input_digest(c), fb=scenes.digest(fb), z=scenes.digest(z), stats=line,
                         width=c["width"], height=c["height"], bpp=c["bpp"])
        if name in cases.FULL_BUFFER_CASES:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), fb=fb, z=z)
        print(name, line)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
