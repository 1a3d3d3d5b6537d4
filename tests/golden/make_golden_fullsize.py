"""Generate tests/golden/golden_fullsize.json: BASELINE configs[3] and [4] at their stated sizes, rendered by the
REFERENCE ITSELF (oracle/_ref/ref_harness_fast = the reference's own our_gl.cpp + tgaimage.cpp compiled in place,
-O3 -DNDEBUG -ffp-contract=off; see oracle/Makefile).  Records sha256 of the framebuffer bytes and of the z-buffer
bit patterns plus the print_render_stats() line.  Build container only; ~15 s + ~1.5 min of CPU.

    python tests/golden/make_golden_fullsize.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from make_golden import input_digest  # noqa: E402
from oracle import orc  # noqa: E402
from tinyrenderder_amd import scenes  # noqa: E402


def main():
    assert os.path.exists(orc.REF_HARNESS_FAST), "oracle/_ref/ref_harness_fast missing: run `make -C oracle` where /root/reference exists"
    out = {}
    for name, build in cases.FULLSIZE_CASES.items():
        c = build()
        fb, z, line, secs = orc.run_reference(c["width"], c["height"], c["bpp"], c["viewport"], c["draws"], c["textures"], c["clear"],
                                              c["zclear"], harness=orc.REF_HARNESS_FAST, with_time=True)
        out[name] = dict(inputs=input_digest(c), fb=scenes.digest(fb), z=scenes.digest(z), stats=line,
                         width=c["width"], height=c["height"], bpp=c["bpp"], reference_rasterize_seconds=round(secs, 2))
        print(name, line, f"{secs:.1f} s", flush=True)
    with open(os.path.join(HERE, "golden_fullsize.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
