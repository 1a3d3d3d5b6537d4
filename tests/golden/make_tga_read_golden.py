"""Generate tests/golden/tga_read/*.tga and tga_read_golden.json by running the REFERENCE's TGAImage::read_tga_file
(oracle/_ref/ref_harness tgaread = the reference's own tgaimage.cpp compiled in place) on small TGA files: files its
own writer produces, plus hand-made headers, origins, truncations and malformed streams.  Build container only; the
committed .tga files and digests are data.

    python tests/golden/make_tga_read_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import orc  # noqa: E402
from tinyrenderder_amd import scenes  # noqa: E402


def header(w, h, bpp, dtype, desc=0, idlen=0, cmap=0):
    hd = bytearray(18)
    hd[0] = idlen; hd[1] = cmap; hd[2] = dtype
    hd[12] = w & 255; hd[13] = w >> 8; hd[14] = h & 255; hd[15] = h >> 8
    hd[16] = bpp * 8; hd[17] = desc
    return bytes(hd)


def files():
    rng = scenes.SplitMix64(77)
    out = {}
    for bpp in (1, 3, 4):
        img = ((rng.u64(9 * 7 * bpp) % np.uint64(5)) * np.uint64(50)).astype(np.uint8).reshape(7, 9, bpp)     # few values -> real RLE runs
        for vflip in (0, 1):
            for rle in (0, 1):
                out[f"writer_bpp{bpp}_vflip{vflip}_rle{rle}"] = orc.tga_encode(img, bool(vflip), bool(rle))
    img = (rng.u64(6 * 5 * 3) % np.uint64(256)).astype(np.uint8).reshape(5, 6, 3)
    raw = img.tobytes()
    out["origin_right_bottom"] = header(6, 5, 3, 2, desc=0x10) + raw
    out["origin_right_top"] = header(6, 5, 3, 2, desc=0x30) + raw
    out["alpha_bits_in_descriptor"] = header(6, 5, 3, 2, desc=0x28) + raw
    out["id_field_5_bytes"] = header(6, 5, 3, 2, desc=0x20, idlen=5) + b"hello" + raw
    out["raw_truncated"] = header(6, 5, 3, 2, desc=0x20) + raw[:-17]
    out["raw_no_pixels"] = header(6, 5, 3, 2)
    out["raw_trailing_bytes"] = header(6, 5, 3, 2, desc=0x20) + raw + b"TRUEVISION-XFILE.\0"
    rle = orc.tga_encode(img, False, True)
    out["rle_truncated_mid_pixel"] = rle[:len(rle) - 4]
    out["rle_truncated_at_header"] = rle[:40]
    out["rle_empty_stream"] = header(6, 5, 3, 10, desc=0x20)
    out["rle_packet_overruns_image"] = header(2, 2, 3, 10, desc=0x20) + bytes([0x85, 1, 2, 3])            # 6 pixels into 4
    out["rle_raw_packet_overruns_image"] = header(2, 2, 1, 11, desc=0x20) + bytes([4, 9, 8, 7, 6, 5])    # 5 raw pixels into 4
    out["rle_exact_fit_then_garbage"] = header(2, 2, 3, 10) + bytes([0x83, 10, 20, 30, 0xff, 1, 2, 3])
    out["gray_rle_type11"] = header(4, 3, 1, 11, desc=0x20) + bytes([0x83, 200, 3, 1, 2, 3, 4, 0x83, 9])
    out["colormapped_type1"] = header(4, 3, 1, 1, cmap=1) + bytes(12)
    out["bits16"] = header(4, 3, 2, 2) + bytes(24)
    out["zero_width"] = header(0, 3, 3, 2)
    out["zero_height"] = header(3, 0, 3, 2)
    out["short_header"] = header(4, 3, 3, 2)[:11]
    out["empty_file"] = b""
    return out


def main():
    assert orc.ref_available(), "oracle/_ref/ref_harness missing: run `make -C oracle` where /root/reference exists"
    gold = {}
    d = os.path.join(HERE, "tga_read")
    os.makedirs(d, exist_ok=True)
    for name, data in files().items():
        open(os.path.join(d, name + ".tga"), "wb").write(data)
        img = orc.run_reference_tga_read(data)
        gold[name] = {"ok": False} if img is None else {"ok": True, "shape": list(img.shape), "sha256": scenes.digest(img)}
        print(name, gold[name])
    json.dump(gold, open(os.path.join(HERE, "tga_read_golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
