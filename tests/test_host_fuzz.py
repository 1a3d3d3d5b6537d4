"""The host parsers that read untrusted files (TGA reader, OBJ reader) under AddressSanitizer + UBSan: tests/host/fuzz_parsers.cpp
runs every truncation and a few thousand seeded mutations of the tests/golden/tga_read fixtures and of hand-written OBJ texts.
(The reference's own reader writes one pixel out of bounds on an overrunning RLE packet, tgaimage.cpp:128-160; the port must not.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_survive_mutated_files_under_asan_ubsan(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "host"), "asan"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(ROOT, "tests", "host", "fuzz_parsers_asan"), os.path.join(ROOT, "tests", "golden", "tga_read"),
                        str(tmp_path), "120"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "no sanitizer report" in r.stdout
