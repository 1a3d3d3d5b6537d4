"""The C-ABI library loads and exports every symbol include/trgl.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from tinyrenderder_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "trgl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trgl_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_listed_in_python_mirror():
    assert _declared_symbols() == sorted(api.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(api.LIB_PATH), "libtrgl.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(api.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/trgl.h but not exported"


def test_struct_sizes_match_header():
    # trgl_uniforms: 16+3+3+3+1 doubles + 4 int32 ; trgl_stats: 2 u64 + 4 i32 + 2 doubles
    assert ctypes.sizeof(api.Uniforms) == 26 * 8 + 16
    assert ctypes.sizeof(api.Stats) == 16 + 16 + 16


def test_format_stats_needs_no_gpu():
    L = api.load_library()
    s = api.Stats(3, 5, 1, 2, 30, 40, -0.5, 0.25)
    buf = ctypes.create_string_buffer(256)
    assert L.trgl_format_stats(ctypes.byref(s), buf, 256) == 0
    assert buf.value.decode() == "DEBUG: triangles=3 fragments_drawn=5 bbox=[1,2] - [30,40] z-range=[-0.500000,0.250000]\n"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(api, "_lib", None)
    with pytest.raises(api.TrglError):
        api.load_library(str(tmp_path / "nope.so"))
    monkeypatch.setattr(api, "_lib", None)
