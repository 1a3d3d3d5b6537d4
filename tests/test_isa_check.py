"""tinyrenderder_amd/csrc/tools/check_scan_regs.py: k_raster requests a triangle's record with hand-written scalar loads and waits for
them one visit later; the compiler must not touch the 32 destination registers in between.  The build runs the check on the ISA of
the object that goes into the library; here it is run (a) on two hand-made kernels, one clean and one with a conflict behind a branch,
and (b) on the ISA the build left behind, if it is there."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tinyrenderder_amd", "csrc", "tools", "check_scan_regs.py")

CLEAN = """
_ZN1a8k_rasterILi0EEEvv:
\ts_lshl_b32 s0, s94, 7
\ts_load_dwordx16 s[52:67], s[84:85], s0 offset:0x0
\ts_load_dwordx16 s[68:83], s[84:85], s0 offset:0x40
\tv_add_f64 v[10:11], s[36:37], -v[50:51]
\ts_cbranch_vccz .LBB0_2
\tv_mul_f64 v[14:15], s[42:43], v[12:13]
.LBB0_2:
\ts_waitcnt lgkmcnt(0)
\tv_add_f64 v[10:11], s[52:53], -v[50:51]
\ts_endpgm
"""
# the branch target reads s[20:21] while the request for s[8:23] is still out; the fall-through path waits first
DIRTY = """
_ZN1a8k_rasterILi0EEEvv:
\ts_load_dwordx16 s[8:23], s[84:85], s0 offset:0x40
\ts_cbranch_vccz .LBB0_2
\ts_waitcnt lgkmcnt(0)
\ts_branch .LBB0_3
.LBB0_2:
\ts_mov_b64 exec, s[20:21]
\ts_waitcnt lgkmcnt(0)
.LBB0_3:
\tv_mov_b32 v1, s8
\ts_endpgm
"""


def run(text, tmp_path, name):
    f = tmp_path / name
    f.write_text(text)
    return subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)


def test_checker_accepts_a_clean_kernel(tmp_path):
    r = run(CLEAN, tmp_path, "clean.s")
    assert r.returncode == 0 and "1 k_raster variants checked, 0 conflicts" in r.stdout, r.stdout + r.stderr


def test_checker_finds_a_conflict_behind_a_branch(tmp_path):
    r = run(DIRTY, tmp_path, "dirty.s")
    assert r.returncode == 1 and "s_mov_b64 exec, s[20:21]" in r.stdout and "1 conflicts" in r.stdout, r.stdout + r.stderr


def test_built_kernel_isa_is_clean():
    isa = os.path.join(ROOT, "tinyrenderder_amd", "csrc", "build", "kernels_raster-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(isa):
        import pytest
        pytest.skip("no ISA left by the build in this checkout (the build itself runs the check)")
    r = subprocess.run([sys.executable, TOOL, isa], capture_output=True, text=True)
    assert r.returncode == 0 and "12 k_raster variants checked, 0 conflicts" in r.stdout, r.stdout[-2000:]
