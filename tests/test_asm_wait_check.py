"""The text tower issues some global loads / stores from inline asm and waits for them itself (csrc/clip_tower.hip): the compiler
must not touch those registers in between.  tools/asm_wait_check.py walks the gfx950 ISA of the product build for that."""
import os
import shutil
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_checker_rules_on_synthetic_isa():
    import asm_wait_check as A
    isa = """
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[2:3], off sc1
	;;#ASMEND
	scratch_store_dword off, v11, off offset:4
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f32_e32 v1, v10, v11
	;;#ASMSTART
	global_store_dwordx4 v[2:3], v[20:23], off sc1
	;;#ASMEND
	v_mov_b32_e32 v20, 0
	;;#ASMSTART
	global_store_dwordx4 v[2:3], v[24:27], off sc1
	s_nop 0
	;;#ASMEND
	v_mov_b32_e32 v24, 0
""".splitlines()
    f = A.check_function("k", list(enumerate(isa, 1)))
    assert len(f) == 2
    assert "scratch_store_dword" in f[0][1] and f[0][2] == [11]
    assert "v_mov_b32_e32 v20" in f[1][1]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_text_tower_product_build_is_clean():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_wait_check.py"),
                        os.path.join(ROOT, "avlen_amd", "csrc", "clip_tower.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "6 kernels walked, 0 finding(s)" in r.stdout
