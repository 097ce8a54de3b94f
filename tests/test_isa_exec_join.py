"""CPU: the compiler's gfx950 assembly of every device source, looked over for the miscompile met in round 5: a register copy
every lane needs, placed at the end of a divergent region AHEAD of the s_or_b64 that restores EXEC (tools/dbg/exec_join_check.py).  The
kernels' `if (lane == 0) ...` regions are everywhere; whether a build has the bad shape depends on register allocation, i.e. on any edit."""
import os
import shutil
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))


def test_checker_finds_the_shape(tmp_path):
    import exec_join_check
    bad = tmp_path / "bad.s"
    bad.write_text("""_Z1kv:
	s_and_saveexec_b64 s[2:3], s[34:35]
	s_cbranch_execz .LBB0_2
	global_atomic_add_x2 v47, v[2:3], s[4:5] offset:296
.LBB0_2:
	v_mov_b64_e32 v[8:9], v[0:1]
	s_mov_b64 s[62:63], s[0:1]
	s_or_b64 exec, exec, s[2:3]
	s_branch .LBB0_4
.LBB0_3:
	v_mov_b64_e32 v[8:9], v[0:1]
.LBB0_4:
	v_mov_b32_e32 v7, v8
	s_endpgm
""")
    good = tmp_path / "good.s"
    good.write_text(bad.read_text().replace("	v_mov_b64_e32 v[8:9], v[0:1]\n	s_mov_b64 s[62:63], s[0:1]\n	s_or_b64 exec, exec, s[2:3]\n",
                                             "	s_or_b64 exec, exec, s[2:3]\n	v_mov_b64_e32 v[8:9], v[0:1]\n	s_mov_b64 s[62:63], s[0:1]\n"))
    assert len(exec_join_check.check(str(bad))) == 1
    assert exec_join_check.check(str(good)) == []
    # the else side of an if / else switches lanes on purpose
    els = tmp_path / "else.s"
    els.write_text("""_Z1kv:
	s_and_saveexec_b64 s[22:23], vcc
	s_xor_b64 s[22:23], exec, s[22:23]
	s_cbranch_execz .LBB0_6
	v_mov_b32_e32 v12, v41
.LBB0_6:
	s_andn2_saveexec_b64 s[22:23], s[22:23]
	v_mov_b32_e32 v12, v43
	s_or_b64 exec, exec, s[22:23]
	v_add_u32_e32 v12, v11, v12
	s_endpgm
""")
    assert exec_join_check.check(str(els)) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_product_assembly_has_no_copy_ahead_of_the_exec_restore():
    p = subprocess.run([os.path.join(ROOT, "tools", "isa_check.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    assert p.returncode == 0 and b"0 place(s) flagged" in p.stdout, p.stdout.decode()[-3000:]
