"""The library's settings registry (csrc/host/tune.c): one table of names, bsx_tune_set() between calls, $BSX_TUNE for a whole process --
instead of the three dozen environment variables of rounds 1-5, some of which were read with getenv() on every launch."""
import ctypes as C
import os
import re
import subprocess
import sys
import pytest
from biscuit_amd import _lib as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_set_get_and_unknown_names():
    L = B.lib()
    L.bsx_tune_str.restype = C.c_char_p
    L.bsx_tune_long.restype = C.c_long
    L.bsx_tune_long.argtypes = [C.c_char_p, C.c_long]
    names = B.tune_names()
    assert len(names) == len(set(names)) >= 25 and "seed_form" in names and "pos_cap" in names
    assert L.bsx_tune_str(b"pos_cap") is None and L.bsx_tune_long(b"pos_cap", 77) == 77
    B.tune("pos_cap", 5000)
    try:
        assert L.bsx_tune_str(b"pos_cap") == b"5000" and L.bsx_tune_long(b"pos_cap", 77) == 5000 and L.bsx_tune_is_set(b"pos_cap") == 1
    finally:
        B.tune("pos_cap", None)
    assert L.bsx_tune_is_set(b"pos_cap") == 0
    with pytest.raises(B.BsxError):
        B.tune("no_such_setting", 1)
    # every name has a description (the table IS the documentation)
    L.bsx_tune_doc.restype = C.c_char_p
    for i in range(len(names)):
        assert len(L.bsx_tune_doc(i)) > 10


def test_env_variable_is_parsed_once_and_unknown_names_are_reported():
    code = ("import ctypes as C; from biscuit_amd import _lib as B; L = B.lib(); L.bsx_tune_long.restype = C.c_long; L.bsx_tune_long.argtypes = [C.c_char_p, C.c_long];"
            "print(L.bsx_tune_long(b'pos_cap', -1), L.bsx_tune_long(b'x4', -1), L.bsx_tune_long(b'host_chain', -1), L.bsx_tune_is_set(b'seed_form'), L.bsx_phases())")
    env = dict(os.environ, PYTHONPATH=ROOT, BSX_TUNE="pos_cap=123,x4=0,host_chain,bogus=1,seed_form=classic", BSX_PHASES="2")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout.split() == [b"123", b"0", b"1", b"1", b"2"]
    assert b'no setting named "bogus"' in p.stderr


def test_tune_env_helper():
    e = B.tune_env({"BSX_POS_CAP": "9", "BSX_STREAM_DEPTH": "4", "x4": 0, "BSX_PHASES": "1", "BSX_CHUNK_SIZE": "60000"})
    assert e["BSX_STREAM_DEPTH"] == "4" and e["BSX_PHASES"] == "1" and e["BSX_CHUNK_SIZE"] == "60000"
    assert sorted(e["BSX_TUNE"].split(",")) == ["pos_cap=9", "x4=0"] and "BSX_POS_CAP" not in e


def test_the_library_reads_few_environment_variables():
    """<= 20 getenv names in the library, none of the tuning knobs among them"""
    names = set()
    for d in ("biscuit_amd/csrc/host", "biscuit_amd/csrc/hip"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".c", ".h", ".hip", ".hpp")):
                names |= set(re.findall(r'getenv\("([A-Za-z_0-9]+)"\)', open(os.path.join(ROOT, d, f)).read()))
    assert len(names) <= 20, sorted(names)
    assert names <= {"BSX_TUNE", "BSX_PHASES", "BSX_DEVICE", "BSX_HOST_THREADS", "BSX_INFLATE_THREADS", "BSX_STREAM_DEPTH", "BSX_NO_STREAM", "BSX_CHUNK_SIZE",
                     "BSX_PROF_SAMPLE", "BSX_TRACE_ALLOC", "BSX_INDEX_TRACE",
                     # several processes (csrc/host/cli.c, bsx_align_main_ranks_with): what every launcher sets, where the output goes, where the ranks meet
                     "RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "BSX_OUT", "BSX_GATHER_ID"}, sorted(names)
