"""CPU: the wave-parallel form of klib's introsort that the device's chain filter runs (k_regions.hip, rg_introsort_par: every Hoare partition
made at once from the two stop masks, the closing insertion pass as a stable rank) leaves EXACTLY the permutation of the reference's own
`ks_introsort` template (lib/aln/ksort.h:184-234, instantiated in oracle/_ref with mem_flt's comparator shape: records compared on one field,
descending).  The order of chains of equal weight is part of mem_chain_flt's result (memchain.c:426), so the permutation is what is pinned, on
keys with many ties, pre-sorted inputs (klib's depth limit and comb sort) and every length up to the 256 the kernel takes."""
import ctypes as C
import os
import random
import sys
import numpy as np
import pytest
from oracle_lib import ref_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))
import parsort_model as M   # noqa: E402


def _cases(rnd, n_cases):
    for case in range(n_cases):
        n = rnd.choice([3, 5, 16, 17, 18, 33, 40, 64, 65, 100, 128, 129, 200, 256]) if case % 3 else rnd.randint(1, 256)
        kind = rnd.randint(0, 4)
        if kind == 0:
            w = [rnd.randint(19, 22) for _ in range(n)]
        elif kind == 1:
            w = [rnd.randint(19, 22) if rnd.random() < 0.95 else rnd.randint(30, 150) for _ in range(n)]
        elif kind == 2:
            w = [rnd.randint(1, 300) for _ in range(n)]
        elif kind == 3:
            w = [20] * n
        else:
            w = sorted((rnd.randint(19, 40) for _ in range(n)), reverse=rnd.random() < 0.5)
        yield w


def test_model_equals_sequential_klib():
    rnd = random.Random(11)
    for w in _cases(rnd, 1500):
        a = [(w[i], i) for i in range(len(w))]
        b = list(a)
        assert M.klib_introsort(a, comb=True) and M.par_introsort(b, comb=True)
        assert a == b


def test_model_equals_the_reference_template():
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built")
    R.ref_introsort_kv_desc.argtypes = [C.c_int64, C.c_void_p]
    rnd = random.Random(12)
    n_comb = 0
    for w in _cases(rnd, 1500):
        n = len(w)
        kv = np.empty((n, 2), np.int64)
        kv[:, 0] = w
        kv[:, 1] = np.arange(n)
        R.ref_introsort_kv_desc(n, kv.ctypes.data_as(C.c_void_p))
        b = [(w[i], i) for i in range(n)]
        if not M.par_introsort(list(b)):
            n_comb += 1
        assert M.par_introsort(b, comb=True)
        assert [x[1] for x in b] == kv[:, 1].tolist(), (n, w)
    assert n_comb > 0   # the depth-limit case was among them
