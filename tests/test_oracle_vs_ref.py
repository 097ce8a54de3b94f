"""CPU, only where oracle/_ref exists (the build container, or a box that received the prebuilt .so):
live randomized comparison of the oracle port with the reference's own compiled functions."""
import ctypes as C
import numpy as np
import pytest
import oracle_lib
import simdata

R = oracle_lib.ref_lib()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref not built (reference sources absent)")
u8p, i8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_uint64)


def P(a, t):
    return a.ctypes.data_as(t)


def test_live_dp_kernels():
    PL = oracle_lib.port_lib()
    rng = np.random.default_rng(99)
    for it in range(600):
        a = int(rng.choice([1, 1, 2])); b = int(rng.choice([2, 4, 1, 20]))
        M = np.zeros(25, np.int8); R.ref_fill_scmat(int(rng.integers(1, 3)), a, b, P(M, i8p))
        tlen = int(rng.integers(5, 700)); t = rng.integers(0, 4, tlen).astype(np.uint8)
        q = simdata.mutate(t[: int(rng.integers(2, min(tlen, 280)))], rng, float(rng.choice([0, 0.05, 0.2])), float(rng.choice([0, 0.03, 0.2])))
        if len(q) < 2:
            continue
        gp = [int(x) for x in rng.choice([[6, 1, 6, 1], [5, 2, 7, 1], [1, 1, 1, 1]])]
        o1 = (C.c_int * 6)(); o2 = (C.c_int * 6)()
        args = (len(q), P(q, u8p), tlen, P(t, u8p), P(M, i8p), *gp, int(rng.choice([100, 10])), 10, 100, int(rng.integers(1, 150)))
        R.ref_ksw_extend2(*args, o1); PL.oracle_extend1(*args, o2)
        assert list(o1) == list(o2)
        xtra = 0x80000 | 0x40000 | 19 | (0x10000 if len(q) * a < 250 else 0)
        s1 = (C.c_int * 7)(); s2 = (C.c_int * 7)()
        q1, t1, q2, t2 = q.copy(), t.copy(), q.copy(), t.copy()
        R.ref_ksw_align2(len(q), P(q1, u8p), tlen, P(t1, u8p), P(M, i8p), *gp, xtra, s1)
        PL.oracle_sw1(len(q), P(q2, u8p), tlen, P(t2, u8p), P(M, i8p), *gp, xtra, s2)
        assert list(s1) == list(s2)


def test_builder_files_load_in_reference(small_index):
    """<base>.{par,dau}.{bwt,sa} written by the repo's builder: the reference's loader accepts them, its
    bwt_sa over all ranks is a permutation, and bwt_cal_sa recomputes exactly the stored samples"""
    for tag in ("par", "dau"):
        h = C.c_void_p(R.ref_bwt_load((small_index.base + ".%s.bwt" % tag).encode(), (small_index.base + ".%s.sa" % tag).encode()))
        meta = (C.c_uint64 * 8)(); R.ref_bwt_meta(h, meta)
        n = int(meta[5])
        assert n == 2 * small_index.l_pac
        ks = np.arange(1, n + 1, 97, dtype=np.uint64)
        out = np.zeros(len(ks), np.uint64)
        R.ref_bwt_sa_batch(h, C.c_int64(len(ks)), P(ks, u64p), P(out, u64p))
        assert out.max() < n and len(set(out.tolist())) == len(ks)
        buf = np.zeros(int(meta[7]), np.uint64)
        nsa = R.ref_bwt_cal_sa(h, 32, P(buf, u64p), C.c_uint64(len(buf)))
        stored = np.fromfile(small_index.base + ".%s.sa" % tag, dtype=np.uint64)[7:]
        assert (buf[1:nsa] == stored).all()
