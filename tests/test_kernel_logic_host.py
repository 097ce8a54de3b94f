"""CPU: the per-lane state machine of the seed kernel (csrc/hip/seed_core.hpp, exactly the code k_seed
runs per lane) compiled with g++ and compared with the oracle on thousands of strand searches."""
import ctypes as C
import os
import subprocess
import numpy as np
import simdata
from biscuit_amd.api import default_opt, SEED_DT
from biscuit_amd import _lib as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _harness():
    so = os.path.join(ROOT, "tests", "_build", "libhostlogic.so")
    src = os.path.join(ROOT, "tests", "host_kernel_logic.cpp")
    deps = [src] + [os.path.join(ROOT, "biscuit_amd", "csrc", "hip", h) for h in ("seed_core.hpp", "seed_tab.hpp", "dev_common.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + ROOT + "/include", "-I" + ROOT + "/biscuit_amd/csrc/host",
                               "-I" + ROOT + "/biscuit_amd/csrc/hip", src, "-o", so, "-L" + ROOT + "/biscuit_amd", "-lbiscuit_amd",
                               "-Wl,-rpath," + ROOT + "/biscuit_amd"])
    return C.CDLL(so)


def test_seed_fsm_equals_oracle(small_index, port):
    import test_gpu_kernels as T
    H = _harness()
    seqs = T._reads(small_index, n_pairs=300) + [np.zeros(10, np.uint8), np.full(40, 4, np.uint8), np.array([0, 1, 2], np.uint8)]
    buf, offs = simdata.read_buffer(seqs)
    tasks = T._tasks(seqs, offs)
    for variant in range(2):
        opt = default_opt()
        if variant == 1:
            opt.min_seed_len, opt.split_width, opt.max_mem_intv, opt.split_factor = 15, 3, 8, 1.2
        port.set_opt(opt); port.set_reads(buf)
        port.counters(reset=True)
        pi, po = port.seed(opt, tasks)
        pc = port.counters()
        out = np.zeros((len(pi) + 16, 4), np.uint64); off = np.zeros(len(tasks) + 1, np.int64); ctr = (C.c_uint64 * 2)()
        rc = H.hostlogic_seed(small_index.h, C.byref(opt), buf.ctypes.data_as(C.c_void_p), C.c_int64(len(tasks)), tasks.ctypes.data_as(C.c_void_p),
                              512, out.ctypes.data_as(C.c_void_p), C.c_int64(len(out)), off.ctypes.data_as(C.c_void_p), ctr)
        assert rc == 0
        assert (po == off).all() and (pi == out[:len(pi)]).all()
        assert pc[0] == ctr[0] and pc[1] == ctr[1]    # identical FM-block touch counts (algorithmic bytes)
        assert len(pi) > 1000


def _seed_tab_case(index, port, seqs, variants=(0, 1), Ks=(0, 2, 5, 8, -1)):
    """the table machine (seed_tab.hpp) for several table depths against the oracle's interval lists: identical for every depth
    (0 = no table: every step an FM extension; -1 = the depth the device would pick for this text)"""
    import test_gpu_kernels as T
    H = _harness()
    H.hostlogic_seed_tab.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    buf, offs = simdata.read_buffer(seqs)
    tasks = T._tasks(seqs, offs)
    n_sym = 2 * index.l_pac
    depth = 0
    p3 = 3
    while depth < 18 and p3 <= n_sym // 8:
        depth += 1
        p3 *= 3
    looks = {}
    for variant in variants:
        opt = default_opt()
        if variant == 1:
            opt.min_seed_len, opt.split_width, opt.max_mem_intv, opt.split_factor = 15, 3, 8, 1.2
        if variant == 2:
            opt.min_seed_len, opt.split_width, opt.max_mem_intv, opt.split_factor = 6, 30, 50, 1.0
        port.set_opt(opt); port.set_reads(buf)
        pi, po = port.seed(opt, tasks)
        for K in Ks:
            K = depth if K < 0 else K
            out = np.zeros((len(pi) + 16, 4), np.uint64); off = np.zeros(len(tasks) + 1, np.int64); ctr = (C.c_uint64 * 3)()
            rc = H.hostlogic_seed_tab(index.h, C.byref(opt), buf.ctypes.data_as(C.c_void_p), C.c_int64(len(tasks)), tasks.ctypes.data_as(C.c_void_p), K,
                                      8192, out.ctypes.data_as(C.c_void_p), C.c_int64(len(out)), off.ctypes.data_as(C.c_void_p), ctr)
            assert rc == 0, (variant, K, rc)
            assert (po == off).all(), (variant, K)
            assert (pi == out[:len(pi)]).all(), (variant, K)
            looks[(variant, K)] = (ctr[0] + ctr[1], ctr[2])
    return looks, depth


def test_seed_table_machine_equals_oracle(small_index, port):
    import test_gpu_kernels as T
    rng = np.random.default_rng(11)
    seqs = T._reads(small_index, n_pairs=300) + [np.zeros(10, np.uint8), np.full(40, 4, np.uint8), np.array([0, 1, 2], np.uint8)]
    # reads with N runs, reads of one letter, reads shorter than the table is deep, a read ending in N
    seqs += [np.concatenate([s[:40], np.full(3, 4, np.uint8), s[43:]]) for s in seqs[:40]]
    seqs += [np.full(60, 0, np.uint8), np.full(25, 3, np.uint8), np.concatenate([seqs[5][:30], [4]]).astype(np.uint8), seqs[7][:19].copy(), seqs[9][:20].copy()]
    seqs += [rng.integers(0, 4, 70).astype(np.uint8) for _ in range(30)]
    looks, depth = _seed_tab_case(small_index, port, seqs, variants=(0, 1, 2))
    assert depth >= 8
    # the table takes most of the work: fewer FM blocks with it than without
    assert looks[(0, depth)][0] < 0.6 * looks[(0, 0)][0] and looks[(0, depth)][1] > 0


def test_reference_window_fetch():
    """dev_fetch_window (a dword of pac per lane, either strand) against the base-by-base definition, bntseq.h _get_pac
    and bns_get_seq's reverse-complement rule (bntseq.c:373-391)."""
    H = _harness()
    rng = np.random.default_rng(5)
    l_pac = 4099
    bases = rng.integers(0, 4, l_pac).astype(np.uint8)
    pac = np.zeros(l_pac // 4 + 1 + 16, np.uint8)
    for i, b in enumerate(bases):
        pac[i >> 2] |= b << ((~i & 3) << 1)
    both = np.concatenate([bases, 3 - bases[::-1]])
    cases = [(0, 1), (0, 768), (l_pac - 1, 1), (l_pac, 1), (2 * l_pac - 768, 768), (l_pac - 300, 300), (l_pac, 300), (3, 13), (l_pac + 3, 13)]
    for _ in range(2000):
        span = int(rng.integers(1, 769))
        beg = int(rng.integers(0, l_pac - span + 1)) + (l_pac if rng.random() < .5 else 0)
        cases.append((beg, span))
    for beg, span in cases:
        win = np.full(span + 8, 0xee, np.uint8)
        H.hostlogic_window(pac.ctypes.data_as(C.c_void_p), C.c_int64(l_pac), C.c_int64(beg), C.c_int(span), win.ctypes.data_as(C.c_void_p))
        assert (win[:span] == both[beg:beg + span]).all(), (beg, span)
        assert (win[span:] == 0xee).all()
