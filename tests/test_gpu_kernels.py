"""-m gpu parity tests: every HIP kernel, called through the C ABI, against the CPU restatement in
oracle/ (itself pinned to the real reference functions by test_oracle_vs_ref.py).  Bit-exact."""
import numpy as np
import pytest
import simdata
from biscuit_amd.api import default_opt, SEED_DT, SA_DT

pytestmark = pytest.mark.gpu


def _reads(small_index, n_pairs=400, read_len=150, seed=5, **kw):
    import os
    contigs = _contigs(small_index)
    pairs = simdata.make_pairs(contigs, n_pairs, read_len, seed, sub=0.01, indel=0.004, pbat_frac=0.2, chimera_frac=0.05,
                               bad_mate_frac=0.05, n_frac=0.05, **kw)
    seqs = []
    for _, r1, r2 in pairs:
        seqs += [r1, r2]
    return seqs


def _contigs(index):
    """decode the forward genome back from <base>.bis.pac + .ann"""
    pac = np.fromfile(index.base + ".bis.pac", dtype=np.uint8)
    l = index.l_pac
    i = np.arange(l)
    g = (pac[i >> 2] >> ((~i & 3) << 1)) & 3
    out = []
    with open(index.base + ".bis.ann") as f:
        f.readline()
        while True:
            h = f.readline()
            if not h:
                break
            name = h.split()[1]
            off, ln, _ = [int(x) for x in f.readline().split()]
            out.append((name, g[off:off + ln].astype(np.uint8)))
    return out


def _tasks(seqs, offs):
    t = np.zeros(2 * len(seqs), dtype=SEED_DT)
    for i, s in enumerate(seqs):
        for p in (0, 1):
            t[2 * i + p] = (offs[i], len(s), p)
    return t


def test_seed_and_sa_match_oracle(small_index, port, device):
    opt = default_opt()
    seqs = _reads(small_index) + [np.zeros(10, np.uint8), np.full(40, 4, np.uint8), np.zeros(0, np.uint8)]
    seqs[-1] = np.array([0, 1, 2], np.uint8)
    buf, offs = simdata.read_buffer(seqs)
    tasks = _tasks(seqs, offs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    pi, po = port.seed(opt, tasks)
    port.counters(reset=True)
    pi, po = port.seed(opt, tasks)
    device.counters(reset=True)
    di, do = device.seed(opt, tasks)
    assert (po == do).all()
    assert pi.shape == di.shape and (pi == di).all()
    assert int(po[-1]) > len(seqs)          # the test is not vacuous
    pc, dc = port.counters(), device.counters()
    assert pc[0] == dc[0] and pc[1] == dc[1], (pc, dc)
    # K3 on every occurrence the chaining step would look up (capped like memchain.c:325)
    jobs = []
    for t in range(len(tasks)):
        for k in range(po[t], po[t + 1]):
            x0, _, x2, _ = [int(v) for v in pi[k]]
            for j in range(min(x2, 50)):
                jobs.append((x0 + j, int(tasks[t]["parent"]), 0))
    jobs = np.array(jobs, dtype=SA_DT)
    assert (port.sa(jobs) == device.sa(jobs)).all()


def test_seed_overflow_path(small_index, port, device):
    """a poly-A read against a genome with a long A run is pathological; capacity retry must agree"""
    opt = default_opt()
    opt.max_mem_intv = 1 << 30
    seqs = _reads(small_index, n_pairs=50, read_len=100, seed=9)
    buf, offs = simdata.read_buffer(seqs)
    tasks = _tasks(seqs, offs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    pi, po = port.seed(opt, tasks)
    di, do = device.seed(opt, tasks)
    assert (po == do).all() and (pi == di).all()
