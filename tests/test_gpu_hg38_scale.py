"""-m gpu: BASELINE configs[1] at its real size -- an hg38-sized genome (3.1 Gbp, SYNTHETIC: hg38 is not available
offline; SURVEY 8(d) config 2 fallback) whose two FM indices (6.2 G symbols each) are built on the device.  Suffix-array
ranks and forward-reverse coordinates exceed 2^32 here, which no smaller genome exercises:
  * the index itself: adjacent suffixes are in lexicographic order (checked on the text, independently of any FM code),
    LF steps agree with the suffix array, counts add up;
  * K1-K3 on the device == the CPU restatement at ranks/positions > 2^32;
  * the whole path: SAM of the HIP pipeline == SAM of the CPU restatement byte for byte on a sample of pairs, records
    valid against the genome, reads found where they were simulated from (including reverse-strand hits whose
    forward-reverse coordinates are > 2^32).
Set BSX_TEST_GENOME_MBP to run the same checks on a smaller genome."""
import ctypes as C
import os
import numpy as np
import pytest
import samcheck
import simdata
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, Device, default_opt, SEED_DT, SA_DT

pytestmark = pytest.mark.gpu
MBP = float(os.environ.get("BSX_TEST_GENOME_MBP", "3100"))


class PacContig:
    """str-like view of one contig of the packed genome (what samcheck indexes and slices)"""

    def __init__(self, pac, off, n):
        self.pac, self.off, self.n = pac, off, n

    def __len__(self):
        return self.n

    def _codes(self, a, b):
        i = np.arange(self.off + a, self.off + b, dtype=np.int64)
        return (self.pac[i >> 2] >> ((~i & 3) << 1)) & 3

    def __getitem__(self, k):
        if isinstance(k, slice):
            a, b, _ = k.indices(self.n)
            return simdata.BASES[self._codes(a, b)].tobytes().decode()
        return "ACGT"[int(self._codes(k, k + 1)[0])]


def _make(profile):
    L = B.lib()
    n = int(MBP * 1e6)
    idx = Index.synthetic(n, seed=2024, n_contigs=24 if n >= 1_000_000_000 else 8, profile=profile)   # (bench.py's genomes: same seed and shape)
    dev = Device(0)
    dev.build_index(idx, fill_host=True)
    L.bsx_index_pac.restype = C.POINTER(C.c_uint8)
    L.bsx_index_pac.argtypes = [C.c_void_p]
    pac = np.ctypeslib.as_array(L.bsx_index_pac(idx.h), shape=(idx.l_pac // 4 + 1,))
    L.bsx_index_contig.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.bsx_index_n_seqs.argtypes = [C.c_void_p]
    contigs = []
    for i in range(L.bsx_index_n_seqs(idx.h)):
        nm, off, ln = C.c_char_p(), C.c_int64(), C.c_int64()
        B.check(L.bsx_index_contig(idx.h, i, C.byref(nm), C.byref(off), C.byref(ln)), "contig")
        contigs.append((nm.value.decode(), off.value, ln.value))
    import oracle_lib
    port = oracle_lib.Port(idx, n_threads=16)
    return dict(idx=idx, dev=dev, pac=pac, contigs=contigs, port=port, l_pac=idx.l_pac)


@pytest.fixture(scope="module")
def big():
    g = _make(0)
    yield g
    g["dev"].close()
    g["idx"].close()


def _text(pac, l_pac, parent, pos, n):
    """n symbols of the converted text [fwd ; revcomp(fwd)] from pos"""
    i = np.arange(pos, min(pos + n, 2 * l_pac), dtype=np.int64)
    f = np.where(i < l_pac, i, 2 * l_pac - 1 - i)
    b = (pac[f >> 2] >> ((~f & 3) << 1)) & 3
    b = np.where(i < l_pac, b, 3 - b)
    return np.where(b == 1, 3, b) if parent else np.where(b == 2, 0, b)


def test_index_sorted_and_consistent(big):
    dev, port, pac, l_pac = big["dev"], big["port"], big["pac"], big["l_pac"]
    n = 2 * l_pac
    rng = np.random.default_rng(1)
    for parent in (1, 0):
        k = rng.integers(1, n, 3000).astype(np.uint64)   # ranks up to 6.2e9 (> 2^32 when the genome is hg38-sized)
        if n > 1 << 32:
            k[:1000] = rng.integers(1 << 32, n, 1000).astype(np.uint64)
        jobs = np.zeros(2 * len(k), dtype=SA_DT)
        jobs["k"] = np.concatenate([k, k + 1]); jobs["parent"] = parent
        pos = dev.sa(jobs)
        assert (pos == port.sa(jobs)).all()                # device LF walk + dense sample == host walk over the file-format arrays
        assert (pos < n).all()
        a, b = pos[:len(k)], pos[len(k):]
        for x, y in zip(a, b):                             # suffix of rank k < suffix of rank k+1, compared on the text itself
            tx, ty = _text(pac, l_pac, parent, int(x), 4000), _text(pac, l_pac, parent, int(y), 4000)
            m = min(len(tx), len(ty))
            d = np.nonzero(tx[:m] != ty[:m])[0]
            if len(d):
                assert tx[d[0]] < ty[d[0]], (parent, int(x), int(y))
            else:
                assert len(tx) < len(ty) or m == 4000


def _reads_at(big, starts, rev, read_len=150):
    """bisulfite reads (directional R1-like) from given forward starts / strands: exercises chosen coordinates"""
    pac, l_pac = big["pac"], big["l_pac"]
    rng = np.random.default_rng(7)
    seqs = []
    for s, r in zip(starts, rev):
        i = np.arange(s, s + read_len, dtype=np.int64)
        g = ((pac[i >> 2] >> ((~i & 3) << 1)) & 3).astype(np.uint8)
        if r:
            g = (3 - g[::-1]).astype(np.uint8)
        c = g == 1
        g[c & (rng.random(read_len) > 0.2)] = 3
        seqs.append(g)
    return seqs


def test_seed_and_sa_beyond_2_32(big):
    dev, port, l_pac = big["dev"], big["port"], big["l_pac"]
    opt = default_opt()
    rng = np.random.default_rng(3)
    # reads from the first tenth of the genome on the reverse strand: forward-reverse coordinates near 2*l_pac (> 2^32)
    starts = rng.integers(1000, l_pac // 10, 600)
    seqs = _reads_at(big, starts, [True] * 300 + [False] * 300)
    buf, offs = simdata.read_buffer(seqs)
    tasks = np.zeros(2 * len(seqs), dtype=SEED_DT)
    for i, s in enumerate(seqs):
        for p in (0, 1):
            tasks[2 * i + p] = (offs[i], len(s), p)
    for be in (port, dev):
        be.set_opt(opt)
        be.set_reads(buf)
    pi, po = port.seed(opt, tasks)
    di, do = dev.seed(opt, tasks)
    assert (po == do).all() and pi.shape == di.shape and (pi == di).all()
    assert int(po[-1]) > len(tasks)
    if 2 * l_pac > 1 << 32:
        assert (pi[:, 0] > (1 << 32)).any() and (pi[:, 1] > (1 << 32)).any()   # ranks beyond 32 bits were really used
    jobs = []
    for t in range(len(tasks)):
        for k in range(po[t], po[t + 1]):
            x0, _, x2, _ = [int(v) for v in pi[k]]
            for j in range(min(x2, 20)):
                jobs.append((x0 + j, int(tasks[t]["parent"]), 0))
    jobs = np.array(jobs, dtype=SA_DT)
    pp, dp = port.sa(jobs), dev.sa(jobs)
    assert (pp == dp).all()
    if 2 * l_pac > 1 << 32:
        assert (pp > (1 << 32)).any()


def test_sam_identical_and_where_simulated(big):
    L = B.lib()
    idx, dev, port = big["idx"], big["dev"], big["port"]
    n_pairs = 20000
    opt = default_opt()
    opt.n_threads = 16
    opt.flag |= 0x10 | 0x2
    L.bsx_sim_pairs_truth.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p), C.c_void_p]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    truth = np.zeros(2 * n_pairs, dtype=np.int64)
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs_truth(idx.h, n_pairs, 150, 4242, 200, 500, 0.005, 0.1, C.byref(p), truth.ctypes.data_as(C.c_void_p)), "sim_pairs")
    n = 2 * n_pairs
    r = C.cast(p, C.POINTER(B.Read))
    try:
        B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, n, p, None), "process_seqs")
        hip = [C.string_at(r[i].sam) for i in range(n)]
        L.bsx_sim_reset_reads(p, n)
        be = port.backend()
        os.environ["BSX_HOST_THREADS"] = "16"
        B.check(L.bsx_process_seqs_backend(C.byref(be), C.byref(opt), idx.h, 0, n, p, None), "process_seqs(cpu restatement)")
        cpu = [C.string_at(r[i].sam) for i in range(n)]
        bad = [i for i in range(n) if hip[i] != cpu[i]]
        assert not bad, "SAM differs for %d reads, first: %r vs %r" % (len(bad), hip[bad[0]], cpu[bad[0]])
        genome = {nm: PacContig(big["pac"], off, ln) for nm, off, ln in big["contigs"]}
        ctg_off = {nm: off for nm, off, ln in big["contigs"]}
        hdr, recs = samcheck.parse_sam(b"".join(hip[:8000]).decode())
        for rec in recs:
            samcheck.check_record(rec, genome, 150)
        samcheck.check_pairs(recs)
        prim = [x for x in recs if not x["flag"] & 0x900]
        assert len(prim) == 8000
        assert np.mean([not x["flag"] & 4 for x in prim]) > 0.97
        # where they were simulated from: primary records of confidently mapped reads lie inside their fragment
        ok = tot = 0
        for x in prim:
            if x["flag"] & 4 or x["mapq"] < 30:
                continue
            pi = int(x["qname"][1:])
            s, fl = int(truth[2 * pi]), int(truth[2 * pi + 1]) >> 1
            g = ctg_off[x["rname"]] + x["pos"] - 1
            tot += 1
            ok += s - 10 <= g <= s + fl + 10
        assert tot > 6000 and ok / tot > 0.995, (ok, tot)
    finally:
        L.bsx_sim_free_reads(p, n)


def _run_both(big, opt, p, n, pes=None):
    """HIP pipeline and CPU restatement on the same reads; returns the two lists of SAM texts"""
    L = B.lib()
    idx, dev, port = big["idx"], big["dev"], big["port"]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    r = C.cast(p, C.POINTER(B.Read))
    B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, n, p, pes), "process_seqs")
    hip = [C.string_at(r[i].sam) for i in range(n)]
    L.bsx_sim_reset_reads(p, n)
    be = port.backend()
    os.environ["BSX_HOST_THREADS"] = "16"
    B.check(L.bsx_process_seqs_backend(C.byref(be), C.byref(opt), idx.h, 0, n, p, pes), "process_seqs(cpu restatement)")
    cpu = [C.string_at(r[i].sam) for i in range(n)]
    L.bsx_sim_reset_reads(p, n)
    return hip, cpu


def test_directional_and_pbat_libraries(big):
    """BASELINE configs[3] at full genome size: the same pairs (a tenth of them PBAT-like, the two reads trading roles) aligned as a
    non-directional library (-b 0: four strand searches per pair), as a directional one (-b 1: two; -b 3 is the same thing for pairs): SAM == CPU restatement each time, and -b 1 leaves the PBAT-like pairs (only those) without a proper hit."""
    L = B.lib()
    idx = big["idx"]
    n_pairs = 6000
    L.bsx_sim_pairs_truth.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p), C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs_truth(idx.h, n_pairs, 150, 77, 200, 500, 0.005, 0.1, C.byref(p), None), "sim_pairs")
    n = 2 * n_pairs
    try:
        mapped = {}
        for b in (0, 1, 3):
            opt = default_opt()
            opt.n_threads = 16
            opt.flag |= 0x10 | 0x2
            opt.parent = b
            hip, cpu = _run_both(big, opt, p, n)
            bad = [i for i in range(n) if hip[i] != cpu[i]]
            assert not bad, "-b %d: SAM differs for %d reads, first: %r vs %r" % (b, len(bad), hip[bad[0]], cpu[bad[0]])
            mapped[b] = np.mean([not (int(s.split(b"\t")[1]) & 4) for s in hip])
        assert mapped[0] > 0.97
        assert mapped[0] > mapped[1] > 0.85          # the PBAT-like tenth is what -b 1 cannot place
        assert mapped[3] == mapped[1]                # paired-end: any -b other than 0 means directional (bwamem.c:352-372)
    finally:
        L.bsx_sim_free_reads(p, n)


def test_long_reads(big):
    """BASELINE configs[4] shape at full genome size: single 1 kb reads (the seed-SW filter, bands of 100-200, i16 local alignment,
    global alignments with up to 400 x 1000 traceback bytes): SAM == CPU restatement; reads found where simulated."""
    L = B.lib()
    idx = big["idx"]
    n_pairs = 1500
    L.bsx_sim_pairs_truth.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p), C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs_truth(idx.h, n_pairs, 1000, 5, 1000, 1400, 0.01, 0.0, C.byref(p), None), "sim_pairs")
    n = 2 * n_pairs
    try:
        opt = default_opt()
        opt.n_threads = 16
        opt.flag |= 0x10            # single-end: every read on its own
        hip, cpu = _run_both(big, opt, p, n)
        bad = [i for i in range(n) if hip[i] != cpu[i]]
        assert not bad, "SAM differs for %d reads, first: %r vs %r" % (len(bad), hip[bad[0]][:300], cpu[bad[0]][:300])
        prim = [s.split(b"\n")[0].split(b"\t") for s in hip]
        assert np.mean([not (int(f[1]) & 4) for f in prim]) > 0.97
        assert np.mean([int(f[4]) >= 30 for f in prim]) > 0.9
    finally:
        L.bsx_sim_free_reads(p, n)


def test_command_line_equals_end_to_end_oracle_from_index_files(big, tmp_path_factory):
    """The index the device built, written out as the reference's seven files, then the two command lines on those files: the product
    (`biscuit_align`: HIP kernels + C host pipeline) and oracle/e2e.py -- the end-to-end restatement that shares no host code with the
    product and runs the reference's own bwt_restore_* / bwt_smem1a / bwt_sa / ksw_* (oracle/_ref) over the same files.  Suffix-array
    ranks and forward-reverse coordinates are beyond 2^32 on both sides.  Needs ~11 GB of disk for the files (skipped without it)."""
    import shutil
    import subprocess
    import sys
    import e2e_cases as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libbiscuit_ref.so")):
        pytest.skip("oracle/_ref is absent")
    d = str(tmp_path_factory.mktemp("hg38_files"))
    need = int(big["l_pac"] * 3.6) + (2 << 30)
    if shutil.disk_usage(d).free < need:
        pytest.skip("not enough disk for the index files (%d GB)" % (need >> 30))
    L = B.lib()
    idx = big["idx"]
    idx.save(d + "/g")
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_sim_write_fastq.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    n_pairs = 2500
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 4242, 200, 500, 0.008, 0.15, C.byref(p)), "sim_pairs")
    B.check(L.bsx_sim_write_fastq(p, 2 * n_pairs, (d + "/r1.fq").encode(), (d + "/r2.fq").encode(), 0), "write_fastq")
    L.bsx_sim_free_reads(p, 2 * n_pairs)
    try:
        for args in (["-@", "4", "g", "r1.fq", "r2.fq"], ["-@", "4", "-b", "1", "g", "r1.fq"]):
            want = E.run_e2e(args, d)
            got = E.run_exe(os.path.join(root, "biscuit_amd", "biscuit_align"), args, d)
            assert got.count(b"\n") > n_pairs
            E.assert_same_sam(got, want, " ".join(args))
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
