"""CPU: the host pipeline end to end (`oracle_align` = biscuit_align's driver over the CPU restatement
of the kernels).  Checks SAM validity against the genome, thread-count independence, option handling
and a committed regression fixture.  NOTE: the fixture under tests/golden/pe_small.sam was produced by
THIS repository's CPU path -- the reference's chaining/pairing/formatting files cannot be built offline
(un-vendored headers), so it guards against regressions, it is not a reference output."""
import os
import subprocess
import numpy as np
import pytest
import simdata
import samcheck
from biscuit_amd.api import Index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "oracle", "oracle_align")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(args, cwd):
    p = subprocess.run([CPU] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return "\n".join(l for l in p.stdout.decode().split("\n") if not l.startswith("@PG"))


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("cpu_align"))
    contigs = simdata.make_genome(300000, seed=21, n_contigs=3)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    pairs = simdata.make_pairs(contigs, 1500, 100, 1, frag=(180, 320), sub=0.005)
    hard = simdata.make_pairs(contigs, 1500, 150, 2, sub=0.01, indel=0.006, pbat_frac=0.3, chimera_frac=0.06, bad_mate_frac=0.06, n_frac=0.03)
    for tag, ps in (("a", pairs), ("b", hard)):
        simdata.write_fastq(d + "/%s1.fq" % tag, [(n, a) for n, a, b in ps])
        simdata.write_fastq(d + "/%s2.fq" % tag, [(n, b) for n, a, b in ps])
    simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, 40, 1000, 5))
    # edge cases: empty-ish, all-N, shorter than the seed length, adaptor tail
    with open(d + "/edge.fq", "w") as f:
        f.write("@e1\nACGT\n+\nIIII\n@e2\n%s\n+\n%s\n@e3\n%s\n+\n%s\n" % ("N" * 60, "I" * 60, "ACGTTGCATG" * 2, "I" * 20))
    return d


def test_sam_valid_and_mostly_correct(data):
    genome = samcheck.load_genome(data + "/g.fa")
    for args, rl in ((["g", "a1.fq", "a2.fq"], 100), (["g", "b1.fq", "b2.fq"], 150), (["-b", "1", "g", "a1.fq", "a2.fq"], 100)):
        hdr, recs = samcheck.parse_sam(run(["-@", "4"] + args, data))
        assert [h for h in hdr if h.startswith("@SQ")] == sorted(h for h in hdr if h.startswith("@SQ"))
        for r in recs:
            samcheck.check_record(r, genome, rl)
        samcheck.check_pairs(recs)
        prim = [r for r in recs if not r["flag"] & 0x900]
        assert len(prim) == 3000
        mapped = [r for r in prim if not r["flag"] & 4]
        assert len(mapped) > 0.97 * len(prim)
        assert np.mean([r["mapq"] >= 40 for r in mapped]) > 0.85
        assert all(r["tags"].get("YD") in ("f", "r", "u") for r in mapped)
    # directional simple pairs: nearly all proper pairs
    hdr, recs = samcheck.parse_sam(run(["-@", "4", "g", "a1.fq", "a2.fq"], data))
    assert np.mean([bool(r["flag"] & 2) for r in recs if not r["flag"] & 0x900]) > 0.95


def test_threads_do_not_change_output(data):
    a = run(["-@", "1", "g", "b1.fq", "b2.fq"], data)
    os.environ["BSX_HOST_THREADS"] = "7"
    try:
        b = run(["-@", "1", "g", "b1.fq", "b2.fq"], data)
    finally:
        del os.environ["BSX_HOST_THREADS"]
    assert a == b


def test_options_take_effect_and_edge_reads(data):
    base = run(["-@", "4", "g", "b1.fq", "b2.fq"], data)
    for extra in (["-a"], ["-b", "1"], ["-S", "-P"], ["-T", "60"], ["-Y"], ["-I", "300,50"], ["-k", "25"]):
        assert run(["-@", "4"] + extra + ["g", "b1.fq", "b2.fq"], data) != base, extra
    assert "\tRG:Z:x" in run(["-@", "2", "-R", "@RG\\tID:x\\tSM:y", "g", "a1.fq"], data)
    hdr, recs = samcheck.parse_sam(run(["-@", "2", "g", "edge.fq"], data))
    assert len(recs) == 3 and all(r["flag"] & 4 for r in recs)
    hdr, recs = samcheck.parse_sam(run(["-@", "2", "-F", "g", "a1.fq"], data))
    assert not hdr
    hdr, recs = samcheck.parse_sam(run(["-@", "4", "g", "long.fq"], data))
    genome = samcheck.load_genome(data + "/g.fa")
    for r in recs:
        samcheck.check_record(r, genome)
    assert sum(1 for r in recs if not r["flag"] & 0x904) >= 38
    one = run(["-1", "ACGTTGCATGACGTTGCATGACGTTGCATGACGTTGCATG", "g"], data)
    assert "inputread" in one


def test_regression_fixture():
    """committed FASTA/FASTQ -> committed SAM (see module docstring for what this fixture is)"""
    import tempfile
    d = tempfile.mkdtemp()
    Index.build(os.path.join(GOLD, "g24k.fa"), d + "/g").close()
    got = run(["-@", "2", d + "/g", os.path.join(GOLD, "pe_small_1.fq"), os.path.join(GOLD, "pe_small_2.fq")], d)
    want = open(os.path.join(GOLD, "pe_small.sam")).read()
    assert got.strip() == want.strip()


def _load_pairs(L, B, C, idx, n_pairs, seed):
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, 100, seed, 150, 400, 0.01, 0.2, C.byref(p)), "sim_pairs")
    return p


def test_chunk_stream_equals_chunk_by_chunk(data):
    """The chunk pipeline (front halves of later chunks in flight while an older chunk's back half runs) over
    CPU-restatement contexts: every chunk's SAM text equals what the synchronous bsx_process_seqs path gives,
    at every depth."""
    import ctypes as C
    from biscuit_amd import _lib as B
    from biscuit_amd.api import default_opt
    from oracle_lib import Port
    L = B.lib()
    idx = Index(data + "/g")
    opt = default_opt()
    opt.n_threads = 2
    opt.flag |= 0x10 | 0x2
    n_pairs, n_chunks = 300, 5
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_stream_open_backends.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.bsx_stream_push.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.bsx_stream_flush.argtypes = [C.c_void_p]
    L.bsx_stream_close.argtypes = [C.c_void_p]
    L.bsx_stream_close.restype = None
    chunks = [_load_pairs(L, B, C, idx, n_pairs, 50 + k) for k in range(n_chunks)]
    ports = [Port(idx, 2) for _ in range(4)]

    def sam_of(k):
        r = C.cast(chunks[k], C.POINTER(B.Read))
        return b"".join(C.string_at(r[i].sam) for i in range(2 * n_pairs))

    try:
        be0 = ports[0].backend()
        want = []
        for k in range(n_chunks):
            B.check(L.bsx_process_seqs_backend(C.byref(be0), C.byref(opt), idx.h, 2 * n_pairs * k, 2 * n_pairs, chunks[k], None), "process")
            want.append(sam_of(k))
            L.bsx_sim_reset_reads(chunks[k], 2 * n_pairs)
        assert len(set(want)) == n_chunks and all(w.count(b"\n") >= 2 * n_pairs for w in want)
        for depth in (1, 2, 3, 4):
            bes = (B.Backend * depth)(*[ports[i].backend() for i in range(depth)])
            s = C.c_void_p()
            B.check(L.bsx_stream_open_backends(depth, bes, C.byref(opt), idx.h, None, C.byref(s)), "open")
            for k in range(n_chunks):
                B.check(L.bsx_stream_push(s, 2 * n_pairs * k, 2 * n_pairs, chunks[k]), "push")
                done = k - (depth - 1)
                if done >= 0:
                    assert sam_of(done) == want[done], (depth, done)
            B.check(L.bsx_stream_flush(s), "flush")
            for k in range(n_chunks):
                assert sam_of(k) == want[k], (depth, k)
                L.bsx_sim_reset_reads(chunks[k], 2 * n_pairs)
            L.bsx_stream_close(s)
    finally:
        for c in chunks:
            L.bsx_sim_free_reads(c, 2 * n_pairs)
        idx.close()


def test_pending_and_declined_strand_searches_take_the_host_path(data):
    """The device regions pass may decline a strand search (-1: seed and chain it on the host) or leave it pending
    (BSX_REGIONS_PENDING: still being seeded again on a side stream; regions_finish settles it at the start of the chunk's
    back half), or hand its interval list back (< -1: chain it on the host from these).  A stand-in regions pass that
    computes no regions — a fifth of the strand searches pending, a fifth handed back with their intervals (seeded by the
    CPU restatement inside the stand-in), the others declined, the pending ones declined by regions_finish — must
    therefore give the SAM of the plain host path: the strand searches go through host chaining in two instalments, one
    in the front half and one in the back half."""
    import ctypes as C
    from biscuit_amd import _lib as B
    from biscuit_amd.api import default_opt
    from oracle_lib import Port
    L = B.lib()
    idx = Index(data + "/g")
    opt = default_opt()
    opt.n_threads = 2
    opt.flag |= 0x10 | 0x2
    n_pairs, n_chunks = 300, 3
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_stream_open_backends.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.bsx_stream_push.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.bsx_stream_flush.argtypes = [C.c_void_p]
    L.bsx_stream_close.argtypes = [C.c_void_p]
    L.bsx_stream_close.restype = None
    chunks = [_load_pairs(L, B, C, idx, n_pairs, 70 + k) for k in range(n_chunks)]
    ports = [Port(idx, 2) for _ in range(2)]
    PENDING = -100
    p64, p32, pp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)
    BATCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, pp, p64, p64, p32, pp, p64, p64)
    FINISH = C.CFUNCTYPE(C.c_int, C.c_void_p, pp, p64, p64, p32)
    n_of, calls = {}, {"batch": 0, "finish": 0, "pending": 0, "handed_back": 0}
    PL = ports[0].L   # the oracle library (test infrastructure), not the product library
    PL.oracle_port_seed_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, pp, p64, p64]

    def batch(ctx, o, n, tasks, out, cap, off, cnt, di, dc, doff):
        n_of[ctx] = n
        calls["batch"] += 1
        for i in range(n):
            off[i] = 0
            cnt[i] = PENDING if i % 5 == 0 else (-10 if i % 5 == 1 else -1)
        sel = [i for i in range(n) if i % 5 == 1]   # these come back with their interval lists, in task order
        src = C.cast(tasks, C.POINTER(C.c_int32))   # bsx_seed_task_t = 3 x 32 bits
        sub = (C.c_int32 * (3 * len(sel) + 3))()
        for j, i in enumerate(sel):
            sub[3 * j], sub[3 * j + 1], sub[3 * j + 2] = src[3 * i], src[3 * i + 1], src[3 * i + 2]
        calls["handed_back"] += len(sel)
        return PL.oracle_port_seed_batch(ctx, o, len(sel), sub, di, dc, doff)

    def finish(ctx, out, cap, off, cnt):
        calls["finish"] += 1
        for i in range(n_of[ctx]):
            if cnt[i] == PENDING:
                cnt[i] = -1
                calls["pending"] += 1
        return 0

    cb = (BATCH(batch), FINISH(finish))   # kept alive for the duration of the test

    def sam_of(k):
        r = C.cast(chunks[k], C.POINTER(B.Read))
        return b"".join(C.string_at(r[i].sam) for i in range(2 * n_pairs))

    def standin(port):
        be = port.backend()
        be.regions_batch = C.cast(cb[0], C.c_void_p).value
        be.regions_finish = C.cast(cb[1], C.c_void_p).value
        return be

    try:
        be0 = ports[0].backend()
        want = []
        for k in range(n_chunks):
            B.check(L.bsx_process_seqs_backend(C.byref(be0), C.byref(opt), idx.h, 2 * n_pairs * k, 2 * n_pairs, chunks[k], None), "process")
            want.append(sam_of(k))
            L.bsx_sim_reset_reads(chunks[k], 2 * n_pairs)
        be1 = standin(ports[0])
        for k in range(n_chunks):
            B.check(L.bsx_process_seqs_backend(C.byref(be1), C.byref(opt), idx.h, 2 * n_pairs * k, 2 * n_pairs, chunks[k], None), "process")
            assert sam_of(k) == want[k], k
            L.bsx_sim_reset_reads(chunks[k], 2 * n_pairs)
        assert calls["batch"] == n_chunks and calls["finish"] == n_chunks and calls["pending"] > 0 and calls["handed_back"] > 0
        bes = (B.Backend * 2)(standin(ports[0]), standin(ports[1]))
        s = C.c_void_p()
        B.check(L.bsx_stream_open_backends(2, bes, C.byref(opt), idx.h, None, C.byref(s)), "open")
        for k in range(n_chunks):
            B.check(L.bsx_stream_push(s, 2 * n_pairs * k, 2 * n_pairs, chunks[k]), "push")
        B.check(L.bsx_stream_flush(s), "flush")
        for k in range(n_chunks):
            assert sam_of(k) == want[k], ("stream", k)
        L.bsx_stream_close(s)
    finally:
        for c in chunks:
            L.bsx_sim_free_reads(c, 2 * n_pairs)
