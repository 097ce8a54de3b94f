"""-m gpu: the hg38-LIKE genome (profile 1 of csrc/host/sim.c: SINE/LINE/LTR-like families of up to a million copies, satellite arrays,
~43 % repeats) at the size bench.py reports its `hg38_like_genome` number on (3.1 Gbp, two FM indices of 6.2 G symbols built on the
device).  A module of its own: the clean genome of tests/test_gpu_hg38_scale.py and this one are never resident together (each is
~32 GB of index plus per-chunk buffers, and the builder's peak is ~125 GB)."""
import ctypes as C
import os
import numpy as np
import pytest
import samcheck
import simdata
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, Device, default_opt, SEED_DT, SA_DT
from test_gpu_hg38_scale import _make, _run_both, PacContig, MBP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hard():
    """the genome behind bench.py's `hg38_like_genome` line: profile 1 of csrc/host/sim.c (SINE/LINE/LTR-like families of up to a million
    copies, satellite arrays: ~43 % repeats), same seed, size and contigs"""
    g = _make(1)
    yield g
    g["dev"].close()
    g["idx"].close()


# ---- the hg38-like genome at the size the bench reports a number on: what is hot there (the HBM tiers walking over-represented
# intervals past max_occ, the second seeding pass, thousands of seeds per strand search, strand searches chained on the host)
def _sim(idx, n_pairs, seed, sub=0.005, pbat=0.1, truth=None):
    L = B.lib()
    L.bsx_sim_pairs_truth.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p), C.c_void_p]
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs_truth(idx.h, n_pairs, 150, seed, 200, 500, sub, pbat, C.byref(p), truth.ctypes.data_as(C.c_void_p) if truth is not None else None), "sim_pairs")
    return p


@pytest.mark.parametrize("max_occ", [500, 100])
def test_hg38_like_sam_identical(hard, max_occ, capfd):
    """20 k pairs against the 3.1 Gbp hg38-like genome, defaults and -c 100: SAM of the HIP pipeline == SAM of the CPU restatement byte
    for byte; the paths that are hot on this genome were really taken (strand searches seeded again with longer lists; with -c 100,
    over-represented intervals)."""
    import re
    L = B.lib()
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    n_pairs = 20000
    n = 2 * n_pairs
    p = _sim(hard["idx"], n_pairs, 31337)
    opt = default_opt()
    opt.n_threads = 16
    opt.flag |= 0x10 | 0x2
    opt.max_occ = max_occ
    B.tune("phases", "1")
    try:
        capfd.readouterr()
        hip, cpu = _run_both(hard, opt, p, n)
        err = capfd.readouterr().err
        bad = [i for i in range(n) if hip[i] != cpu[i]]
        assert not bad, "-c %d: SAM differs for %d reads, first: %r vs %r" % (max_occ, len(bad), hip[bad[0]][:400], cpu[bad[0]][:400])
        m = re.search(r"redo of (\d+) strand searches", err)
        assert m and int(m.group(1)) > 0, err[-600:]              # the second seeding pass ran
        m = re.search(r"left tier 1: (\d+), left tier 1b: (\d+)", err)
        assert m and int(m.group(2)) > 0, err[-600:]              # strand searches reached the HBM tiers
        mapped = np.mean([not (int(s.split(b"\t")[1]) & 4) for s in hip])
        assert mapped > 0.9, mapped
        # the first HBM tier in steps (tier2_export=1: its chains exported, extended ahead, the seed loop with its regions in HBM), the reads with
        # long region lists left to the host's de-duplication (long_dedup=0), mate rescue planned on the device (msw_plan=1): other routes, the same SAM
        L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
        r = C.cast(p, C.POINTER(B.Read))
        # (tier2_export=2: every seed of the exported main lists extended ahead; tier3_early=0: the last HBM tier behind the others instead of beside them;
        # msw_plan=1: mate rescue's plan pass and its K5 batch on the device, k_msw.hip)
        for name, val in (("tier2_export", "1"), ("tier2_export", "2"), ("long_dedup", "0"), ("msw_plan", "1"), ("tier3_early", "0")):
            B.tune(name, val)
            try:
                capfd.readouterr()
                B.check(L.bsx_process_seqs(hard["dev"].h, C.byref(opt), hard["idx"].h, 0, n, p, None), "process_seqs(%s)" % name)
                err2 = capfd.readouterr().err
                other = [C.string_at(r[i].sam) for i in range(n)]
                L.bsx_sim_reset_reads(p, n)
            finally:
                B.tune(name, None)
            assert other == hip, (name, val)
            if name == "tier2_export":
                assert "tier 2 (chains -> regions)" in err2, err2[-800:]
            if name == "tier3_early":
                assert "tier 3 beside them" not in err2 and "tier 3 beside them" in err, err2[-800:]
            if name == "msw_plan":
                assert "alignments planned and run on the device" in err2, err2[-800:]
    finally:
        B.tune("phases", None)
        L.bsx_sim_free_reads(p, n)


def test_long_region_lists_deduplicated_on_the_device(hard):
    """C5 for the reads a lane of k_dedup cannot hold (more than 32 regions: against this genome 15 % of the reads, with 60 % of a chunk's
    regions): bsx_regions_dedup2 (k_dedup_long, a wavefront per read -- ranks by counting, klib's own loop on one lane for lists with tied
    keys, the redundancy scan 64 earlier regions at a time) against mem_sort_deduplicate as the host runs it (region.c through
    bsx_hook_regs_sort_dedup), read by read, on the regions a regions batch left on the device."""
    L = B.lib()
    idx, dev = hard["idx"], hard["dev"]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    n_pairs = 30000
    n = 2 * n_pairs
    p = _sim(idx, n_pairs, 4242, sub=0.01, pbat=0.2)
    reads = C.cast(p, C.POINTER(B.Read))
    opt = default_opt()
    opt.flag |= 0x10 | 0x2
    try:
        seqs = [bytes(C.string_at(reads[i].seq, reads[i].l_seq)) for i in range(n)]
        buf = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
        offs = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64)
        tasks = np.zeros(2 * n, dtype=SEED_DT)
        for i in range(n):   # the reference's call order for -b 0 (bwamem.c:352-372): read 1 parent then daughter, read 2 daughter then parent
            for k, par in enumerate((1, 0) if i % 2 == 0 else (0, 1)):
                tasks[2 * i + k] = (offs[i], len(seqs[i]), par)
        dev.set_opt(opt)
        dev.set_reads(buf)
        regs, roff, rn = dev.regions(opt, tasks)
        cap, lcap = L.bsx_regions_dedup_cap(), L.bsx_regions_dedup_long_cap()
        out_n = np.zeros(n, dtype=np.int32)
        out_idx = np.zeros(n * cap, dtype=np.uint8)
        loff = np.zeros(n, dtype=np.int64)
        lidx, lc = C.c_void_p(), C.c_int64(0)
        L.bsx_regions_dedup2.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        B.check(L.bsx_regions_dedup2(dev.h, C.byref(opt), n, 2, out_n.ctypes.data_as(C.c_void_p), out_idx.ctypes.data_as(C.c_void_p), loff.ctypes.data_as(C.c_void_p),
                                     C.byref(lidx), C.byref(lc)), "bsx_regions_dedup2")
        pool = np.ctypeslib.as_array(C.cast(lidx, C.POINTER(C.c_uint16)), shape=(max(1, lc.value),)).copy() if lidx.value else np.zeros(1, dtype=np.uint16)
        L.bsx_hook_regs_sort_dedup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.bsx_hook_regs_sort_dedup.restype = C.c_int
        n_long = n_long_done = n_tied = biggest = dropped = n_over = 0
        for i in range(n):
            if rn[2 * i] < 0 or rn[2 * i + 1] < 0:
                assert out_n[i] == -1
                continue
            cat = np.concatenate([regs[roff[2 * i]:roff[2 * i] + rn[2 * i]], regs[roff[2 * i + 1]:roff[2 * i + 1] + rn[2 * i + 1]]])
            is_long = len(cat) > cap
            n_long += is_long
            n_over += len(cat) > lcap
            if not is_long:
                assert loff[i] == -1
            if out_n[i] < 0:
                continue
            keep = np.zeros(max(1, len(cat)), dtype=np.int32)
            m = L.bsx_hook_regs_sort_dedup(C.byref(opt), idx.h, cat.ctypes.data_as(C.c_void_p), len(cat), keep.ctypes.data_as(C.c_void_p))
            assert m >= 0, i
            if is_long:
                assert loff[i] >= 0
                got = pool[loff[i]:loff[i] + out_n[i]].astype(np.int32)
                n_long_done += 1
                biggest = max(biggest, len(cat))
                n_tied += len(set(cat["re"].tolist())) < len(cat)
                dropped += m < len(cat)
            else:
                got = out_idx[i * cap:i * cap + out_n[i]].astype(np.int32)
            assert out_n[i] == m and (got == keep[:m]).all(), (i, len(cat), list(got[:40]), list(keep[:m][:40]))
        # the long lists are there, most of them were finished on the device, lists with tied ends (klib's order of equal keys) among them
        # (what is not: lists beyond the kernel's tables, and reads where two regions have to be aligned across the gap between them -- a quarter of
        # the reads with a hundred regions have such a pair: the host's merge rounds, which batch those alignments)
        assert n_long > 0.05 * n and n_long_done > 0.6 * n_long and n_tied > 50 and biggest > 200 and dropped > 100, (n_long, n_long_done, n_over, n_tied, biggest, dropped)
        if lidx.value:
            from biscuit_amd.api import _libc_free
            _libc_free(lidx)
    finally:
        L.bsx_sim_free_reads(p, n)


def test_hg38_like_full_chunk_properties(hard):
    """One full chunk of the bench (-@ 16: 1 066 666 reads) on the bench's hg38-like genome: every record of a sample valid against the
    genome, reads found where they were simulated from, the same chunk twice gives the same SAM (checksum of checksums), and the pipelined
    stream gives it too."""
    import zlib
    L = B.lib()
    idx, dev = hard["idx"], hard["dev"]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_stream_open.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.bsx_stream_push.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.bsx_stream_flush.argtypes = [C.c_void_p]
    L.bsx_stream_close.argtypes = [C.c_void_p]
    L.bsx_stream_close.restype = None
    opt = default_opt()
    opt.n_threads = 16
    opt.flag |= 0x10 | 0x2
    n_pairs = (opt.chunk_size * 16) // 300
    n = 2 * n_pairs
    truth = np.zeros(2 * n_pairs, dtype=np.int64)
    p = _sim(idx, n_pairs, 900, pbat=0.0, truth=truth)
    r = C.cast(p, C.POINTER(B.Read))

    def crc():
        c = 0
        for i in range(n):
            c = zlib.crc32(C.string_at(r[i].sam), c)
        return c

    try:
        B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, n, p, None), "process_seqs")
        first = crc()
        genome = {nm: PacContig(hard["pac"], off, ln) for nm, off, ln in hard["contigs"]}
        ctg_off = {nm: off for nm, off, ln in hard["contigs"]}
        rng = np.random.default_rng(3)
        pick = rng.choice(n_pairs, 3000, replace=False)
        text = b"".join(C.string_at(r[int(i) * 2 + e].sam) for i in pick for e in (0, 1)).decode()
        hdr, recs = samcheck.parse_sam(text)
        for rec in recs:
            samcheck.check_record(rec, genome, 150)
        samcheck.check_pairs(recs)
        prim = [x for x in recs if not x["flag"] & 0x900]
        assert len(prim) == 6000
        assert np.mean([not x["flag"] & 4 for x in prim]) > 0.95
        ok = tot = 0
        for x in prim:
            if x["flag"] & 4 or x["mapq"] < 30:
                continue
            pi = int(x["qname"][1:])
            s, fl = int(truth[2 * pi]), int(truth[2 * pi + 1]) >> 1
            g = ctg_off[x["rname"]] + x["pos"] - 1
            tot += 1
            ok += s - 10 <= g <= s + fl + 10
        assert tot > 3000 and ok / tot > 0.99, (ok, tot)
        L.bsx_sim_reset_reads(p, n)
        B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, n, p, None), "process_seqs")
        assert crc() == first
        L.bsx_sim_reset_reads(p, n)
        s = C.c_void_p()
        B.check(L.bsx_stream_open(dev.h, C.byref(opt), idx.h, None, C.byref(s)), "stream_open")
        B.check(L.bsx_stream_push(s, 0, n, p), "push")
        B.check(L.bsx_stream_flush(s), "flush")
        L.bsx_stream_close(s)
        assert crc() == first
    finally:
        L.bsx_sim_free_reads(p, n)


def test_hg38_like_command_line_equals_end_to_end_oracle(hard, tmp_path_factory):
    """The command line against oracle/e2e.py (the independent end-to-end restatement over the reference's own kernels) from the index
    files of the hg38-like genome, 2 500 pairs: defaults and -c 100."""
    import shutil
    import e2e_cases as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libbiscuit_ref.so")):
        pytest.skip("oracle/_ref is absent")
    d = str(tmp_path_factory.mktemp("hg38_like_files"))
    need = int(hard["l_pac"] * 3.6) + (2 << 30)
    if shutil.disk_usage(d).free < need:
        pytest.skip("not enough disk for the index files (%d GB)" % (need >> 30))
    L = B.lib()
    idx = hard["idx"]
    idx.save(d + "/g")
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_sim_write_fastq.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    n_pairs = 2500
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 777, 200, 500, 0.008, 0.15, C.byref(p)), "sim_pairs")
    B.check(L.bsx_sim_write_fastq(p, 2 * n_pairs, (d + "/r1.fq").encode(), (d + "/r2.fq").encode(), 0), "write_fastq")
    L.bsx_sim_free_reads(p, 2 * n_pairs)
    try:
        for args in (["-@", "4", "g", "r1.fq", "r2.fq"], ["-@", "4", "-c", "100", "g", "r1.fq", "r2.fq"]):
            want = E.run_e2e(args, d)
            got = E.run_exe(os.path.join(root, "biscuit_amd", "biscuit_align"), args, d)
            assert got.count(b"\n") > n_pairs
            E.assert_same_sam(got, want, " ".join(args))
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))

