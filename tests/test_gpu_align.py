"""-m gpu: the whole `biscuit align` path.  SAM produced by the product (HIP kernels) must equal,
byte for byte (minus @PG), the SAM produced by the same host pipeline over the CPU restatement of
the kernels (oracle/).  Covers BASELINE.json's configs at sizes the CPU path finishes in seconds."""
import os
import subprocess
import numpy as np
import pytest
import simdata

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "biscuit_amd", "biscuit_align")
CPU = os.path.join(ROOT, "oracle", "oracle_align")


def run(exe, args, cwd, env=None, want_stderr=False):
    from biscuit_amd._lib import tune_env
    e = dict(os.environ)
    e.update(tune_env(env or {}))   # the library's settings (BSX_POS_CAP -> pos_cap ...) travel in one variable, $BSX_TUNE
    p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, env=e)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    sam = b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))
    return (sam, p.stderr.decode()) if want_stderr else sam


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    from biscuit_amd.api import Index
    d = str(tmp_path_factory.mktemp("align"))
    contigs = simdata.make_genome(1000000, seed=21, n_contigs=3)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    # config 1 shape: 2x100 directional pairs; plus a harder 2x150 set (indels, chimeras, PBAT-like, bad mates, Ns)
    p100 = simdata.make_pairs(contigs, 5000, 100, 1, frag=(180, 320), sub=0.005)
    p150 = simdata.make_pairs(contigs, 4000, 150, 2, sub=0.01, indel=0.006, pbat_frac=0.3, chimera_frac=0.06, bad_mate_frac=0.06, n_frac=0.03)
    for tag, pairs in (("a", p100), ("b", p150)):
        simdata.write_fastq(d + "/%s1.fq" % tag, [(n, a) for n, a, b in pairs])
        simdata.write_fastq(d + "/%s2.fq" % tag, [(n, b) for n, a, b in pairs])
    simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, 300, 1000, 5))
    # reads between the short kernels' 256 bases and a kilobase, mixed lengths in one chunk; and reads past the long kernels' tables
    mixed = []
    for k, ln in enumerate((300, 420, 600, 760, 900, 1024)):
        mixed += [("m%d_%s" % (ln, n), q) for n, q in simdata.make_single(contigs, 120, ln, 40 + k)]
    simdata.write_fastq(d + "/mixed.fq", mixed)
    simdata.write_fastq(d + "/xlong.fq", simdata.make_single(contigs, 60, 1500, 9) + simdata.make_single(contigs, 200, 150, 10))
    return d


CASES = [
    ("pe100_default", ["-@", "4", "g", "a1.fq", "a2.fq"]),
    ("pe150_default", ["-@", "4", "g", "b1.fq", "b2.fq"]),
    ("pe150_directional", ["-@", "4", "-b", "1", "g", "b1.fq", "b2.fq"]),
    ("pe150_all_softclip", ["-@", "4", "-a", "-Y", "g", "b1.fq", "b2.fq"]),
    ("pe150_norescue_nopair", ["-@", "4", "-S", "-P", "g", "b1.fq", "b2.fq"]),
    ("pe150_fixed_isize_rg", ["-@", "4", "-I", "350,60", "-R", "@RG\\tID:x\\tSM:y", "-C", "g", "b1.fq", "b2.fq"]),
    ("se150_default", ["-@", "4", "g", "b1.fq"]),
    ("se150_daughter", ["-@", "4", "-b", "3", "g", "b2.fq"]),
    ("se150_clip", ["-@", "4", "-J", "AGATCGGAAGAGC", "-z", "10", "-5", "2", "-3", "1", "g", "b1.fq"]),
    ("se150_scoring", ["-@", "4", "-A", "2", "-B", "3", "-O", "5,7", "-E", "2,1", "-L", "4,6", "-T", "40", "-k", "17", "-w", "60", "g", "b1.fq"]),
    ("long_1kb", ["-@", "4", "g", "long.fq"]),
    ("long_mixed_lengths", ["-@", "4", "g", "mixed.fq"]),
    ("long_1kb_scoring", ["-@", "4", "-A", "2", "-B", "5", "-O", "7,8", "-k", "17", "-w", "80", "g", "long.fq"]),
    ("short_with_seed_sw_filter", ["-@", "4", "-W", "5", "g", "b1.fq", "b2.fq"]),   # a small -W turns mem_flt_chained_seeds on for 150 bp reads
    ("longer_than_device_tables", ["-@", "4", "g", "xlong.fq"]),
    # options the device chaining/extension pass reads: occurrence cap, chain filter knobs, strand restriction, band, clip penalties
    ("pe150_chain_knobs", ["-@", "4", "-c", "12", "-D", "0.3", "-W", "25", "-m", "30", "-G", "4000", "g", "b1.fq", "b2.fq"]),
    ("se150_bsstrand_band", ["-@", "4", "-f", "1", "-w", "12", "-L", "0,9", "-r", "1.2", "-y", "30", "g", "b1.fq"]),
    # scores beyond the packed 16-bit seed filter's range (199 columns x 170 > 2^15): launch_seedsw leaves its job list empty and every
    # seed of the filter is aligned in 32-bit arithmetic
    ("long_1kb_big_scores", ["-@", "4", "-A", "170", "g", "long.fq"]),
]


@pytest.mark.parametrize("name,args", CASES, ids=[c[0] for c in CASES])
def test_sam_identical(data, name, args):
    want = run(CPU, args, data)
    got = run(HIP, args, data)
    assert got.count(b"\n") > 100
    if got != want:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(gl, wl)):
            assert a == b, "first difference at line %d:\nHIP: %s\nCPU: %s" % (i, a[:600], b[:600])
        assert len(gl) == len(wl)


def test_occurrence_table_overflow(data):
    """strand searches that find no room in the chunk-wide occurrence table (k_occ_expand) walk the suffix array inline
    instead; the slots they reserved must not be read as ranks.  Same SAM as with room for everything."""
    args = CASES[1][1]
    want = run(HIP, args, data)
    for cap in ("0", "5000", "60000"):
        assert run(HIP, args, data, env={"BSX_POS_CAP": cap}) == want, cap


def test_seed_filter_job_list_overflow(data):
    """the seed filter's alignments go through a chunk-wide job list (k_seedsw_prep -> k_swl16 -> k_seedsw_apply); seeds that find no
    room in it are aligned a wavefront at a time by the last of the three.  Same SAM with no room, little room and room for all."""
    for ci in (10, 13):
        args = CASES[ci][1]
        want = run(CPU, args, data)
        for cap in (None, "8", "3000"):
            assert run(HIP, args, data, env={"BSX_SSW_CAP": cap} if cap else None) == want, (CASES[ci][0], cap)


def _on_device(stderr):
    import re
    m = re.findall(r"\[M::regions\] on device (\d+) \| declined: (.*)", stderr)
    assert m, stderr[-1500:]
    on = sum(int(a) for a, _ in m)
    off = sum(int(x) for _, rest in m for x in re.findall(r" (\d+)", rest))
    return on, off


@pytest.mark.parametrize("name,args", [CASES[1], CASES[3], CASES[9], CASES[10], CASES[11], CASES[12], CASES[13], CASES[15], CASES[16]], ids=lambda c: c if isinstance(c, str) else "")
def test_device_regions_equal_host_chaining(data, name, args):
    """k_regions (SA lookup + chaining + chain filter + chain-to-region on the device) against the same
    strand searches chained on the host through the batch kernels: identical SAM, and the device pass
    really took the bulk of the work."""
    dev, err = run(HIP, args, data, env={"BSX_PHASES": "1"}, want_stderr=True)
    host = run(HIP, args, data, env={"BSX_HOST_CHAIN": "1"})
    assert dev == host
    on, off = _on_device(err)
    assert on > off, (on, off)
    # the sequence in which every tier exports its chains (and the seed-SW filter runs ahead of chains -> regions) is for chunks with
    # long reads or an active filter only: ordinary chunks keep the faster one
    special = name.startswith("long") or "filter" in name
    assert ("every tier exports: 1" in err) == special and ("every tier exports: 0" in err) == (not special), name


def test_device_regions_equal_host_chaining_repeat_rich(tmp_path):
    """Same A/B at the bench's workload shape (synthetic genome with repeat families, 2x150 pairs, -b 0):
    60k pairs through bsx_process_seqs with the device regions pass and with host chaining; every read's
    SAM text must be identical.  This is the data where chains per strand search run into the dozens."""
    import ctypes as C
    import zlib
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    L = B.lib()
    d = str(tmp_path)
    B.check(L.bsx_sim_genome((d + "/g.fa").encode(), C.c_int64(8000000), C.c_uint64(77), 5, C.c_double(0.08)), "sim_genome")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "index_build")
    idx = Index(d + "/g")
    dev = Device(0)
    dev.upload_index(idx)
    opt = default_opt()
    opt.n_threads = 4
    opt.flag |= 0x10 | 0x2
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    n_pairs = 60000
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 5, 200, 500, 0.01, 0.2, C.byref(p)), "sim_pairs")
    reads = C.cast(p, C.POINTER(B.Read))

    def crc():
        c = 0
        for i in range(2 * n_pairs):
            c = zlib.crc32(C.string_at(reads[i].sam), c)
        return c

    try:
        seen = set()
        for max_occ in (500, 8):   # 8: most repeat seeds are over-represented, the first-max_occ visiting rule (memchain.c:325-326) runs on the device
            opt.max_occ = max_occ
            B.tune("host_chain", None)
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "process_seqs")
            ps = B.PhaseStats()
            L.bsx_last_phase_stats(C.byref(ps))
            assert ps.n_host_tasks * 5 < ps.n_tasks, (max_occ, ps.n_host_tasks, ps.n_tasks)
            # with max_occ = 8 hundreds of strand searches have an over-represented interval that has to be walked past its first 8 occurrences
            # (memchain.c:325-326): the HBM tiers do that themselves (533 of 240 000 were left to the host before they did, 27 since)
            assert ps.n_host_tasks < 200, (max_occ, ps.n_host_tasks)
            a = crc()
            L.bsx_sim_reset_reads(p, 2 * n_pairs)
            B.tune("host_chain", "1")
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "process_seqs")
            L.bsx_last_phase_stats(C.byref(ps))
            assert ps.n_host_tasks == ps.n_tasks
            assert crc() == a, max_occ
            L.bsx_sim_reset_reads(p, 2 * n_pairs)
            B.tune("host_chain", None)
            # strand searches whose interval lists overflow are seeded again on a side stream with longer lists; a short
            # first-pass list sends ordinary reads down that path, collected before the front half returns or (async) by
            # regions_finish at the start of the back half
            # -- or, when there are many of them (round 5), seeded again inside the chunk's one launch sequence (the setting redo_merge_min: from how many)
            B.tune("seed_mem_cap", "28")
            for merge_min, asy in (("1000000000", "0"), ("1000000000", "1"), ("1", "0")):
                B.tune("async_redo", asy)
                B.tune("redo_merge_min", merge_min)
                B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "process_seqs")
                L.bsx_last_phase_stats(C.byref(ps))
                assert crc() == a, (max_occ, merge_min, asy)
                assert ps.n_host_tasks * 5 < ps.n_tasks
                if asy == "1":
                    assert ps.n_redo_tasks > 0
                if merge_min == "1":
                    assert ps.n_redo_tasks == 0
                L.bsx_sim_reset_reads(p, 2 * n_pairs)
            B.tune("seed_mem_cap", None)
            B.tune("async_redo", None)
            B.tune("redo_merge_min", None)
            seen.add(a)
        assert len(seen) == 2   # the cap does change the alignments
    finally:
        for k in ("host_chain", "seed_mem_cap", "async_redo", "redo_merge_min"):
            B.tune(k, None)
        L.bsx_sim_free_reads(p, 2 * n_pairs)
        dev.close()
        idx.close()


def test_chunk_stream_equals_chunk_by_chunk_on_device(tmp_path):
    """bsx_stream_* (device lanes, priority streams, front halves of later chunks overlapping an older chunk's back
    half) against bsx_process_seqs chunk by chunk: identical SAM text for every chunk at depths 2 and 3."""
    import ctypes as C
    import zlib
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    L = B.lib()
    d = str(tmp_path)
    B.check(L.bsx_sim_genome((d + "/g.fa").encode(), C.c_int64(4000000), C.c_uint64(78), 4, C.c_double(0.08)), "sim_genome")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "index_build")
    idx = Index(d + "/g")
    dev = Device(0)
    dev.upload_index(idx)
    opt = default_opt()
    opt.n_threads = 4
    opt.flag |= 0x10 | 0x2
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_stream_open.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.bsx_stream_push.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.bsx_stream_flush.argtypes = [C.c_void_p]
    L.bsx_stream_close.argtypes = [C.c_void_p]
    L.bsx_stream_close.restype = None
    n_pairs, n_chunks = 20000, 5
    chunks = []
    for k in range(n_chunks):
        p = C.c_void_p()
        B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 300 + k, 200, 500, 0.01, 0.2, C.byref(p)), "sim_pairs")
        chunks.append(p)

    def crc(k):
        r = C.cast(chunks[k], C.POINTER(B.Read))
        c = 0
        for i in range(2 * n_pairs):
            c = zlib.crc32(C.string_at(r[i].sam), c)
        return c

    try:
        want = []
        for k in range(n_chunks):
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 2 * n_pairs * k, 2 * n_pairs, chunks[k], None), "process_seqs")
            want.append(crc(k))
            L.bsx_sim_reset_reads(chunks[k], 2 * n_pairs)
        assert len(set(want)) == n_chunks
        for depth in ("2", "3"):
            os.environ["BSX_STREAM_DEPTH"] = depth
            s = C.c_void_p()
            B.check(L.bsx_stream_open(dev.h, C.byref(opt), idx.h, None, C.byref(s)), "stream_open")
            for k in range(n_chunks):
                B.check(L.bsx_stream_push(s, 2 * n_pairs * k, 2 * n_pairs, chunks[k]), "push")
            B.check(L.bsx_stream_flush(s), "flush")
            L.bsx_stream_close(s)
            for k in range(n_chunks):
                assert crc(k) == want[k], (depth, k)
                L.bsx_sim_reset_reads(chunks[k], 2 * n_pairs)
    finally:
        os.environ.pop("BSX_STREAM_DEPTH", None)
        for c in chunks:
            L.bsx_sim_free_reads(c, 2 * n_pairs)
        dev.close()
        idx.close()


def test_cli_many_chunks_through_the_stream(data):
    """The CLI drives chunks through bsx_stream_*: with a small chunk size the run is dozens of chunks, several in
    flight at once; the SAM must still equal the CPU driver's (same chunking, so the same per-chunk insert-size
    statistics) and the one-chunk-at-a-time product path's."""
    env = {"BSX_CHUNK_SIZE": "30000"}
    args = ["-@", "4", "g", "b1.fq", "b2.fq"]
    want = run(CPU, args, data, env=env)
    got = run(HIP, args, data, env=env)
    sync = run(HIP, args, data, env=dict(env, BSX_NO_STREAM="1"))
    assert got.count(b"\n") > 8000
    assert got == want
    assert sync == want


@pytest.mark.parametrize("env", [
    {"BSX_DEVICE_SA_INTV": "32"},                                   # the files' suffix-array sample instead of the denser device one
    {"BSX_DEVICE_SA_INTV": "1"},                                    # every rank sampled: K3 is a plain load
    {"BSX_REGIONS_OCC": "3", "BSX_MID_QUOTA": "1", "BSX_C2R_QUOTA": "2"},   # another register-allocation target, short-lived region workgroups
    {"BSX_RESERVE_CU_EVERY": "0", "BSX_STREAM_DEPTH": "1"},         # no reserved CUs, no overlap of chunks
    {"BSX_SEED_QUOTA": "0", "BSX_REGIONS_QUOTA": "1", "BSX_STREAM_DEPTH": "4"},   # persistent seeding waves, one task per region wave
    {"BSX_REGIONS_MID": "0", "BSX_SEED_TRIP_BUDGET": "200"},        # no LDS tier between the first and the HBM tiers; most strand searches handed to the second seeding pass
    {"BSX_HOST_DEDUP": "1"},                                        # C5 (mem_sort_deduplicate) of every read on the host instead of by k_dedup
    {"msw_plan": "1", "back_slices": "3"},                          # mate rescue's plan pass by k_msw_plan on the device, its K5 batch from device memory
    {"tier2_export": "1", "long_dedup": "0", "back_slices": "1", "back_threads": "1"},   # the first HBM tier in steps; long lists de-duplicated on the host; the back half unsliced
    {"tier2_export": "2", "tier3_early": "0", "tier3_order": "0", "tier3_wgs": "1", "seed_budget2": "1"},   # ... with every seed extended ahead; the last HBM tier behind the others, in list order; a short second seeding pass
    {"reserve_cu_every": "8", "BSX_STREAM_DEPTH": "4"},             # CU-masked front-half streams (a CU of every shader engine left out), the small copies through the unmasked stream
    {"reserve_cu_every": "4", "small_copies_unmasked": "0"},
], ids=["sa32", "sa1", "occ", "noreserve_depth1", "quotas_depth4", "nomid_budget", "host_dedup", "msw_plan", "tier2x_nolongdedup_noslices", "tier2x_all_seeds_tier3_late", "cu_mask8", "cu_mask4"])
def test_device_tuning_knobs_do_not_change_the_output(data, env):
    """Launch shapes, occupancy targets, the device-side suffix-array sample and the pipeline depth are performance knobs:
    the SAM must be byte-identical whatever they are set to."""
    base_env = {"BSX_CHUNK_SIZE": "60000"}
    args = ["-@", "4", "g", "b1.fq", "b2.fq"]
    want = run(HIP, args, data, env=base_env)
    got = run(HIP, args, data, env=dict(base_env, **env))
    assert got == want and got.count(b"\n") > 8000


def test_sim_chunk_equals_cpu_restatement(tmp_path):
    """A repeat-rich simulated chunk (the bench's generator: repeat families, tandem repeats) through the product
    (device seeding, K3, chaining, extension in all three region tiers) and through the CPU restatement of the kernels with
    the host's own chaining (oracle/): every read's SAM text must be identical.  Two settings of max_occ, so that the
    over-represented-seed rule is compared against the host implementation as well."""
    import ctypes as C
    import zlib
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    from oracle_lib import Port
    L = B.lib()
    d = str(tmp_path)
    B.check(L.bsx_sim_genome((d + "/g.fa").encode(), C.c_int64(6000000), C.c_uint64(91), 6, C.c_double(0.10)), "sim_genome")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "index_build")
    idx = Index(d + "/g")
    dev = Device(0)
    dev.upload_index(idx)
    port = Port(idx, 16)
    be = port.backend()
    opt = default_opt()
    opt.n_threads = 4
    opt.flag |= 0x10 | 0x2
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    n_pairs = 40000
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 17, 200, 500, 0.01, 0.2, C.byref(p)), "sim_pairs")
    reads = C.cast(p, C.POINTER(B.Read))

    def sams():
        return [C.string_at(reads[i].sam) for i in range(2 * n_pairs)]

    os.environ["BSX_HOST_THREADS"] = "16"
    try:
        for max_occ in (500, 10):
            opt.max_occ = max_occ
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "process_seqs")
            ps = B.PhaseStats()
            L.bsx_last_phase_stats(C.byref(ps))
            assert ps.n_host_tasks * 10 < ps.n_tasks, (max_occ, ps.n_host_tasks, ps.n_tasks)
            got = sams()
            L.bsx_sim_reset_reads(p, 2 * n_pairs)
            B.check(L.bsx_process_seqs_backend(C.byref(be), C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "cpu restatement")
            want = sams()
            L.bsx_sim_reset_reads(p, 2 * n_pairs)
            bad = [i for i in range(2 * n_pairs) if got[i] != want[i]]
            assert not bad, "max_occ %d: %d reads differ, first:\nHIP: %s\nCPU: %s" % (max_occ, len(bad), got[bad[0]][:400], want[bad[0]][:400])
    finally:
        os.environ.pop("BSX_HOST_THREADS", None)
        L.bsx_sim_free_reads(p, 2 * n_pairs)
        dev.close()
        idx.close()


def test_extensions_ahead_equal_inline(data):
    """chains -> regions with the extensions of every chain's best seed made ahead of the seed loop, four to a wavefront (k_ext4, the
    default), against all extensions inline in the wavefront-per-strand-search loop (BSX_X4=0): identical SAM."""
    for name, args in (CASES[0], CASES[1], CASES[8], CASES[9], CASES[11]):
        want = run(HIP, args, data, env={"BSX_X4": "0"})
        assert run(HIP, args, data) == want, name


def test_dedup_kernel_against_host_function(tmp_path):
    """C5 directly: bsx_regions_dedup (k_dedup, a lane per read) on the regions a regions batch left on the device, against
    mem_sort_deduplicate as the host runs it (csrc/host/region.c, through bsx_hook_regs_sort_dedup) on the same regions.
    Repeat-rich genome, non-directional search (two strand searches per read): reads with up to a dozen regions, ties in
    end position and in score."""
    import ctypes as C
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    L = B.lib()
    d = str(tmp_path)
    B.check(L.bsx_sim_genome((d + "/g.fa").encode(), C.c_int64(4000000), C.c_uint64(77), 4, C.c_double(0.15)), "sim_genome")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "index_build")
    idx = Index(d + "/g")
    dev = Device(0)
    dev.upload_index(idx)
    opt = default_opt()
    opt.flag |= 0x10 | 0x2
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    n_pairs = 20000
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 5, 200, 500, 0.01, 0.2, C.byref(p)), "sim_pairs")
    reads = C.cast(p, C.POINTER(B.Read))
    n = 2 * n_pairs
    seqs = [bytes(C.string_at(reads[i].seq, reads[i].l_seq)) for i in range(n)]
    buf = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    from biscuit_amd.api import SEED_DT
    tasks = np.zeros(2 * n, dtype=SEED_DT)
    for i in range(n):   # the reference's call order for -b 0: parent (C>T) then daughter (G>A) search of each read
        for k, par in enumerate((1, 0)):
            tasks[2 * i + k] = (offs[i], len(seqs[i]), par)
    dev.set_opt(opt)
    dev.set_reads(buf)
    regs, roff, rn = dev.regions(opt, tasks)
    cap = L.bsx_regions_dedup_cap()
    out_n = np.zeros(n, dtype=np.int32)
    out_idx = np.zeros(n * cap, dtype=np.uint8)
    L.bsx_regions_dedup.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    B.check(L.bsx_regions_dedup(dev.h, C.byref(opt), n, 2, out_n.ctypes.data_as(C.c_void_p), out_idx.ctypes.data_as(C.c_void_p)), "bsx_regions_dedup")
    L.bsx_hook_regs_sort_dedup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.bsx_hook_regs_sort_dedup.restype = C.c_int
    done = multi = dropped = left = 0
    for i in range(n):
        if rn[2 * i] < 0 or rn[2 * i + 1] < 0:
            assert out_n[i] == -1
            continue
        cat = np.concatenate([regs[roff[2 * i]:roff[2 * i] + rn[2 * i]], regs[roff[2 * i + 1]:roff[2 * i + 1] + rn[2 * i + 1]]])
        if out_n[i] < 0:
            left += 1   # more regions than the kernel holds, or a pair of regions to test for concatenation: the host's rounds
            continue
        keep = np.zeros(max(1, len(cat)), dtype=np.int32)
        m = L.bsx_hook_regs_sort_dedup(C.byref(opt), idx.h, cat.ctypes.data_as(C.c_void_p), len(cat), keep.ctypes.data_as(C.c_void_p))
        assert m >= 0, i    # the host function needed no concatenation score either
        got = out_idx[i * cap:i * cap + out_n[i]]
        assert out_n[i] == m and (got == keep[:m]).all(), (i, len(cat), list(got), list(keep[:m]))
        done += 1
        multi += len(cat) > 2
        dropped += m < len(cat)
    assert done > 0.95 * n and multi > 1000 and dropped > 1000 and left < 0.02 * n, (done, multi, dropped, left)
    L.bsx_sim_free_reads(p, n)
    dev.close()
    idx.close()


def test_failed_output_ends_cleanly_with_chunks_in_flight(data):
    """A write to stdout that fails (ENOSPC) while several chunks are in the pipeline: the front halves in flight are joined and the
    stream is closed before their reads are released, and the command exits with status 1 and the reference's message -- not with a
    signal (the streamed loop of csrc/host/cli.c; utils.c:214-240 for the reference's behaviour)."""
    args = ["-@", "4", "g", "b1.fq", "b2.fq"]
    e = dict(os.environ, BSX_CHUNK_SIZE="20000")     # ~15 chunks, four in flight
    ok = subprocess.run([HIP] + args, cwd=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=e)
    assert ok.returncode == 0 and ok.stdout.count(b"\n") > 8000
    for _ in range(3):
        with open("/dev/full", "wb") as full:
            bad = subprocess.run([HIP] + args, cwd=data, stdout=full, stderr=subprocess.PIPE, timeout=600, env=e)
        assert bad.returncode == 1, (bad.returncode, bad.stderr.decode()[-800:])
        assert b"failed to write" in bad.stderr
