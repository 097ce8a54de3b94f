"""-m gpu: size-independent properties at the bench's chunk size (BASELINE configs[1] shape: 2x150 bp, -@16 = 533,333
pairs per chunk), where the CPU checker would take minutes: the pipelined stream must reproduce the chunk-by-chunk
output exactly, a second run must reproduce the first, and the records themselves must be consistent with the genome
(CIGAR/MD/NM/ZC recomputed from the reference, mate fields, nearly everything mapped where it was simulated from)."""
import ctypes as C
import os
import zlib
import numpy as np
import pytest
import samcheck

pytestmark = pytest.mark.gpu


def test_full_chunk_properties(tmp_path):
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    L = B.lib()
    d = str(tmp_path)
    B.check(L.bsx_sim_genome((d + "/g.fa").encode(), C.c_int64(24000000), C.c_uint64(5), 6, C.c_double(0.05)), "sim_genome")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "index_build")
    idx = Index(d + "/g")
    dev = Device(0)
    dev.upload_index(idx)
    opt = default_opt()
    opt.n_threads = 16
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
    n_pairs = (opt.chunk_size * 16) // 300
    n = 2 * n_pairs
    chunks = []
    for k in range(3):
        p = C.c_void_p()
        B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 900 + k, 200, 500, 0.005, 0.0, C.byref(p)), "sim_pairs")
        chunks.append(p)

    def crc(k):
        r = C.cast(chunks[k], C.POINTER(B.Read))
        c = 0
        for i in range(n):
            c = zlib.crc32(C.string_at(r[i].sam), c)
        return c

    try:
        # chunk by chunk, twice: idempotence
        first = []
        for k in range(3):
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n * k, n, chunks[k], None), "process_seqs")
            first.append(crc(k))
        # validity of a sample of the records of chunk 0 against the genome
        genome = samcheck.load_genome(d + "/g.fa")
        r = C.cast(chunks[0], C.POINTER(B.Read))
        rng = np.random.default_rng(3)
        text = b"".join(C.string_at(r[int(i) * 2 + e].sam) for i in rng.choice(n_pairs, 4000, replace=False) for e in (0, 1)).decode()
        hdr, recs = samcheck.parse_sam(text)
        for rec in recs:
            samcheck.check_record(rec, genome, 150)
        samcheck.check_pairs(recs)
        prim = [x for x in recs if not x["flag"] & 0x900]
        assert len(prim) == 8000
        assert np.mean([not x["flag"] & 4 for x in prim]) > 0.97
        assert np.mean([bool(x["flag"] & 2) for x in prim]) > 0.9
        for k in range(3):
            L.bsx_sim_reset_reads(chunks[k], n)
        B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, n, chunks[0], None), "process_seqs")
        assert crc(0) == first[0]
        L.bsx_sim_reset_reads(chunks[0], n)
        # the pipelined stream
        s = C.c_void_p()
        B.check(L.bsx_stream_open(dev.h, C.byref(opt), idx.h, None, C.byref(s)), "stream_open")
        for k in range(3):
            B.check(L.bsx_stream_push(s, n * k, n, chunks[k]), "push")
        B.check(L.bsx_stream_flush(s), "flush")
        L.bsx_stream_close(s)
        assert [crc(k) for k in range(3)] == first
    finally:
        for c in chunks:
            L.bsx_sim_free_reads(c, n)
        dev.close()
        idx.close()
