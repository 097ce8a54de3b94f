"""CPU: the oracle (oracle/port.c) and the host primitives against tests/golden/ref_vectors.npz --
inputs and outputs recorded from the REAL reference functions (tests/golden/make_vectors.py).
Bit-exact.  This is what pins the checker on machines without /root/reference."""
import ctypes as C
import os
import numpy as np
import pytest
import oracle_lib
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, default_opt

HERE = os.path.dirname(os.path.abspath(__file__))
u8p, i8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_uint64)


def P(a, t):
    return a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def V():
    return np.load(os.path.join(HERE, "golden", "ref_vectors.npz"))


@pytest.fixture(scope="module")
def PL():
    return oracle_lib.port_lib()


def _mat(V, a, b, which):
    key = {(1, 2): 0, (2, 3): 1, (1, 9): 2}.get((a, b))
    if key is not None:
        return np.ascontiguousarray(V["scmat"][key][which])
    o = default_opt()
    o.a, o.b = a, b
    B.lib().bsx_opt_fill_matrices(C.byref(o))
    return np.array([o.mat, o.ctmat, o.gamat][which], dtype=np.int8)


def test_scoring_matrices_and_defaults(V):
    L = B.lib()
    for k, (a, b) in enumerate(((1, 2), (2, 3), (1, 9))):
        o = default_opt()
        o.a, o.b = a, b
        L.bsx_opt_fill_matrices(C.byref(o))
        got = np.stack([np.array(o.mat, np.int8), np.array(o.ctmat, np.int8), np.array(o.gamat, np.int8)])
        assert (got == V["scmat"][k]).all()
    buf = C.create_string_buffer(4096)
    L.bsx_hook_opt_defaults(buf, 4096)
    assert buf.value == V["opt_defaults"].tobytes()


def test_hash_and_mapq(V):
    L = B.lib()
    L.bsx_hook_hash64.restype = C.c_uint64
    L.bsx_hook_hash64.argtypes = [C.c_uint64]
    for k, h in zip(V["hash_in"], V["hash_out"]):
        assert L.bsx_hook_hash64(int(k)) == int(h)
    L.bsx_hook_mapq.argtypes = [C.c_int] * 3 + [C.c_float, C.c_int] + [C.c_int] * 6 + [C.c_int64, C.c_int64, C.c_int, C.c_float]
    for r in V["mapq"]:
        score, sub, csub, sub_n, qb, qe, rb, re, seedcov = [int(x) for x in r[:9]]
        q = L.bsx_hook_mapq(1, 2, 19, 50.0, 3, score, sub, csub, sub_n, qb, qe, rb, re, seedcov, float(np.float32(r[9])))
        assert q == int(r[10]), r


def test_introsort_permutation(V):
    """the exact permutation of klib's unstable introsort (incl. comb-sort fallback), asc and desc: the generic form (a comparison callback) and
    the one compiled per element type (sort_tmpl.h: what the back half's sorts run)"""
    L = B.lib()
    off = V["sort_off"]
    for i in range(len(off) - 1):
        keys = V["sort_keys"][off[i]:off[i + 1]]
        n = len(keys)
        for desc, want in ((0, V["sort_perm_asc"]), (1, V["sort_perm_desc"])):
            for fn in (L.bsx_hook_sort_kv, L.bsx_hook_sort_kv_typed):
                kv = np.stack([keys, np.arange(n, dtype=np.int64)], 1).copy()
                fn(C.c_int64(n), kv.ctypes.data_as(C.c_void_p), desc)
                assert (kv[:, 1] == want[off[i]:off[i + 1]]).all(), (i, n, desc)


def test_btree_lookup_and_order(V):
    """kb_intervalp's `lower` and the in-order traversal, duplicates included (t = 3 nodes)"""
    L = B.lib()
    L.bsx_bt_new.restype = C.c_void_p
    L.bsx_bt_put.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.bsx_bt_lower.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_bt_traverse.argtypes = [C.c_void_p, C.c_void_p]
    L.bsx_bt_free.argtypes = [C.c_void_p]
    oo, ro = V["bt_ops_off"], V["bt_res_off"]
    for c in range(len(oo) - 1):
        ops = V["bt_ops"][oo[c]:oo[c + 1]].reshape(-1, 2)
        res = V["bt_res"][ro[c]:ro[c + 1]]
        t = C.c_void_p(L.bsx_bt_new())
        nid = 0
        k = 0
        for op, x in ops:
            if op == 0:
                L.bsx_bt_put(t, int(x), nid); assert res[k] == nid; nid += 1; k += 1
            elif op == 1:
                assert L.bsx_bt_lower(t, int(x)) == int(res[k]); k += 1
            else:
                ids = np.zeros(nid + 1, np.int32)
                n = L.bsx_bt_traverse(t, ids.ctypes.data_as(C.c_void_p))
                assert n == int(x) and (ids[:n] == res[k:k + n]).all()
        L.bsx_bt_free(t)


def test_extend_sw_global_vs_reference_vectors(V, PL):
    qo, to = V["ext_qoff"], V["ext_toff"]
    for i, (par, want) in enumerate(zip(V["ext_par"], V["ext_out"])):
        a, b, which, od, ed, oi, ei, w, eb, zd, h0 = [int(x) for x in par]
        q = np.ascontiguousarray(V["ext_q"][qo[i]:qo[i + 1]]); t = np.ascontiguousarray(V["ext_t"][to[i]:to[i + 1]])
        M = _mat(V, a, b, which)
        o = (C.c_int * 6)()
        PL.oracle_extend1(len(q), P(q, u8p), len(t), P(t, u8p), P(M, i8p), od, ed, oi, ei, w, eb, zd, h0, o)
        assert list(o) == list(want), i
    qo, to = V["sw_qoff"], V["sw_toff"]
    for i, (par, want) in enumerate(zip(V["sw_par"], V["sw_out"])):
        a, b, which, od, ed, oi, ei, xtra = [int(x) for x in par]
        q = V["sw_q"][qo[i]:qo[i + 1]].copy(); t = V["sw_t"][to[i]:to[i + 1]].copy()
        M = _mat(V, a, b, which)
        o = (C.c_int * 7)()
        PL.oracle_sw1(len(q), P(q, u8p), len(t), P(t, u8p), P(M, i8p), od, ed, oi, ei, xtra, o)
        assert list(o) == list(want), (i, hex(xtra))
    qo, to, co = V["gl_qoff"], V["gl_toff"], V["gl_coff"]
    for i, par in enumerate(V["gl_par"]):
        a, b, which, od, ed, oi, ei, w, wc = [int(x) for x in par]
        q = np.ascontiguousarray(V["gl_q"][qo[i]:qo[i + 1]]); t = np.ascontiguousarray(V["gl_t"][to[i]:to[i + 1]])
        M = _mat(V, a, b, which)
        cg = (C.c_uint32 * 2048)(); n = C.c_int()
        s = PL.oracle_global1(len(q), P(q, u8p), len(t), P(t, u8p), P(M, i8p), od, ed, oi, ei, w, wc, C.byref(n), cg, 2048)
        assert s == int(V["gl_score"][i]) and list(cg[:n.value]) == list(V["gl_cigar"][co[i]:co[i + 1]]), i


@pytest.fixture(scope="module")
def g24k(tmp_path_factory):
    d = tmp_path_factory.mktemp("g24k")
    return Index.build(os.path.join(HERE, "golden", "g24k.fa"), str(d / "g"))


def test_fm_index_vs_reference_vectors(V, PL, g24k):
    """index built by the repo's builder from the committed FASTA; SMEM / LAST-like seeds / occ / SA equal
    what the reference's bwt.c returned on it"""
    PL.oracle_sa.restype = C.c_uint64
    PL.oracle_sa.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    ro, so = V["fm_roff"], V["fm_soff"]
    for i, par in enumerate(V["fm_par"]):
        p, x, mi, n1, r1, q1 = [int(v) for v in par]
        rd = np.ascontiguousarray(V["fm_reads"][ro[i]:ro[i + 1]])
        out = np.zeros(4 * 512, np.uint64); ret = C.c_int()
        n = PL.oracle_smem1(g24k.h, p, len(rd), P(rd, u8p), x, mi, P(out, u64p), 512, C.byref(ret))
        assert n == n1 and ret.value == r1 and (out[:4 * n] == V["fm_smem"][so[i]:so[i + 1]]).all(), i
        a = np.zeros(4, np.uint64)
        q = PL.oracle_seed_strategy1(g24k.h, p, len(rd), P(rd, u8p), x, 19, 20, P(a, u64p))
        assert q == q1 and (a == V["fm_ss1"][i]).all(), i
    ks = V["fm_k"]
    for p in (0, 1):
        for i, k in enumerate(ks):
            c = np.zeros(4, np.uint64)
            PL.oracle_occ4(g24k.h, p, C.c_uint64(int(k)), P(c, u64p))
            assert (c == V["fm_occ4_%d" % p][i]).all(), (p, k)
        got = np.array([PL.oracle_sa(g24k.h, p, int(k)) for k in ks[ks >= 1]], dtype=np.uint64)
        assert (got == V["fm_sa_%d" % p]).all()


def _ragged(V, key, off):
    o = V[off]
    return [V[key][o[i]:o[i + 1]] for i in range(len(o) - 1)]


def test_three_pass_seeding_vs_reference_vectors(V, g24k):
    """K1+K2 as mem_collect_intv composes them (memchain.c:50-106): the CPU restatement's interval lists == the lists
    obtained by driving the real bwt_smem1a / bwt_seed_strategy1 (tests/golden/make_vectors.py), read by read."""
    import simdata
    from biscuit_amd.api import SEED_DT
    port = oracle_lib.Port(g24k, n_threads=2)
    reads = _ragged(V, "fm_reads", "fm_roff")
    buf, offs = simdata.read_buffer(reads)
    tasks = np.zeros(len(reads), dtype=SEED_DT)
    for i, r in enumerate(reads):
        tasks[i] = (offs[i], len(r), int(V["fm_par"][i][0]))
    opt = default_opt()
    port.set_opt(opt)
    port.set_reads(buf)
    iv, off = port.seed(opt, tasks)
    co = V["fm_coff"] // 4
    assert (np.asarray(off) == co).all()
    assert (iv.reshape(-1) == V["fm_collect"]).all()
    assert int(co[-1]) > 1000


def test_nt4_table_vs_reference(V):
    L = B.lib()
    L.bsx_hook_nt4_table.restype = C.POINTER(C.c_uint8)
    t = np.ctypeslib.as_array(L.bsx_hook_nt4_table(), shape=(256,))
    assert (t == V["nt4_table"]).all()   # nst_nt4_table, bntseq.c:49-66, recorded as data


def test_fastq_grammar_vs_reference_kseq(V, tmp_path):
    """csrc/host/fastq.c against records the reference's own kseq_read (utils.c:53 instantiation of kseq.h:182-222) produced
    for the same bytes: name / comment / sequence / quality, including where reading stops."""
    import test_fastq_reader as T
    nt4 = V["nt4_table"]
    cases = [bytes(x).decode() for x in _ragged(V, "fq_case", "fq_case_off")]
    wants = [bytes(x).decode() for x in _ragged(V, "fq_recs", "fq_recs_off")]
    assert len(cases) >= 8
    for i, (text, want) in enumerate(zip(cases, wants)):
        fn = str(tmp_path / ("c%d.fq" % i))
        open(fn, "w").write(text)
        got = [r for ch in T.read_all(fn, None, 1 << 30) for r in ch]
        exp = []
        for line in want.split("\n"):
            if not line or line.startswith("END"):
                break
            name, comment, seq, qual = line.split("\t")
            if len(name) > 2 and name[-2] == "/" and name[-1].isdigit():
                name = name[:-2]                       # trim_readno (bwa.c:58-63)
            codes = "".join("ACGTN"[min(int(nt4[ord(c)]), 4)] for c in seq)
            exp.append((name, "" if comment == "*" else comment, codes, None if qual == "*" else qual))
        assert got == exp, (i, got[:3], exp[:3])


def test_sam_header_vs_reference(V, tmp_path):
    """bsx_sam_header against text printed by the real bwa_print_sam_hdr (bwa.c:654-684): @SQ lines sorted by name unless the
    -H text brings its own, then the -H text, then @PG."""
    L = B.lib()
    L.bsx_sam_header.restype = C.c_void_p
    L.bsx_sam_header.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    ins = [bytes(x).decode() for x in _ragged(V, "hdr_in", "hdr_in_off")]
    outs = [bytes(x).decode() for x in _ragged(V, "hdr_out", "hdr_out_off")]
    assert len(ins) >= 5
    for k, (spec, want) in enumerate(zip(ins, outs)):
        names, lens, hl, pg = spec.split("\x1e")
        names, lens = names.split("\x1f"), [int(x) for x in lens.split("\x1f")]
        fa = str(tmp_path / ("h%d.fa" % k))
        with open(fa, "w") as f:
            for n, l in zip(names, lens):
                f.write(">%s\n%s\n" % (n, "ACGT" * (l // 4) + "ACGT"[:l % 4]))
        idx = Index.from_fasta(fa)
        p = L.bsx_sam_header(idx.h, hl.encode() if hl else None, pg.encode() if pg else None)
        got = C.string_at(p).decode()
        from biscuit_amd.api import _libc_free
        _libc_free(p)
        idx.close()
        assert got == want, (k, got[:200], want[:200])


def test_read_clipping_vs_reference(V):
    """D1: read_clipping (adaptor search, fixed clips, quality clipping; bwamem.c:258-303) -- 400 reads recorded from the reference's own
    (static) functions, compiled through oracle/ref_statics.c; and the read-name rule of check_paired_read_names (bwamem.c:210-216)."""
    L = B.lib()
    L.bsx_hook_clip_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
    ins = _ragged(V, "clip_in", "clip_in_off")
    n_adapt = n_qual = 0
    for rec, want in zip(ins, V["clip_out"]):
        l, la, c5, c3, mbq, hasq = [int(x) for x in rec[:6]]
        seq = np.ascontiguousarray(rec[6:6 + l].astype(np.uint8))
        ad = np.ascontiguousarray(rec[6 + l:6 + l + la].astype(np.uint8))
        qual = bytes(int(x) for x in rec[6 + l + la:6 + l + la + l]) if hasq else None
        opt = default_opt()
        opt.clip5, opt.clip3, opt.min_base_qual = c5, c3, mbq
        out = (C.c_int * 5)()
        L.bsx_hook_clip_read(C.byref(opt), l, seq.ctypes.data_as(C.c_void_p), qual, ad.ctypes.data_as(C.c_void_p) if la else None, la, out)
        got, ref = list(out), [int(x) for x in want]
        if ref[3] == 0:
            # nothing of the read is left: the reference's second quality loop then looks at qual[-1] (bwamem.c:280-281, one byte before the
            # string) and its clip3 depends on what lies there; the read is unmapped either way and clip3 is never printed.  Not compared.
            got[2] = ref[2] = -1
        assert got == ref, (l, la, c5, c3, mbq, list(out), list(want))
        n_adapt += want[0] > 0
        n_qual += want[1] > c5 or want[2] > c3 + want[0]
    assert n_adapt > 100 and n_qual > 100
    L.bsx_hook_pair_names_ok.argtypes = [C.c_char_p, C.c_char_p]
    for pair in bytes(V["names_ok"]).decode().split("\x1e"):
        n1, n2 = pair.split("\x1f")
        assert L.bsx_hook_pair_names_ok(n1.encode(), n2.encode()) == 1
    for n1, n2 in (("r1", "r3"), ("a/1", "b/2"), ("x2", "x1")):
        assert L.bsx_hook_pair_names_ok(n1.encode(), n2.encode()) == 0


def test_header_inline_functions_of_the_pairing_code(V):
    """mem_infer_isize / mem_alnreg_isize / is_proper_pair / get_pri_idx / region_depos (mem_alnreg.h:75-144), get_rlen (bwamem.h:200) and
    bns_depos (bntseq.h:92): the reference's own inline functions (compiled through oracle/ref_shim.c, recorded in ref_vectors.npz)
    against the product's restatements in region.c / sam.c and against oracle/backhalf.py, which the end-to-end oracle uses."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import backhalf
    import e2e
    L = B.lib()
    i64p = C.POINTER(C.c_int64)
    L.bsx_hook_infer_isize.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, i64p]
    L.bsx_hook_reg_isize.argtypes = [C.c_int64, i64p, i64p, i64p]
    L.bsx_hook_is_proper_pair.argtypes = [C.c_int64, i64p, i64p, C.c_int, C.c_int]
    L.bsx_hook_get_pri_idx.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.bsx_hook_region_depos.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    L.bsx_hook_depos.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int)]
    L.bsx_hook_depos.restype = C.c_int64
    L.bsx_hook_get_rlen.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    l_pac = 1000000
    for r in V["isize_infer"]:
        p1, p2, r1, r2, l1, l2, ok, isz = [int(x) for x in r]
        iz = C.c_int64(0)
        assert L.bsx_hook_infer_isize(p1, p2, r1, r2, l1, l2, C.byref(iz)) == ok and (not ok or iz.value == isz)
        got = backhalf.infer_isize(p1, p2, r1, r2, l1, l2)
        assert (got is not None) == bool(ok) and (not ok or got == isz)
    n_ok = n_pp = 0
    for r in V["isize_pair"]:
        a, b = np.array(r[0:5], np.int64), np.array(r[5:10], np.int64)
        low, high, ok, isz, pp = [int(x) for x in r[10:15]]
        iz = C.c_int64(0)
        assert L.bsx_hook_reg_isize(l_pac, a.ctypes.data_as(i64p), b.ctypes.data_as(i64p), C.byref(iz)) == ok and (not ok or iz.value == isz), r
        assert L.bsx_hook_is_proper_pair(l_pac, a.ctypes.data_as(i64p), b.ctypes.data_as(i64p), low, high) == pp, r
        d1 = dict(rid=int(a[0]), rb=int(a[1]), re=int(a[2]), qb=int(a[3]), qe=int(a[4]))
        d2 = dict(rid=int(b[0]), rb=int(b[1]), re=int(b[2]), qb=int(b[3]), qe=int(b[4]))
        got = backhalf.alnreg_isize(l_pac, d1, d2)
        assert (got is not None) == bool(ok) and (not ok or got == isz), r
        assert int(got is not None and low <= got <= high) == pp
        n_ok += ok; n_pp += pp
    assert n_ok > 100 and n_pp > 50
    for r in V["pri_idx"]:
        ratio, i = float(r[0]), int(r[1])
        sc = np.array(r[2:10], np.int32); sa = np.array(r[10:18], np.int32)
        want = int(r[18])
        assert L.bsx_hook_get_pri_idx(ratio, 8, sc.ctypes.data_as(C.POINTER(C.c_int)), sa.ctypes.data_as(C.POINTER(C.c_int)), i) == want
        regs = [dict(score=int(sc[k]), secondary_all=int(sa[k])) for k in range(8)]
        assert backhalf._pri_idx(dict(XA_drop_ratio=ratio), regs, i) == want and e2e._pri_idx(dict(XA_drop_ratio=ratio), regs, i) == want
    for r in V["region_depos"]:
        lp, off, rb, re, pos, isr, dp, isr2 = [int(x) for x in r]
        assert L.bsx_hook_region_depos(lp, off, rb, re) == pos
        assert backhalf.region_depos(lp, [off], dict(rid=0, rb=rb, re=re)) == pos
        fl = C.c_int(-1)
        assert L.bsx_hook_depos(lp, rb, C.byref(fl)) == dp and fl.value == isr2
    cg, co = V["rlen_cigar"], V["rlen_cigar_off"]
    for i, want in enumerate(V["rlen_out"]):
        c = np.ascontiguousarray(cg[co[i]:co[i + 1]], np.uint32)
        assert L.bsx_hook_get_rlen(len(c), c.ctypes.data_as(C.POINTER(C.c_uint32))) == int(want)
        assert backhalf.get_rlen([int(x) for x in c]) == int(want) and e2e.get_rlen([int(x) for x in c]) == int(want)
