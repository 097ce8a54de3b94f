"""CPU: the product's host implementations of mem_mark_primary_se, mem_pestat and mem_pair (csrc/host/region.c, reached through
csrc/host/hooks.c) against oracle/backhalf.py -- an independent restatement of the same reference functions written from
lib/aln/mem_alnreg.c:252-380 and lib/aln/mem_pair.c:41-270.  The reference files cannot be compiled here; this
makes these rows two implementations that must agree bit for bit (doubles included) instead of one compared with itself."""
import ctypes as C
import os
import sys
import numpy as np
import pytest
import simdata
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, default_opt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import backhalf  # noqa: E402


class HookReg(C.Structure):
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64)] + [(k, C.c_int32) for k in
                ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt")] + \
               [("hash", C.c_uint64), ("flag", C.c_int32), ("mapq", C.c_int32), ("frac_rep", C.c_float), ("bss", C.c_uint8), ("parent", C.c_uint8), ("pad", C.c_uint8 * 2),
                ("pos", C.c_int32), ("n_cigar", C.c_int32), ("NM", C.c_int32), ("bss_u", C.c_int32), ("is_rev", C.c_uint32), ("ZC", C.c_uint32), ("ZR", C.c_uint32),
                ("pad2", C.c_uint32), ("cigar", C.c_void_p)]


KEYS = ("rb", "re", "qb", "qe", "rid", "score", "is_alt", "bss")


def to_c(regs):
    a = (HookReg * max(1, len(regs)))()
    for k, r in enumerate(regs):
        for f in KEYS:
            setattr(a[k], f, int(r[f]))
        for f in ("sub", "sub_n", "alt_sc", "secondary", "secondary_all", "csub", "seedcov", "flag", "mapq"):
            setattr(a[k], f, int(r.get(f, 0)))
        a[k].frac_rep = float(r.get("frac_rep", 0.0))
    return a


def opt_dict(o):
    return {k: getattr(o, k) for k in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "min_seed_len", "max_ins", "mask_level")}


@pytest.fixture(scope="module")
def small(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("bh"))
    contigs = simdata.make_genome(400000, seed=3, n_contigs=3)
    simdata.write_genome(d + "/g.fa", contigs)
    idx = Index.build(d + "/g.fa", d + "/g")
    L = B.lib()
    L.bsx_index_contig.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    offs, lens = [], []
    for i in range(3):
        nm, off, ln = C.c_char_p(), C.c_int64(), C.c_int64()
        B.check(L.bsx_index_contig(idx.h, i, C.byref(nm), C.byref(off), C.byref(ln)), "contig")
        offs.append(off.value)
        lens.append(ln.value)
    yield idx, offs, lens
    idx.close()


def rand_regs(rng, n, l_pac, alt_frac=0.15):
    out = []
    for _ in range(n):
        qb = int(rng.integers(0, 110))
        qe = qb + int(rng.integers(20, 151 - qb))
        rb = int(rng.integers(0, 2 * l_pac - 400))
        out.append({"rb": rb, "re": rb + (qe - qb) + int(rng.integers(-3, 4)), "qb": qb, "qe": qe, "rid": 0,
                    "score": int(rng.choice([30, 45, 45, 60, 80, 100, 100, 120, 149])) if rng.random() < 0.5 else int(rng.integers(20, 150)),
                    "is_alt": int(rng.random() < alt_frac), "bss": int(rng.integers(0, 2)), "sub_n": 0})   # sub_n: zero as mem_alnreg_t comes from calloc
    return out


def test_mark_primary(small):
    idx, _, _ = small
    L = B.lib()
    opt = default_opt()
    od = opt_dict(opt)
    L.bsx_hook_mark_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
    rng = np.random.default_rng(5)
    seen_second_round = seen_sub_n = 0
    for trial in range(4000):
        n = int(rng.integers(1, 11))
        regs = rand_regs(rng, n, idx.l_pac, alt_frac=0.0 if trial % 3 == 0 else 0.3)
        rid = int(rng.integers(0, 1 << 40))
        a = to_c(regs)
        n_pri = L.bsx_hook_mark_primary(C.byref(opt), a, n, rid)
        want_pri = backhalf.mark_primary_se(od, regs, rid)
        assert n_pri == want_pri, trial
        for k, r in enumerate(regs):
            got = tuple(getattr(a[k], f) for f in ("score", "qb", "qe", "is_alt", "sub", "sub_n", "alt_sc", "secondary", "secondary_all", "hash"))
            want = tuple(r[f] for f in ("score", "qb", "qe", "is_alt", "sub", "sub_n", "alt_sc", "secondary", "secondary_all", "hash"))
            assert got == want, (trial, k, got, want)
        seen_second_round += 0 < want_pri < n
        seen_sub_n += any(r["sub_n"] > 0 for r in regs)
    assert seen_second_round > 500 and seen_sub_n > 500


def _pairs(rng, n_pairs, l_pac, offs, lens):
    """reads[2i], reads[2i+1]: mostly a proper pair ~N(300, 40) apart on opposite strands, plus decoys, other contigs and strands"""
    reads = []
    for _ in range(n_pairs):
        rid = int(rng.integers(0, 3))
        ins = int(rng.normal(300, 40))
        f = offs[rid] + int(rng.integers(0, lens[rid] - 800))
        flip = rng.random() < 0.5
        r1 = {"rb": f, "re": f + 150, "qb": 0, "qe": 150, "rid": rid, "score": int(rng.integers(100, 151)), "is_alt": 0, "bss": 0}
        e = f + max(160, ins)                      # forward end of the mate, which lies on the reverse strand
        r2 = {"rb": 2 * l_pac - e, "re": 2 * l_pac - e + 150, "qb": 0, "qe": 150, "rid": rid, "score": int(rng.integers(100, 151)), "is_alt": 0, "bss": 0}
        if flip:
            r1, r2 = r2, r1
        u = rng.random()
        if u < 0.05:
            r2 = dict(r2, bss=1)
        elif u < 0.10:
            r2 = dict(r2, rid=(rid + 1) % 3)
        elif u < 0.15:
            r2 = dict(r2, rb=r1["rb"] + 40, re=r1["rb"] + 190)   # same strand
        elif u < 0.18:
            r2 = None
        lists = []
        for r in (r1, r2):
            if r is None:
                lists.append([])
                continue
            lst = [r]
            for _ in range(int(rng.integers(0, 3))):           # lower hits elsewhere, some overlapping the best on the read
                qb = int(rng.integers(0, 100))
                d = dict(r, rb=int(rng.integers(0, 2 * l_pac - 400)), qb=qb, qe=qb + int(rng.integers(30, 151 - qb)), score=int(rng.integers(30, r["score"] + 1)))
                d["re"] = d["rb"] + d["qe"] - d["qb"]
                lst.append(d)
            lists.append(lst)
        reads += lists
    return reads


def test_pestat_and_pair(small):
    idx, offs, lens = small
    L = B.lib()
    opt = default_opt()
    od = opt_dict(opt)
    l_pac = idx.l_pac
    L.bsx_hook_pestat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bsx_hook_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(11)
    for n_pairs in (4, 40, 3000):          # too few pairs (failed), a small and a large sample
        reads = _pairs(rng, n_pairs, l_pac, offs, lens)
        flat = [r for lst in reads for r in lst]
        off = np.zeros(len(reads) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(lst) for lst in reads])
        a = to_c(flat)
        pes = B.PeStat()
        L.bsx_hook_pestat(C.byref(opt), idx.h, len(reads), a, off.ctypes.data_as(C.c_void_p), C.byref(pes))
        want = backhalf.pestat(od, l_pac, reads)
        assert pes.failed == want["failed"], n_pairs
        if want["failed"]:
            continue
        assert (pes.low, pes.high, pes.avg, pes.std) == (want["low"], want["high"], want["avg"], want["std"]), (n_pairs, pes.avg, want["avg"], pes.std, want["std"])
        assert 400 < pes.avg < 500 and 20 < pes.std < 60   # the simulated insert + one read length (isize counts the mate, mem_alnreg.h:75-83)
        # mem_pair of every pair with those statistics; regions ordered and counted as mem_mark_primary_se leaves them
        n_proper = n_sub_seen = 0
        for i in range(len(reads) >> 1):
            pr = [sorted(reads[2 * i], key=lambda r: -r["score"]), sorted(reads[2 * i + 1], key=lambda r: -r["score"])]
            if rng.random() < 0.3 and pr[0] and pr[1]:   # a second candidate near the mate: competing pairings
                d = dict(pr[1][0], rb=pr[1][0]["rb"] + 7, re=pr[1][0]["re"] + 7, score=max(20, pr[1][0]["score"] - int(rng.integers(0, 12))))
                pr[1].insert(1, d)
            npri = [len(pr[0]), len(pr[1])]
            out = (C.c_int * 5)()
            a0, a1 = to_c(pr[0]), to_c(pr[1])
            L.bsx_hook_pair(C.byref(opt), idx.h, C.byref(pes), a0, len(pr[0]), npri[0], a1, len(pr[1]), npri[1], i, out)
            w = backhalf.pair(od, l_pac, offs, want, pr, npri, i)
            assert tuple(out) == w, (n_pairs, i, tuple(out), w)
            n_proper += w[3] >= 0
            n_sub_seen += w[2] > 0
        if n_pairs >= 3000:
            assert n_proper > 0.6 * n_pairs and n_sub_seen > 100


def test_sort_dedup(small):
    """mem_sort_deduplicate: the host function (region.c; the device's k_dedup is compared with it in tests/test_gpu_align.py) against
    the restatement, whose two sorts are the real klib introsort (oracle/_ref) -- the keys are not unique and the order klib leaves
    equal keys in decides which of two identical hits survives."""
    from oracle_lib import ref_lib
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref is not built")
    idx, _, _ = small
    L = B.lib()
    opt = default_opt()
    od = dict(opt_dict(opt), mask_level_redun=opt.mask_level_redun, max_chain_gap=opt.max_chain_gap, w=opt.w)
    l_pac = idx.l_pac
    R.ref_introsort_kv.argtypes = [C.c_int64, C.c_void_p]

    def klib_order(keys):
        kv = np.zeros((len(keys), 2), dtype=np.int64)
        kv[:, 0] = keys
        kv[:, 1] = np.arange(len(keys))
        if len(keys):
            R.ref_introsort_kv(len(keys), kv.ctypes.data_as(C.c_void_p))
        return [int(x) for x in kv[:, 1]]
    L.bsx_hook_regs_sort_dedup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.bsx_hook_regs_sort_dedup.restype = C.c_int
    dt = np.dtype(B.Region)
    rng = np.random.default_rng(23)
    n_cmp = n_drop = n_tie = n_concat = 0
    for trial in range(6000):
        n = int(rng.integers(1, 25))
        base = int(rng.integers(1000, l_pac - 5000)) + (l_pac if rng.random() < 0.5 else 0)
        regs = []
        for _ in range(n):
            if regs and rng.random() < 0.35:       # a near copy of an earlier one: same end, same start, or identical
                r = dict(regs[int(rng.integers(0, len(regs)))])
                u = rng.random()
                if u < 0.4:
                    r["rb"] -= int(rng.integers(0, 4)); r["qb"] = max(0, r["qb"] - int(rng.integers(0, 4)))
                elif u < 0.7:
                    r["score"] = max(20, r["score"] - int(rng.integers(0, 3)))
            else:
                qb = int(rng.integers(0, 100))
                qe = qb + int(rng.integers(25, 151 - qb))
                rb = base + int(rng.integers(-300, 1500)) if rng.random() < 0.8 else int(rng.integers(0, 2 * l_pac - 400))
                r = {"rb": rb, "re": rb + (qe - qb) + int(rng.integers(-2, 3)), "qb": qb, "qe": qe, "rid": 0 if rng.random() < 0.9 else 1,
                     "score": int(rng.integers(25, 150))}
            regs.append(r)
        want = backhalf.sort_dedup(od, l_pac, [dict(r) for r in regs], klib_order)
        arr = np.zeros(n, dtype=dt)
        for k, r in enumerate(regs):
            for f in ("rb", "re", "qb", "qe", "rid", "score"):
                arr[k][f] = r[f]
        keep = np.zeros(n, dtype=np.int32)
        m = L.bsx_hook_regs_sort_dedup(C.byref(opt), idx.h, arr.ctypes.data_as(C.c_void_p), n, keep.ctypes.data_as(C.c_void_p))
        if want is None:
            assert m == -1, trial
            n_concat += 1
            continue
        assert m == len(want) and list(keep[:m]) == want, (trial, list(keep[:max(m, 0)]), want)
        n_cmp += 1
        n_drop += m < n
        n_tie += len(set(r["re"] for r in regs)) < n
    assert n_cmp > 4000 and n_drop > 2000 and n_tie > 2000 and n_concat > 20, (n_cmp, n_drop, n_tie, n_concat)


def test_rescued_hits_added_one_by_one(small):
    """Mate rescue adds hit after hit to a list, each followed by mem_sort_deduplicate (mem_alnreg.c:478-488).  The product keeps the list's
    order by end between the hits and runs only the redundancy scan (region.c, bsx_regs_insert_dedup), falling back to the two sorts
    when one of them would have a tie to break.  Against the restatement's plain sequence with the real klib introsort, on lists full of
    near copies: shared ends, shared starts, equal scores, identical hits, hits that knock out several regions and then lose themselves."""
    from oracle_lib import ref_lib
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref is not built")
    idx, _, _ = small
    L = B.lib()
    opt = default_opt()
    od = dict(opt_dict(opt), mask_level_redun=opt.mask_level_redun, max_chain_gap=opt.max_chain_gap, w=opt.w)
    l_pac = idx.l_pac
    R.ref_introsort_kv.argtypes = [C.c_int64, C.c_void_p]

    def klib_order(keys):
        kv = np.zeros((len(keys), 2), dtype=np.int64)
        kv[:, 0] = keys
        kv[:, 1] = np.arange(len(keys))
        if len(keys):
            R.ref_introsort_kv(len(keys), kv.ctypes.data_as(C.c_void_p))
        return [int(x) for x in kv[:, 1]]
    L.bsx_hook_regs_insert_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.bsx_hook_regs_insert_seq.restype = C.c_int
    dt = np.dtype(B.Region)
    rng = np.random.default_rng(29)
    n_drop = n_tie = n_multi = 0
    for trial in range(1500):
        tie_rate = float(rng.choice([0.0, 0.0, 0.15, 0.4]))       # lists without ties take the incremental path all the way
        n0, nh = int(rng.integers(0, 40)), int(rng.integers(1, 30))
        base = int(rng.integers(1000, l_pac - 8000)) + (l_pac if rng.random() < 0.5 else 0)
        regs = []
        for k in range(n0 + nh):
            if regs and rng.random() < 0.45:
                r = dict(regs[int(rng.integers(0, len(regs)))])
                u = rng.random()
                if u < tie_rate:
                    pass                                            # identical
                elif u < 0.5:
                    d = int(rng.integers(1, 6)); r["rb"] -= d; r["qb"] = max(0, r["qb"] - d); r["score"] += int(rng.integers(-3, 6))
                    if rng.random() > tie_rate:
                        r["re"] += int(rng.integers(1, 4)) * (1 if rng.random() < 0.5 else -1)
                else:
                    d = int(rng.integers(1, 6)); r["re"] += d; r["qe"] += d; r["score"] += int(rng.integers(-3, 6))
            else:
                qb = int(rng.integers(0, 100))
                qe = qb + int(rng.integers(25, 151 - qb))
                rb = base + int(rng.integers(-300, 2500 if tie_rate else 6000)) if rng.random() < 0.85 else int(rng.integers(0, 2 * l_pac - 400))
                r = {"rb": rb, "re": rb + (qe - qb) + int(rng.integers(-2, 3)), "qb": qb, "qe": qe, "rid": 0 if rng.random() < 0.9 else 1,
                     "score": int(rng.integers(25, 150))}
                if tie_rate == 0:
                    while any(x["re"] == r["re"] for x in regs):
                        r["re"] += 1; r["rb"] += 1
            if tie_rate == 0 and any(x["re"] == r["re"] for x in regs):
                r["re"] += 7; r["qe"] += 7
                while any(x["re"] == r["re"] for x in regs):
                    r["re"] += 1
            regs.append(r)
        # the restatement: the starting list as it stands, then per hit: ahead of the first lower score, sort + de-duplicate
        cur = list(range(n0))
        for k in range(n0, n0 + nh):
            ins = next((i for i, c in enumerate(cur) if regs[c]["score"] < regs[k]["score"]), len(cur))
            arr = cur[:ins] + [k] + cur[ins:]
            kept = backhalf.sort_dedup(od, l_pac, [dict(regs[c]) for c in arr], klib_order, can_merge=False)
            n_drop += len(kept) < len(arr)
            n_multi += len(kept) < len(arr) - 1
            cur = [arr[i] for i in kept]
        n_tie += len(set(r["re"] for r in regs)) < len(regs)
        arr = np.zeros(n0 + nh, dtype=dt)
        for k, r in enumerate(regs):
            for f in ("rb", "re", "qb", "qe", "rid", "score"):
                arr[k][f] = r[f]
        keep = np.zeros(n0 + nh, dtype=np.int32)
        m = L.bsx_hook_regs_insert_seq(C.byref(opt), idx.h, arr.ctypes.data_as(C.c_void_p), n0, n0 + nh, keep.ctypes.data_as(C.c_void_p))
        assert list(keep[:m]) == cur, (trial, list(keep[:m]), cur)
    assert n_drop > 2000 and n_multi > 100 and 300 < n_tie < 1100, (n_drop, n_multi, n_tie)


def test_mate_rescue(small):
    """mem_alnreg_matesw: the pipeline's plan / K5 batch / replay form of it (pipeline.c, K5 by the CPU restatement of ksw_align2) against
    the restatement's plain loop, whose Smith-Waterman is the REAL ksw_align2 and whose sorts are the real klib introsort (oracle/_ref)."""
    from oracle_lib import ref_lib, Port
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref is not built")
    idx, offs, lens = small
    L = B.lib()
    opt = default_opt()
    l_pac = idx.l_pac
    pac = np.fromfile(idx.base + ".bis.pac", dtype=np.uint8)
    ii = np.arange(l_pac)
    g = ((pac[ii >> 2] >> ((~ii & 3) << 1)) & 3).astype(np.uint8)
    od = dict(opt_dict(opt), mask_level_redun=opt.mask_level_redun, max_chain_gap=opt.max_chain_gap, w=opt.w, pen_unpaired=opt.pen_unpaired,
              max_matesw=opt.max_matesw, ctmat=(C.c_int8 * 25)(*opt.ctmat), gamat=(C.c_int8 * 25)(*opt.gamat))
    R.ref_introsort_kv.argtypes = [C.c_int64, C.c_void_p]

    def klib_order(keys):
        kv = np.zeros((len(keys), 2), dtype=np.int64)
        kv[:, 0] = keys
        kv[:, 1] = np.arange(len(keys))
        if len(keys):
            R.ref_introsort_kv(len(keys), kv.ctypes.data_as(C.c_void_p))
        return [int(x) for x in kv[:, 1]]

    def ksw_align2(query, target, mat, xtra):
        q = np.array(query, dtype=np.uint8)
        t = np.array(target, dtype=np.uint8)
        out = (C.c_int * 7)()
        R.ref_ksw_align2(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra, out)
        return dict(zip(("score", "te", "qe", "score2", "te2", "tb", "qb"), out))
    rng = np.random.default_rng(41)
    pes_d = {"low": 120, "high": 560, "avg": 330.0, "std": 45.0, "failed": 0}
    pes = B.PeStat(low=120, high=560, set=1, failed=0, avg=330.0, std=45.0)
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    n_pairs = 300
    seqs, lists = [], []
    for _ in range(n_pairs):
        rid = int(rng.integers(0, 3))
        ins = int(rng.integers(200, 480))
        f = offs[rid] + int(rng.integers(0, lens[rid] - 600))
        frag = g[f:f + ins].copy()
        frag[(frag == 1) & (rng.random(ins) < 0.9)] = 3                  # top strand, C>T converted
        for k in rng.integers(0, ins, size=int(rng.integers(0, 4))):    # a few substitutions
            frag[k] = (frag[k] + 1) & 3
        r1 = frag[:100].copy()
        r2 = comp[frag[::-1][:100]]
        if rng.random() < 0.3:
            r1, r2 = r2, r1                                             # the pair the other way round: read 1 on the reverse strand
            reg1 = {"rb": 2 * l_pac - (f + ins), "re": 2 * l_pac - (f + ins) + 100, "bss": 0}
            reg2_true = {"rb": f, "re": f + 100, "bss": 0}
        else:
            reg1 = {"rb": f, "re": f + 100, "bss": 0}
            reg2_true = {"rb": 2 * l_pac - (f + ins), "re": 2 * l_pac - (f + ins) + 100, "bss": 0}
        base = {"qb": 0, "qe": 100, "rid": rid, "score": int(rng.integers(80, 101)), "is_alt": 0}
        l1 = [dict(base, **reg1)]
        u = rng.random()
        if u < 0.5:
            l2 = []                                                     # the mate was not found: to be rescued
        elif u < 0.7:
            far = int(rng.integers(0, 2 * l_pac - 300))
            l2 = [dict(base, rb=far, re=far + 100, bss=0, score=int(rng.integers(40, 90)))]     # found elsewhere: rescue adds the proper one
        elif u < 0.85:
            l2 = [dict(base, **reg2_true)]                              # already properly paired: nothing to do
        else:
            l2 = [dict(base, **reg2_true, score=60), dict(base, rb=reg2_true["rb"] + 1, re=reg2_true["re"] + 1, bss=0, score=55)]
        if rng.random() < 0.3:
            l1.append(dict(base, rb=int(rng.integers(0, 2 * l_pac - 300)), bss=0, score=l1[0]["score"] - int(rng.integers(0, 25))))
            l1[-1]["re"] = l1[-1]["rb"] + 100
        seqs += [r1, r2]
        lists += [l1, l2]
    n = 2 * n_pairs
    # the product: hook over the pipeline's own mate_rescue, K5 by the CPU restatement
    port = Port(idx, 1)
    be = port.backend()
    reads = (B.Read * n)()
    keep_alive = []
    for i, s in enumerate(seqs):
        buf = (C.c_uint8 * len(s))(*[int(x) for x in s])
        keep_alive.append(buf)
        reads[i].l_seq = len(s)
        reads[i].seq = C.cast(buf, C.POINTER(C.c_uint8))
    flat = [dict(r, parent=0) for lst in lists for r in lst]
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(lst) for lst in lists])
    a = to_c(flat)
    cap = len(flat) + 4 * n
    out = (HookReg * cap)()
    out_off = np.zeros(n + 1, dtype=np.int64)
    L.bsx_hook_mate_rescue.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    B.check(L.bsx_hook_mate_rescue(C.byref(be), C.byref(opt), idx.h, C.byref(pes), n, reads, a, off.ctypes.data_as(C.c_void_p), out,
                                   out_off.ctypes.data_as(C.c_void_p), cap), "bsx_hook_mate_rescue")
    anns = list(zip(offs, lens))
    rescued = unchanged = 0
    for p in range(n_pairs):
        pr = [[dict(r) for r in lists[2 * p]], [dict(r) for r in lists[2 * p + 1]]]
        before = [len(pr[0]), len(pr[1])]
        backhalf.matesw(od, l_pac, anns, lambda k: int(g[k]), pes_d, [[int(x) for x in seqs[2 * p]], [int(x) for x in seqs[2 * p + 1]]], pr, ksw_align2, klib_order)
        for w in range(2):
            i = 2 * p + w
            got = [tuple(getattr(out[k], f) for f in ("rb", "re", "qb", "qe", "rid", "score", "csub", "is_alt", "bss", "parent", "seedcov", "secondary"))
                   for k in range(out_off[i], out_off[i + 1])]
            want = [tuple(r.get(f, 0) for f in ("rb", "re", "qb", "qe", "rid", "score", "csub", "is_alt", "bss", "parent", "seedcov", "secondary")) for r in pr[w]]
            assert got == want, (p, w, got, want)
        rescued += len(pr[0]) + len(pr[1]) > sum(before)
        unchanged += len(pr[0]) + len(pr[1]) == sum(before)
    assert rescued > 0.3 * n_pairs and unchanged > 0.1 * n_pairs, (rescued, unchanged)


def test_reg2sam_pe_decisions(small):
    """mem_reg2sam_pe / _nopairing / mem_alnreg_select_format up to the text: which records a pair gets, each with its flag, MAPQ and
    mate, and the state the regions are left in.  Product: the planning pass of sam.c with its trace sink; restatement: oracle/backhalf.py
    with the REAL mem_approx_mapq_se (oracle/_ref)."""
    from oracle_lib import ref_lib
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref is not built")
    idx, offs, lens = small
    L = B.lib()
    l_pac = idx.l_pac
    R.ref_approx_mapq_se.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int] + [C.c_int] * 6 + [C.c_int64, C.c_int64, C.c_int, C.c_float]
    L.bsx_hook_reg2sam_pe_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(77)
    n_paired = n_nopair = n_alt = n_switch = 0
    for variant, extra_flag in enumerate((0, 0x10, 0x8 | 0x10, 0x4)):      # defaults, NO_MULTI (the CLI's), ALL, NOPAIRING
        opt = default_opt()
        opt.flag |= 0x2 | extra_flag
        od = dict(opt_dict(opt), T=opt.T, flag=opt.flag, drop_ratio=opt.drop_ratio, pen_unpaired=opt.pen_unpaired)

        def mapq_se(p):
            return R.ref_approx_mapq_se(opt.a, opt.b, opt.min_seed_len, opt.mapQ_coef_len, opt.mapQ_coef_fac, p["score"], p["sub"], p["csub"], p["sub_n"],
                                        p["qb"], p["qe"], p["rb"], p["re"], p["seedcov"], p["frac_rep"])
        reads = _pairs(rng, 1200, l_pac, offs, lens)
        pes_d = backhalf.pestat(od, l_pac, reads)
        pes = B.PeStat(low=pes_d["low"], high=pes_d["high"], set=1, failed=0, avg=pes_d["avg"], std=pes_d["std"])
        for i in range(len(reads) >> 1):
            pr = []
            for w in range(2):
                lst = [dict(r, sub_n=0, csub=int(rng.integers(0, 60)) if rng.random() < 0.3 else 0, seedcov=int(rng.integers(20, 120)),
                            frac_rep=backhalf.f32(float(rng.random()) * 0.5) if rng.random() < 0.3 else 0.0, flag=0, mapq=0) for r in reads[2 * i + w]]
                if lst and rng.random() < 0.25:       # a competing hit close to the best one
                    lst.append(dict(lst[0], rb=lst[0]["rb"] + 5, re=lst[0]["re"] + 5, score=max(20, lst[0]["score"] - int(rng.integers(0, 15)))))
                if lst and rng.random() < 0.15:       # a hit on an ALT contig
                    lst.append(dict(lst[0], rb=int(rng.integers(0, l_pac - 400)), is_alt=1, score=int(rng.integers(25, 151))))
                    lst[-1]["re"] = lst[-1]["rb"] + 150
                pr.append(lst)
            npri = [backhalf.mark_primary_se(od, pr[w], 2 * i + w) for w in range(2)]
            a0, a1 = to_c(pr[0]), to_c(pr[1])
            trace = ((C.c_int * 6) * 64)()
            lens2 = (C.c_int * 2)(150, 150)
            nt = L.bsx_hook_reg2sam_pe_plan(C.byref(opt), idx.h, C.byref(pes), i, lens2, a0, len(pr[0]), npri[0], a1, len(pr[1]), npri[1], trace, 64)
            want = backhalf.reg2sam_pe(od, l_pac, offs, pes_d, i, pr, npri, mapq_se)
            got = [tuple(trace[k]) for k in range(nt)]
            assert got == want, (variant, i, got, want)
            for w, a in enumerate((a0, a1)):
                for k, r in enumerate(pr[w]):
                    g2 = tuple(getattr(a[k], f) for f in ("flag", "mapq", "sub", "secondary", "secondary_all"))
                    w2 = tuple(r[f] for f in ("flag", "mapq", "sub", "secondary", "secondary_all"))
                    assert g2 == w2, (variant, i, w, k, g2, w2)
            paired = len(want) >= 2 and want[0][2] >= 0 and want[0][5] == 1 and all(t[1] >= 0 for t in want[:2]) and len([t for t in want if t[0] == 0]) <= 2
            n_paired += paired
            n_nopair += not paired
            n_alt += any(t[2] == -2 for t in want)
            n_switch += any(r["secondary"] == -2 for lst in pr for r in lst)
    assert n_paired > 1500 and n_nopair > 200 and n_alt > 20 and n_switch > 5, (n_paired, n_nopair, n_alt, n_switch)


def test_format_sam(small):
    """mem_alnreg_formatSAM with its SA / XA / XB tags: one SAM line of sam.c against the restatement's, field by field of a record
    built at random (clips, indels, secondary / supplementary / unmapped records, mapped / unmapped / absent mates, ALT hits, read
    comments, barcodes, hard and soft clipping, -M)."""
    idx, offs, lens = small
    L = B.lib()
    l_pac = idx.l_pac
    L.bsx_index_contig.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    names = []
    for i in range(3):
        nm, off, ln = C.c_char_p(), C.c_int64(), C.c_int64()
        L.bsx_index_contig(idx.h, i, C.byref(nm), C.byref(off), C.byref(ln))
        names.append(nm.value.decode())
    L.bsx_hook_format_sam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(99)
    keep = []

    def rand_cigar(qlen):
        ops = []
        c5, c3 = (int(rng.integers(1, 20)) if rng.random() < 0.3 else 0), (int(rng.integers(1, 20)) if rng.random() < 0.3 else 0)
        body = qlen - c5 - c3
        if c5:
            ops.append(c5 << 4 | 3)
        if rng.random() < 0.3 and body > 40:
            a = int(rng.integers(10, body - 20))
            ops += [a << 4, int(rng.integers(1, 5)) << 4 | int(rng.integers(1, 3)), (body - a) << 4]
        else:
            ops.append(body << 4)
        if c3:
            ops.append(c3 << 4 | 3)
        return ops

    def rand_reg(mapped=True):
        rb = int(rng.integers(0, 2 * l_pac - 400))
        r = {"rb": rb, "re": rb + 150, "qb": 0, "qe": 150, "rid": int(rng.integers(0, 3)) if mapped else -1, "score": int(rng.integers(30, 151)),
             "sub": int(rng.integers(0, 100)) if rng.random() < 0.7 else 0, "csub": int(rng.integers(0, 80)) if rng.random() < 0.3 else 0,
             "alt_sc": int(rng.integers(20, 150)) if rng.random() < 0.2 else 0, "is_alt": int(rng.random() < 0.15), "bss": int(rng.integers(0, 2)),
             "secondary": -1, "secondary_all": -1, "flag": 0x41 if rng.random() < 0.5 else 0x81, "mapq": int(rng.integers(0, 61)),
             "pos": int(rng.integers(0, 100000)), "is_rev": int(rng.integers(0, 2)), "NM": int(rng.integers(0, 6)), "ZC": int(rng.integers(0, 40)),
             "ZR": int(rng.integers(0, 10)), "bss_u": int(rng.random() < 0.2)}
        r["cigar"] = rand_cigar(150) if mapped and rng.random() < 0.9 else []
        r["md"] = "%dA%d" % (int(rng.integers(0, 70)), int(rng.integers(0, 70))) if r["cigar"] else ""
        if not mapped:
            r.update(flag=0x40 | 0x1 | 0x4, score=0, sub=0, mapq=0, pos=0, is_rev=0)
        return r

    def c_reg(r):
        h = HookReg()
        for f in ("rb", "re", "qb", "qe", "rid", "score", "sub", "csub", "alt_sc", "is_alt", "bss", "secondary", "secondary_all", "flag", "mapq", "pos", "is_rev",
                  "NM", "ZC", "ZR", "bss_u"):
            setattr(h, f, int(r[f]))
        h.n_cigar = len(r["cigar"])
        if r["cigar"]:
            md = r["md"].encode() + b"\0"
            buf = (C.c_uint8 * (4 * len(r["cigar"]) + len(md) + 8))()
            C.memmove(buf, np.array(r["cigar"], dtype=np.uint32).tobytes(), 4 * len(r["cigar"]))
            C.memmove(C.addressof(buf) + 4 * len(r["cigar"]), md, len(md))
            keep.append(buf)
            h.cigar = C.addressof(buf)
        return h
    n_lines = n_xa = n_sa = 0
    for trial in range(3000):
        opt = default_opt()
        opt.flag |= 0x2 | (0x200 if trial % 5 == 0 else 0) | (0x8 if trial % 7 == 0 else 0) | (0x10 if trial % 2 else 0)
        opt.max_XA_hits = 2 if trial % 11 == 0 else opt.max_XA_hits
        od = {"flag": opt.flag, "XA_drop_ratio": opt.XA_drop_ratio, "max_XA_hits": opt.max_XA_hits, "max_XA_hits_alt": opt.max_XA_hits_alt}
        seq0 = rng.integers(0, 5, size=150).astype(np.uint8)
        qual = "".join(chr(int(x)) for x in rng.integers(35, 74, size=150)) if rng.random() < 0.8 else None
        s = {"name": "read%d" % trial, "comment": "c1" if rng.random() < 0.2 else None, "seq0": [int(x) for x in seq0], "qual": qual, "l_seq": int(rng.integers(120, 151)),
             "barcode": "ACGT" if rng.random() < 0.1 else None, "umi": "TTAA" if rng.random() < 0.1 else None}
        rd = B.Read()
        sbuf = (C.c_uint8 * 150)(*s["seq0"])
        keep.append(sbuf)
        rd.l_seq, rd.l_seq0, rd.name = s["l_seq"], 150, s["name"].encode()
        rd.comment = s["comment"].encode() if s["comment"] else None
        rd.qual = qual.encode() if qual else None
        rd.barcode = s["barcode"].encode() if s["barcode"] else None
        rd.umi = s["umi"].encode() if s["umi"] else None
        rd.seq0 = C.cast(sbuf, C.POINTER(C.c_uint8))
        rd.seq = rd.seq0
        regs = [rand_reg() for _ in range(int(rng.integers(1, 5)))]
        for r in regs:     # every region of the list has its CIGAR: the tags of a final pass only read them (the planning pass asked for them)
            if not r["cigar"]:
                r["cigar"], r["md"] = rand_cigar(150), "150"
        p_idx = int(rng.integers(0, len(regs)))
        for i, r in enumerate(regs):
            if i != p_idx and rng.random() < 0.6:       # a secondary of the record's region, near its score or not
                r["secondary"] = r["secondary_all"] = p_idx
                r["score"] = max(20, regs[p_idx]["score"] - int(rng.integers(0, 60)))
                if rng.random() < 0.5:
                    r["flag"] |= 0x100
            elif i != p_idx and rng.random() < 0.5:
                r["flag"] |= 0x800 if rng.random() < 0.5 else 0x10000
        use_list = rng.random() < 0.8
        u = rng.random()
        if u < 0.1 and not use_list:
            p0 = rand_reg(mapped=False)
        else:
            p0 = regs[p_idx]
            if rng.random() < 0.15:
                p0["flag"] |= 0x100
        m0 = None if rng.random() < 0.15 else rand_reg(mapped=rng.random() < 0.85)
        if m0 is not None and rng.random() < 0.5 and p0["rid"] >= 0 and m0["rid"] >= 0:   # a plausible mate: same contig, opposite strand, near
            m0.update(rid=p0["rid"], rb=(2 * l_pac - p0["rb"] - int(rng.integers(150, 600))) % (2 * l_pac - 200), is_rev=1 - p0["is_rev"], pos=p0["pos"] + int(rng.integers(-400, 400)))
            m0["re"] = m0["rb"] + 150
        is_primary = int(rng.random() < 0.7)
        pes_d = {"low": 100, "high": 700}
        pes = B.PeStat(low=100, high=700, set=1, failed=0, avg=350.0, std=60.0)
        rg = b"grp1" if rng.random() < 0.2 else None
        want = backhalf.format_sam(od, l_pac, names, ["", "", ""], s, p0, m0, regs if use_list and p0 is regs[p_idx] else None, p_idx, is_primary, pes_d, rg.decode() if rg else None)
        cregs = (HookReg * len(regs))(*[c_reg(r) for r in regs])
        cp = c_reg(p0)
        cm = c_reg(m0) if m0 is not None else None
        buf = C.create_string_buffer(8192)
        with_list = use_list and p0 is regs[p_idx]
        n = L.bsx_hook_format_sam(C.byref(opt), idx.h, C.byref(rd), C.byref(cp), C.byref(cm) if cm is not None else None, cregs if with_list else None,
                                  len(regs), p_idx if with_list else -1, is_primary, C.byref(pes), rg, buf, 8192)
        assert n > 0
        got = buf.raw[:n].decode()
        assert got == want, (trial, got, want)
        n_lines += 1
        n_xa += "XA:Z:" in want
        n_sa += "SA:Z:" in want
        del keep[:]
    assert n_lines == 3000 and n_xa > 200 and n_sa > 200, (n_xa, n_sa)


def test_strand_search_order():
    """D1: which converted index each read is searched against and in which order (bis_worker1, bwamem.c:311-376), for every value -b can
    give opt->parent, single-end and both reads of a pair"""
    L = B.lib()
    L.bsx_hook_strand_order.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for parent in (0, 1, 3):
        for is_pe in (0, 1):
            for second in (0, 1):
                opt = default_opt()
                opt.parent = parent
                order = (C.c_int * 2)()
                n = L.bsx_hook_strand_order(C.byref(opt), is_pe, second, order)
                assert list(order)[:n] == backhalf.strand_searches(parent, is_pe, second), (parent, is_pe, second)


def test_setsam_position_and_clipping(small):
    """mem_alnreg_setSAM after its alignment (mem_alnreg_format.c:79-120): position on the contig and strand, a leading or trailing deletion
    squeezed out of the CIGAR, the read's clips (adaptor / quality / fixed, and the unaligned ends) added -- sam.c against the restatement"""
    idx, offs, lens = small
    L = B.lib()
    l_pac = idx.l_pac
    L.bsx_hook_setsam_finish.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(123)
    opt = default_opt()
    n_rev = n_squeeze = n_clip = 0
    for trial in range(3000):
        rid = int(rng.integers(0, 3))
        l_seq = int(rng.integers(60, 151))
        qb = int(rng.integers(0, 15)) if rng.random() < 0.4 else 0
        qe = l_seq - (int(rng.integers(0, 15)) if rng.random() < 0.4 else 0)
        f = offs[rid] + int(rng.integers(0, lens[rid] - 400))
        tl = qe - qb + int(rng.integers(-3, 4))
        rev = rng.random() < 0.5
        rb = 2 * l_pac - (f + tl) if rev else f
        reg = {"rb": rb, "re": rb + tl, "qb": qb, "qe": qe, "rid": rid}
        ops = [(qe - qb) << 4]
        u = rng.random()
        if u < 0.15:
            ops = [int(rng.integers(1, 4)) << 4 | 2] + ops
        elif u < 0.3:
            ops = ops + [int(rng.integers(1, 4)) << 4 | 2]
        elif u < 0.45:
            a = int(rng.integers(5, qe - qb - 5))
            ops = [a << 4, 2 << 4 | 1, (qe - qb - a - 2) << 4]
        s = {"l_seq": l_seq, "clip5": int(rng.integers(0, 6)) if rng.random() < 0.3 else 0, "clip3": int(rng.integers(0, 20)) if rng.random() < 0.3 else 0}
        want_pos, want_rev, want_cig = backhalf.setsam_post(l_pac, offs, s, reg, ops)
        rd = B.Read()
        rd.l_seq, rd.clip5, rd.clip3 = s["l_seq"], s["clip5"], s["clip3"]
        h = HookReg()
        for k2, v in reg.items():
            setattr(h, k2, v)
        cg = np.array(ops, dtype=np.uint32)
        tag = B.GlbTag(NM=3, ZC=4, ZR=5, l_md=3, md_off=0, bss_u=0)
        out = (C.c_int * 3)()
        oc = np.zeros(16, dtype=np.uint32)
        omd = C.create_string_buffer(64)
        n = L.bsx_hook_setsam_finish(C.byref(opt), idx.h, C.byref(rd), C.byref(h), cg.ctypes.data_as(C.c_void_p), len(ops), C.byref(tag), b"7A9", out,
                                     oc.ctypes.data_as(C.c_void_p), 16, omd, 64)
        assert n == len(want_cig) and list(oc[:n]) == want_cig and (out[0], out[1], out[2]) == (want_pos, want_rev, 3) and omd.value == b"7A9", \
            (trial, n, list(oc[:max(n, 0)]), want_cig, list(out), want_pos, want_rev)
        n_rev += want_rev
        n_squeeze += len(want_cig) < len(ops) + (1 if want_cig and want_cig[0] & 0xf == 3 else 0) + (1 if want_cig and want_cig[-1] & 0xf == 3 else 0)
        n_clip += any(c & 0xf == 3 for c in want_cig)
    assert n_rev > 1000 and n_squeeze > 500 and n_clip > 1000


def test_flt_chained_seeds(small):
    """C3: mem_flt_chained_seeds / mem_seed_sw (memchain.c:501-568), the long-read seed filter: the pipeline's window + K5 batch + filter form
    (K5 by the CPU restatement of ksw_align2) against the restatement's loop over the REAL ksw_align2.  1 kb reads on both strands, true
    seeds, seeds at wrong loci, seeds too long to be tested, and a short read for which the filter is off."""
    from oracle_lib import ref_lib, Port
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref is not built")
    idx, offs, lens = small
    L = B.lib()
    opt = default_opt()
    l_pac = idx.l_pac
    pac = np.fromfile(idx.base + ".bis.pac", dtype=np.uint8)
    ii = np.arange(l_pac)
    g = ((pac[ii >> 2] >> ((~ii & 3) << 1)) & 3).astype(np.uint8)
    od = dict(opt_dict(opt), min_chain_weight=opt.min_chain_weight, ctmat=(C.c_int8 * 25)(*opt.ctmat), gamat=(C.c_int8 * 25)(*opt.gamat))

    def ksw_align2(query, target, mat, xtra):
        q = np.array(query, dtype=np.uint8)
        t = np.array(target, dtype=np.uint8)
        out = (C.c_int * 7)()
        R.ref_ksw_align2(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra, out)
        return dict(zip(("score", "te", "qe", "score2", "te2", "tb", "qb"), out))
    port = Port(idx, 1)
    be = port.backend()
    L.bsx_hook_flt_chained_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    anns = list(zip(offs, lens))
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    rng = np.random.default_rng(2024)
    n_dropped = n_kept = n_long = n_off = 0
    for trial in range(120):
        lq = 150 if trial % 10 == 0 else int(rng.integers(800, 1400))
        rid = int(rng.integers(0, 3))
        f = offs[rid] + int(rng.integers(0, lens[rid] - lq - 10))
        seg = g[f:f + lq].copy()
        rev = rng.random() < 0.4
        if rev:
            seg = comp[seg[::-1]]
        mut = rng.random(lq) < 0.04
        seg[mut] = (seg[mut] + 1 + rng.integers(0, 3, int(mut.sum()))) & 3
        parent = int(rng.integers(0, 2))
        if parent:
            seg[(seg == 1) & (rng.random(lq) < 0.8)] = 3
        else:
            seg[(seg == 2) & (rng.random(lq) < 0.8)] = 0
        r0 = 2 * l_pac - (f + lq) if rev else f          # where query position 0 lies in forward-reverse coordinates
        seeds, used = [], set()
        for _ in range(int(rng.integers(4, 12))):
            ln = int(rng.integers(19, 60)) if rng.random() < 0.9 else int(rng.integers(200, 260))
            qb = int(rng.integers(0, lq - ln)) if lq > ln else 0
            if lq <= ln:
                continue
            rb = r0 + qb if rng.random() < 0.7 else int(rng.integers(0, 2 * l_pac - 400))
            if (rb, qb, ln) not in used:
                used.add((rb, qb, ln))
                seeds.append((rb, qb, ln))
        sd = np.array(seeds, dtype=np.int64)
        keep = np.zeros(len(seeds), dtype=np.int32)
        score = np.zeros(len(seeds), dtype=np.int32)
        q = np.ascontiguousarray(seg)
        n = L.bsx_hook_flt_chained_seeds(C.byref(be), C.byref(opt), idx.h, lq, q.ctypes.data_as(C.c_void_p), parent, len(seeds), sd.ctypes.data_as(C.c_void_p),
                                         keep.ctypes.data_as(C.c_void_p), score.ctypes.data_as(C.c_void_p))
        want = backhalf.flt_chained_seeds(od, l_pac, anns, lambda k: int(g[k]), [int(x) for x in seg], seeds, parent, ksw_align2)
        assert n == len(want), (trial, n, want)
        for k, (wi, ws) in enumerate(want):
            assert keep[k] == wi and (ws is None or score[k] == ws), (trial, k, keep[:n], score[:n], want)
        if want and want[0][1] is None:
            n_off += 1
        else:
            n_dropped += len(seeds) - len(want)
            n_kept += len(want)
            n_long += sum(1 for s2 in seeds if s2[2] >= 200)
    assert n_off >= 10 and n_dropped > 100 and n_kept > 300 and n_long > 30, (n_off, n_dropped, n_kept, n_long)
