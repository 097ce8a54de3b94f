#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.npz: inputs and the REAL reference's outputs for the kernels
on the hot path, by calling the reference functions compiled from /root/reference/lib/aln
(oracle/_ref/libbiscuit_ref.so, see oracle/Makefile).  Run in the build container only; the
committed .npz is data (inputs + expected outputs) and is what pins the oracle and the HIP kernels
on machines where the reference sources are absent.

    python tests/golden/make_vectors.py
"""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import simdata  # noqa: E402
import oracle_lib  # noqa: E402
from biscuit_amd.api import Index  # noqa: E402

R = oracle_lib.ref_lib()
assert R is not None, "oracle/_ref/libbiscuit_ref.so missing: run `make -C oracle`"
u8p, i8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_uint64)


def P(a, t):
    return a.ctypes.data_as(t)


def mats(a, b):
    out = []
    for w in (0, 1, 2):
        m = np.zeros(25, np.int8)
        R.ref_fill_scmat(w, a, b, P(m, i8p))
        out.append(m)
    return out


def ragged(lst, dt):
    off = np.zeros(len(lst) + 1, np.int64)
    for i, x in enumerate(lst):
        off[i + 1] = off[i] + len(x)
    return (np.concatenate(lst).astype(dt) if lst else np.zeros(0, dt)), off


rng = np.random.default_rng(20240928)
out = {}

# ---- scoring matrices, option defaults, small scalar functions
out["scmat"] = np.stack([np.stack(mats(a, b)) for a, b in ((1, 2), (2, 3), (1, 9))])
buf = C.create_string_buffer(4096)
R.ref_opt_defaults(buf, 4096)
out["opt_defaults"] = np.frombuffer(buf.value, dtype=np.uint8)
hk = rng.integers(0, 2**63, 64, dtype=np.uint64)
out["hash_in"] = hk
out["hash_out"] = np.array([R.ref_hash_64(C.c_uint64(int(k))) for k in hk], dtype=np.uint64)
bw = rng.integers(1, 300, (200, 6)).astype(np.int32)
bw[:, 3] = rng.integers(1, 3, 200)
bw[:, 2] = rng.integers(0, 300, 200)
out["infer_bw_in"] = bw
out["infer_bw_out"] = np.array([R.ref_infer_bw(*[int(v) for v in r]) for r in bw], dtype=np.int32)
mq = []
R.ref_approx_mapq_se.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_float]
for _ in range(300):
    score = int(rng.integers(0, 160)); sub = int(rng.integers(0, 160)); csub = int(rng.integers(0, 100)); sub_n = int(rng.integers(0, 5))
    qb = int(rng.integers(0, 30)); qe = qb + int(rng.integers(20, 150)); rb = int(rng.integers(0, 10**6)); re = rb + qe - qb + int(rng.integers(-5, 6))
    seedcov = int(rng.integers(1, 150)); frac = float(np.float32(rng.random() * 0.5))
    q = R.ref_approx_mapq_se(1, 2, 19, C.c_float(50.0), 3, score, sub, csub, sub_n, qb, qe, rb, re, seedcov, C.c_float(frac))
    mq.append((score, sub, csub, sub_n, qb, qe, rb, re, seedcov, frac, q))
out["mapq"] = np.array(mq, dtype=np.float64)

# ---- sorting permutations (ksort.h template) and B-tree behaviour (kbtree.h template)
sorts = []
for n in list(range(0, 40)) + [63, 64, 65, 100, 257, 1000, 5000]:
    for dup in (1, 4, 50):
        keys = rng.integers(0, max(1, n // dup + 1), n).astype(np.int64)
        kv = np.stack([keys, np.arange(n, dtype=np.int64)], 1).copy()
        R.ref_introsort_kv(C.c_int64(n), kv.ctypes.data_as(C.c_void_p))
        kd = np.stack([keys, np.arange(n, dtype=np.int64)], 1).copy()
        R.ref_introsort_kv_desc(C.c_int64(n), kd.ctypes.data_as(C.c_void_p))
        sorts.append((keys, kv[:, 1].copy(), kd[:, 1].copy()))
# adversarial inputs that exhaust the depth budget (comb-sort fallback)
for n in (200, 3000):
    keys = np.concatenate([np.arange(n // 2), np.arange(n // 2)[::-1]]).astype(np.int64)
    kv = np.stack([keys, np.arange(n, dtype=np.int64)], 1).copy()
    R.ref_introsort_kv(C.c_int64(n), kv.ctypes.data_as(C.c_void_p))
    kd = np.stack([keys, np.arange(n, dtype=np.int64)], 1).copy()
    R.ref_introsort_kv_desc(C.c_int64(n), kd.ctypes.data_as(C.c_void_p))
    sorts.append((keys, kv[:, 1].copy(), kd[:, 1].copy()))
out["sort_keys"], out["sort_off"] = ragged([s[0] for s in sorts], np.int64)
out["sort_perm_asc"], _ = ragged([s[1] for s in sorts], np.int64)
out["sort_perm_desc"], _ = ragged([s[2] for s in sorts], np.int64)

bt_ops, bt_res = [], []
for case in range(40):
    t = C.c_void_p(R.ref_bt_new())
    assert R.ref_bt_t(t) == 3
    n = int(rng.integers(1, 200))
    span = int(rng.choice([10, 100, 10**6]))
    ops, res = [], []
    nid = 0
    for _ in range(n):
        pos = int(rng.integers(0, span))
        if rng.random() < 0.5:
            R.ref_bt_put(t, C.c_int64(pos), C.c_int64(nid)); ops.append((0, pos)); res.append(nid); nid += 1
        else:
            o = (C.c_int64 * 2)(); R.ref_bt_interval(t, C.c_int64(pos), o); ops.append((1, pos)); res.append(int(o[0]))
    ids = (C.c_int64 * (nid + 1))()
    k = R.ref_bt_traverse(t, ids, C.c_int64(nid + 1))
    ops.append((2, k)); res.extend(list(ids[:k]))
    R.ref_bt_free(t)
    bt_ops.append(np.array(ops, np.int64).reshape(-1)); bt_res.append(np.array(res, np.int64))
out["bt_ops"], out["bt_ops_off"] = ragged(bt_ops, np.int64)
out["bt_res"], out["bt_res_off"] = ragged(bt_res, np.int64)

# ---- DP kernels
def mutate(seq, sub, ind):
    return simdata.mutate(seq, rng, sub, ind)

ext = []
for it in range(400):
    a = int(rng.choice([1, 1, 1, 2])); b = int(rng.choice([2, 2, 4, 1])); which = int(rng.integers(1, 3))
    qlen = int(rng.integers(1, 200)) if it % 8 else int(rng.integers(300, 900))
    t = rng.integers(0, 4, qlen + int(rng.integers(0, 250))).astype(np.uint8)
    q = mutate(t[:qlen + 10], float(rng.choice([0, 0.02, 0.1, 0.3])), float(rng.choice([0, 0.01, 0.05])))[:qlen]
    if len(q) == 0:
        continue
    if rng.random() < 0.1: q[rng.integers(0, len(q))] = 4
    if rng.random() < 0.1: t[rng.integers(0, len(t))] = 4
    gp = [int(x) for x in rng.choice([[6, 1, 6, 1], [6, 1, 6, 1], [5, 2, 7, 1], [1, 1, 1, 1], [12, 2, 12, 2]])]
    w = int(rng.choice([100, 200, 5, 20, 1])); eb = int(rng.choice([10, 5, 0])); zd = int(rng.choice([100, 100, 20, 0])); h0 = int(rng.integers(1, 200))
    M = mats(a, b)[which]
    o = (C.c_int * 6)()
    R.ref_ksw_extend2(len(q), P(q, u8p), len(t), P(t, u8p), P(M, i8p), gp[0], gp[1], gp[2], gp[3], w, eb, zd, h0, o)
    ext.append((q, t, [a, b, which] + gp + [w, eb, zd, h0], list(o)))
out["ext_q"], out["ext_qoff"] = ragged([e[0] for e in ext], np.uint8)
out["ext_t"], out["ext_toff"] = ragged([e[1] for e in ext], np.uint8)
out["ext_par"] = np.array([e[2] for e in ext], np.int32)
out["ext_out"] = np.array([e[3] for e in ext], np.int32)

sw = []
for it in range(400):
    a = int(rng.choice([1, 1, 1, 2])); b = int(rng.choice([2, 2, 4, 1, 9, 20])); which = int(rng.integers(1, 3))
    qlen = int(rng.integers(5, 300)); tlen = int(rng.integers(5, 1300))
    t = rng.integers(0, 4, tlen).astype(np.uint8)
    if rng.random() < 0.8 and tlen > qlen:
        s = int(rng.integers(0, tlen - qlen + 1))
        q = mutate(t[s:s + qlen], float(rng.choice([0, 0.02, 0.1, 0.3])), float(rng.choice([0, 0.01, 0.05, 0.2])))
        if rng.random() < 0.3 and tlen > 2 * qlen + 20:
            s2 = int(rng.integers(0, tlen - qlen)); t[s2:s2 + qlen // 2] = t[s:s + qlen // 2]
    else:
        q = rng.integers(0, 4, qlen).astype(np.uint8)
    if len(q) < 2:
        continue
    if rng.random() < 0.1: q[rng.integers(0, len(q))] = 4
    gp = [int(x) for x in rng.choice([[6, 1, 6, 1], [6, 1, 6, 1], [5, 2, 7, 1], [1, 1, 1, 1], [2, 1, 3, 1]])]
    xtra = 0x80000 | (0x40000 if rng.random() < 0.7 else 0) | int(rng.choice([19, 30, 10]))
    if len(q) * a < 250 and rng.random() < 0.7: xtra |= 0x10000
    if rng.random() < 0.05: xtra &= ~0x80000
    M = mats(a, b)[which]
    o = (C.c_int * 7)()
    q1, t1 = q.copy(), t.copy()
    R.ref_ksw_align2(len(q1), P(q1, u8p), len(t1), P(t1, u8p), P(M, i8p), gp[0], gp[1], gp[2], gp[3], xtra, o)
    sw.append((q, t, [a, b, which] + gp + [xtra], list(o)))
out["sw_q"], out["sw_qoff"] = ragged([e[0] for e in sw], np.uint8)
out["sw_t"], out["sw_toff"] = ragged([e[1] for e in sw], np.uint8)
out["sw_par"] = np.array([e[2] for e in sw], np.int32)
out["sw_out"] = np.array([e[3] for e in sw], np.int32)

gl = []
for it in range(300):
    a = int(rng.choice([1, 1, 1, 2])); b = int(rng.choice([2, 2, 4, 1])); which = int(rng.integers(1, 3))
    tlen = int(rng.integers(1, 300)); t = rng.integers(0, 4, tlen).astype(np.uint8)
    q = mutate(t, float(rng.choice([0, 0.02, 0.1])), float(rng.choice([0, 0.01, 0.05])))
    if len(q) < 1:
        continue
    gp = [int(x) for x in rng.choice([[6, 1, 6, 1], [6, 1, 6, 1], [5, 2, 7, 1], [1, 1, 1, 1]])]
    w = abs(len(q) - tlen) + int(rng.choice([3, 5, 20, 100, 400]))
    wc = int(rng.random() < 0.8)
    M = mats(a, b)[which]
    cg = (C.c_uint32 * 2048)(); n = C.c_int()
    s = R.ref_ksw_global2(len(q), P(q, u8p), tlen, P(t, u8p), P(M, i8p), gp[0], gp[1], gp[2], gp[3], w, wc, C.byref(n), cg, 2048)
    gl.append((q, t, [a, b, which] + gp + [w, wc], s, np.array(cg[:n.value], np.uint32)))
out["gl_q"], out["gl_qoff"] = ragged([e[0] for e in gl], np.uint8)
out["gl_t"], out["gl_toff"] = ragged([e[1] for e in gl], np.uint8)
out["gl_par"] = np.array([e[2] for e in gl], np.int32)
out["gl_score"] = np.array([e[3] for e in gl], np.int32)
out["gl_cigar"], out["gl_coff"] = ragged([e[4] for e in gl], np.uint32)

# ---- FM index: a fixed 24 kb genome (committed FASTA); index built by the repo's own builder
fa = os.path.join(HERE, "g24k.fa")
if not os.path.exists(fa):
    simdata.write_genome(fa, simdata.make_genome(24000, seed=77, n_contigs=2))
import tempfile
d = tempfile.mkdtemp()
idx = Index.build(fa, d + "/g")
H = {1: C.c_void_p(R.ref_bwt_load((d + "/g.par.bwt").encode(), (d + "/g.par.sa").encode())),
     0: C.c_void_p(R.ref_bwt_load((d + "/g.dau.bwt").encode(), (d + "/g.dau.sa").encode()))}
l_pac = idx.l_pac
pac = np.fromfile(d + "/g.bis.pac", dtype=np.uint8)
ii = np.arange(l_pac)
fwd = ((pac[ii >> 2] >> ((~ii & 3) << 1)) & 3).astype(np.uint8)
# the reference's own BWT construction (is.c) on the two converted texts must agree with the builder's files
for par in (1, 0):
    text = np.concatenate([fwd, (3 - fwd[::-1])]).astype(np.uint8)
    if par: text[text == 1] = 3
    else: text[text == 2] = 0
    T = np.concatenate([text, np.zeros(1, np.uint8)])
    prim = R.ref_is_bwt(P(T, u8p), len(text))
    meta = (C.c_uint64 * 8)(); R.ref_bwt_meta(H[par], meta)
    assert prim == meta[0], (prim, meta[0])
    words = np.fromfile(d + "/g.%s.bwt" % ("par" if par else "dau"), dtype=np.uint32)[10:]
    blocks = words[: (len(words) // 16) * 16].reshape(-1, 16)[:, 8:].reshape(-1)
    sym = ((blocks[:, None] >> (30 - 2 * np.arange(16, dtype=np.uint32))) & 3).reshape(-1)[: len(text)]
    assert (sym == T[: len(text)]).all(), "BWT symbols differ from the reference's is_bwt"
    out["fm_primary_%d" % par] = np.array([prim], np.int64)
fm_reads, fm_par, fm_smem, fm_ss1 = [], [], [], []
for it in range(300):
    L_ = int(rng.integers(20, 200)); s = int(rng.integers(0, l_pac - L_)); rd = fwd[s:s + L_].copy()
    if rng.random() < 0.5: rd = (3 - rd[::-1]).astype(np.uint8)
    k = rng.random(L_) < 0.02
    rd[k] = (rd[k] + rng.integers(1, 4, int(k.sum()))) % 4
    if rng.random() < 0.2: rd[rng.integers(0, L_)] = 4
    par = int(rng.integers(0, 2))
    conv = rd.copy()
    if par: conv[conv == 1] = 3
    else: conv[conv == 2] = 0
    x = int(rng.integers(0, L_)); mi = int(rng.choice([1, 1, 2, 5, 11]))
    o1 = np.zeros(4 * 512, np.uint64); r1 = C.c_int()
    n1 = R.ref_bwt_smem1a(H[par], H[1 - par], L_, P(conv, u8p), x, mi, C.c_uint64(0), P(o1, u64p), 512, C.byref(r1))
    a1 = np.zeros(4, np.uint64)
    q1 = R.ref_bwt_seed_strategy1(H[par], H[1 - par], L_, P(conv, u8p), x, 19, 20, P(a1, u64p))
    fm_reads.append(conv); fm_par.append((par, x, mi, n1, r1.value, q1))
    fm_smem.append(o1[:4 * n1].copy()); fm_ss1.append(a1)
out["fm_reads"], out["fm_roff"] = ragged(fm_reads, np.uint8)
out["fm_par"] = np.array(fm_par, np.int64)
out["fm_smem"], out["fm_soff"] = ragged(fm_smem, np.uint64)
out["fm_ss1"] = np.array(fm_ss1, np.uint64)
ks = np.concatenate([np.arange(0, 200), rng.integers(0, 2 * l_pac + 1, 800)]).astype(np.uint64)
for par in (0, 1):
    occ = np.zeros((len(ks), 4), np.uint64)
    for i, k in enumerate(ks):
        R.ref_bwt_occ4(H[par], C.c_uint64(int(k)), P(occ[i], u64p))
    out["fm_occ4_%d" % par] = occ
    out["fm_sa_%d" % par] = np.array([R.ref_bwt_sa(H[par], C.c_uint64(int(k))) for k in ks[ks >= 1]], dtype=np.uint64)
out["fm_k"] = ks

# ---- K1+K2 as a whole: mem_collect_intv (memchain.c:50-106) is static in a file that does not build here, so its 20-line
# driver is restated below over the REAL bwt_smem1a / bwt_seed_strategy1 (all FM work is the reference's); the result is
# what the seeding kernel must return for the read, interval for interval.  Default options (mem_opt_init).
def collect_intv(par, conv, min_seed_len=19, split_factor=1.5, split_width=10, max_mem_intv=20, start_width=1):
    L_ = len(conv)
    split_len = int(min_seed_len * split_factor + .499)
    mem = []
    o1 = np.zeros(4 * 1024, np.uint64); r1 = C.c_int()

    def smem1(x, mi):
        n1 = R.ref_bwt_smem1a(H[par], H[1 - par], L_, P(conv, u8p), x, mi, C.c_uint64(0), P(o1, u64p), 1024, C.byref(r1))
        assert n1 <= 1024
        return [tuple(int(v) for v in o1[4 * i:4 * i + 4]) for i in range(n1)], r1.value
    x = 0
    while x < L_:                                    # pass 1 (memchain.c:65-73)
        if conv[x] < 4:
            got, x = smem1(x, start_width)
            mem += [m for m in got if (m[3] & 0xffffffff) - (m[3] >> 32) >= min_seed_len]
        else:
            x += 1
    for k in range(len(mem)):                        # pass 2 (memchain.c:76-85)
        st, en = mem[k][3] >> 32, mem[k][3] & 0xffffffff
        if en - st < split_len or mem[k][2] > split_width:
            continue
        got, _ = smem1((st + en) >> 1, mem[k][2] + 1)
        mem += [m for m in got if (m[3] & 0xffffffff) - (m[3] >> 32) >= min_seed_len]
    x = 0
    a1 = np.zeros(4, np.uint64)
    while x < L_:                                    # pass 3 (memchain.c:88-103)
        if conv[x] < 4:
            x = R.ref_bwt_seed_strategy1(H[par], H[1 - par], L_, P(conv, u8p), x, min_seed_len, max_mem_intv, P(a1, u64p))
            if int(a1[2]) > 0:
                mem.append(tuple(int(v) for v in a1))
        else:
            x += 1
    mem.sort(key=lambda m: m[3])                     # ks_introsort(mem_intv): records with equal info are identical
    return np.array(mem, np.uint64).reshape(-1, 4)


fm_all = [collect_intv(int(fm_par[i][0]), fm_reads[i]) for i in range(len(fm_reads))]
out["fm_collect"], out["fm_coff"] = ragged([a.reshape(-1) for a in fm_all], np.uint64)

# ---- I1: the record grammar (kseq_read, kseq.h:182-222, as instantiated by the reference in utils.c:53) on awkward inputs, the
# base-to-code table (bntseq.c:49-66, read from the source as data: bntseq.c does not build here), the SAM header
# (bwa_print_sam_hdr, bwa.c:654-684)
import re  # noqa: E402
import tempfile  # noqa: E402
FQ_CASES = [
    "@r1 first comment\nACGTNacgtn\n+\nIIIIIIIIII\n@r2/1\nAC\nGT\n+r2\nII\nII\n",
    ">fa1 desc here\nACGT\nTTGA\n>fa2\nA\n\n>fa3\n",
    "@q1\nACGT\n+\n@III\n@q2\tc\nGGCC\n+\n+@+@\n",
    "junk before\n@r\r\nACGT\r\n+\r\nIIII\r\n",
    "@trunc\nACGTACGT\n+\nIII",
    "@a\nAC\n+\nII\n@b\nGT\n+\nII",
    "@noseq\n+\n\n@x y z\nA\n+\nI\n",
    "@long " + "c" * 5000 + "\n" + "ACGT" * 3000 + "\n+\n" + "I" * 12000 + "\n",
    "",
]
fq_out = []
with tempfile.TemporaryDirectory() as td:
    for i, text in enumerate(FQ_CASES):
        fn = os.path.join(td, "c%d.fq" % i)
        open(fn, "w").write(text)
        R.ref_kseq_open.restype = C.c_void_p
        R.ref_kseq_open.argtypes = [C.c_char_p]
        R.ref_kseq_close.argtypes = [C.c_void_p]
        R.ref_kseq_next.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        ks = R.ref_kseq_open(fn.encode())
        buf = C.create_string_buffer(1 << 20)
        recs = []
        while True:
            l = R.ref_kseq_next(ks, buf, len(buf))
            if l < 0:
                recs.append("END %d\n" % l)
                break
            recs.append(buf.value.decode())
        R.ref_kseq_close(ks)
        fq_out.append("".join(recs))
out["fq_case"], out["fq_case_off"] = ragged([np.frombuffer(t.encode(), np.uint8) for t in FQ_CASES], np.uint8)
out["fq_recs"], out["fq_recs_off"] = ragged([np.frombuffer(t.encode(), np.uint8) for t in fq_out], np.uint8)

src = open("/root/reference/lib/aln/bntseq.c").read()
body = src[src.index("nst_nt4_table[256]"):]
body = body[body.index("{") + 1:body.index("};")]
nt4 = np.array([int(v) for v in re.findall(r"\d+", body)], np.uint8)
assert len(nt4) == 256
out["nt4_table"] = nt4

HDR_CASES = [
    (["chr1", "chr10", "chr2", "chrM", "Chr3", "1", "chr1_alt"], [1000, 20, 300, 16569, 5, 7, 9], "", "@PG\tID:biscuit\tPN:biscuit\tVN:x\tCL:biscuit align a b"),
    (["b", "a"], [10, 20], "@RG\tID:g\tSM:s", ""),
    (["b", "a"], [10, 20], "@SQ\tSN:b\tLN:10\n@SQ\tSN:a\tLN:20\n@CO\tx", "@PG\tID:p"),
    (["b", "a"], [10, 20], "@CO\thas @SQ\tin the middle of a line", ""),
    (["z%03d" % (i * 37 % 101) for i in range(101)], list(range(1, 102)), "", ""),
]
R.ref_sam_hdr.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
hdr_in, hdr_out = [], []
for names, lens, hl, pg in HDR_CASES:
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    la = (C.c_int * len(lens))(*lens)
    buf = C.create_string_buffer(1 << 16)
    n = R.ref_sam_hdr(len(names), arr, la, hl.encode() if hl else None, pg.encode() if pg else None, buf, len(buf))
    assert n >= 0
    hdr_in.append("\x1e".join(["\x1f".join(names), "\x1f".join(str(x) for x in lens), hl, pg]))
    hdr_out.append(buf.value.decode())
out["hdr_in"], out["hdr_in_off"] = ragged([np.frombuffer(t.encode(), np.uint8) for t in hdr_in], np.uint8)
out["hdr_out"], out["hdr_out_off"] = ragged([np.frombuffer(t.encode(), np.uint8) for t in hdr_out], np.uint8)

# ---- read_clipping (bwamem.c:286-303, with read_identify_adaptor and clip_read_by_quality; static there: compiled through
# oracle/ref_statics.c) and check_paired_read_names (bwamem.c:210-216; only names it accepts -- it exits on the others)
R.ref_read_clipping.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
crng = np.random.default_rng(4242)
clip_in, clip_out = [], []
for k in range(400):
    l = int(crng.integers(30, 160))
    seq = crng.integers(0, 4, l).astype(np.uint8)
    la = int(crng.integers(0, 25)) if k % 3 else 0
    ad = crng.integers(0, 4, la).astype(np.uint8) if la else np.zeros(0, np.uint8)
    u = crng.random()
    if la and u < 0.3:          # the whole adaptor somewhere in the read
        at = int(crng.integers(0, l - la + 1)); seq[at:at + la] = ad
    elif la and u < 0.6:        # a prefix of it at the 3' end
        pre = int(crng.integers(1, la + 1)); seq[l - pre:] = ad[:pre]
    qual = None
    if k % 4:
        q = crng.integers(33 + 20, 33 + 41, l)
        a5, a3 = int(crng.integers(0, 12)), int(crng.integers(0, 12))
        q[:a5] = crng.integers(33, 33 + 12, a5); q[l - a3:] = crng.integers(33, 33 + 12, a3) if a3 else q[l:]
        if k % 17 == 0:
            q[:] = 33 + 2       # nothing passes the quality threshold
        qual = bytes(int(x) for x in q)
    c5, c3, mbq = int(crng.integers(0, 8)) if k % 5 == 0 else 0, int(crng.integers(0, 8)) if k % 7 == 0 else 0, int(crng.choice([0, 0, 10, 20]))
    res = (C.c_int * 5)()
    R.ref_read_clipping(l, seq.ctypes.data_as(C.c_void_p), qual, ad.ctypes.data_as(C.c_void_p) if la else None, la, c5, c3, mbq, res)
    clip_in.append(np.concatenate([[l, la, c5, c3, mbq, 1 if qual else 0], seq, ad, np.frombuffer(qual, np.uint8) if qual else np.zeros(0, np.uint8)]).astype(np.int32))
    clip_out.append(list(res))
out["clip_in"], out["clip_in_off"] = ragged(clip_in, np.int32)
out["clip_out"] = np.array(clip_out, np.int32)
NAME_OK = [("r1", "r1"), ("read/1", "read/2"), ("x.1", "x.2"), ("a_b_c", "a_b_c"), ("q1", "q2"), ("frag.0001/1", "frag.0001/2")]
for n1, n2 in NAME_OK:
    R.ref_check_paired_read_names(n1.encode(), n2.encode())      # returns: accepted by the reference
out["names_ok"] = np.frombuffer("\x1e".join("\x1f".join(p) for p in NAME_OK).encode(), np.uint8)

# ---- the header-inline functions of the pairing / formatting code (mem_alnreg.h:75-144, bwamem.h:200, bntseq.h:92): a generator of its
# own, so that the vectors above stay what they were
prng = np.random.default_rng(5150)
i64p, i32p, u32p = C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_uint32)
L_PAC = 1000000
R.ref_infer_isize.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, i64p]
R.ref_alnreg_isize.argtypes = [C.c_int64, i64p, i64p, i64p]
R.ref_is_proper_pair.argtypes = [C.c_int64, i64p, i64p, C.c_int, C.c_int]
R.ref_get_pri_idx.argtypes = [C.c_double, C.c_int, i32p, i32p, C.c_int]
R.ref_region_depos.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, i32p]
R.ref_get_rlen.argtypes = [C.c_int, u32p]
R.ref_bns_depos.argtypes = [C.c_int64, C.c_int64, i32p]
R.ref_bns_depos.restype = C.c_int64
rows = []
for _ in range(400):   # pos1 pos2 isrev1 isrev2 len1 len2 -> ok isize
    a = [int(prng.integers(0, 2 * L_PAC)), int(prng.integers(0, 2 * L_PAC)), int(prng.integers(0, 2)), int(prng.integers(0, 2)), int(prng.integers(1, 300)), int(prng.integers(1, 300))]
    iz = C.c_int64(-12345)
    ok = R.ref_infer_isize(*a, C.byref(iz))
    rows.append(a + [ok, iz.value if ok else 0])
out["isize_infer"] = np.array(rows, np.int64)
rows = []
for k in range(1200):   # two regions (rid rb re qb qe) around the strand boundary, often close to each other, + low high -> ok isize proper
    def reg(near=None):
        ln = int(prng.integers(20, 200))
        if near is not None and prng.random() < 0.7:   # a mate within a kilobase on the other strand (mirror image), or the same
            mid = (2 * L_PAC - 1 - near) if prng.random() < 0.8 else near
            rb = int(np.clip(mid + prng.integers(-900, 900), 0, 2 * L_PAC - 1))
        else:
            rb = int(prng.integers(0, 2 * L_PAC)) if k % 7 else int(L_PAC + prng.integers(-3, 4))
        qb = int(prng.integers(0, 40))
        return [int(prng.integers(0, 3)) if prng.random() < 0.2 else 1, rb, rb + ln, qb, qb + ln]
    a = reg(); b = reg(a[1])
    low, high = int(prng.integers(-50, 300)), int(prng.integers(300, 1200))
    aa, bb = np.array(a, np.int64), np.array(b, np.int64)
    iz = C.c_int64(-12345)
    ok = R.ref_alnreg_isize(L_PAC, P(aa, i64p), P(bb, i64p), C.byref(iz))
    pp = R.ref_is_proper_pair(L_PAC, P(aa, i64p), P(bb, i64p), low, high)
    rows.append(a + b + [low, high, ok, iz.value if ok else 0, pp])
out["isize_pair"] = np.array(rows, np.int64)
assert 50 < sum(r[-1] for r in rows) < 1150 and 100 < sum(r[-3] for r in rows)   # both answers occur often
rows = []
for _ in range(300):   # XA_drop_ratio, i, then 8 scores and 8 secondary_all -> index
    n = 8
    sc = prng.integers(20, 150, n).astype(np.int32)
    sa = prng.integers(-1, n, n).astype(np.int32)
    ratio = float(np.float32(prng.choice([0.8, 0.5, 1.0, 0.95])))   # the option is a float (mem_opt_t.XA_drop_ratio) widened to the function's double
    if _ % 3 == 0:   # borderline: a[i].score == a[k].score * ratio up to rounding
        sa[:] = 0; sc[0] = 100; sc[1:] = [80, 50, 95, 81, 79, 100, 49]
    i = int(prng.integers(0, n))
    r = R.ref_get_pri_idx(ratio, n, P(sc, i32p), P(sa, i32p), i)
    rows.append([ratio, i] + [float(x) for x in sc] + [float(x) for x in sa] + [r])
out["pri_idx"] = np.array(rows, np.float64)
rows = []
for _ in range(300):   # l_pac offset rb re -> pos is_rev ; and bns_depos of rb
    off = int(prng.integers(0, L_PAC // 2))
    ln = int(prng.integers(1, 300))
    fwd = prng.random() < 0.5
    rb = int(prng.integers(off, L_PAC - ln)) if fwd else int(prng.integers(L_PAC, 2 * L_PAC - off - ln))
    isr, isr2 = C.c_int(-1), C.c_int(-1)
    pos = R.ref_region_depos(L_PAC, off, rb, rb + ln, C.byref(isr))
    dp = R.ref_bns_depos(L_PAC, rb, C.byref(isr2))
    rows.append([L_PAC, off, rb, rb + ln, pos, isr.value, dp, isr2.value])
out["region_depos"] = np.array(rows, np.int64)
cg, cgo = [], []
for _ in range(200):   # CIGARs (op in the low 4 bits: M I D S H and N) -> reference length
    n = int(prng.integers(0, 9))
    c = (prng.integers(1, 200, n).astype(np.uint32) << 4) | prng.integers(0, 6, n).astype(np.uint32)
    cg.append(c)
    cgo.append(R.ref_get_rlen(n, P(np.ascontiguousarray(c), u32p)) if n else R.ref_get_rlen(0, None))
out["rlen_cigar"], out["rlen_cigar_off"] = ragged(cg, np.uint32)
out["rlen_out"] = np.array(cgo, np.int32)

np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "ref_vectors.npz"), os.path.getsize(os.path.join(HERE, "ref_vectors.npz")), "bytes")
