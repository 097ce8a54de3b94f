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

np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "ref_vectors.npz"), os.path.getsize(os.path.join(HERE, "ref_vectors.npz")), "bytes")
