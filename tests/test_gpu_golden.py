"""-m gpu: the HIP kernels against outputs recorded from the REAL reference functions
(tests/golden/ref_vectors.npz, made by tests/golden/make_vectors.py from /root/reference/lib/aln).
The DP kernels read their target from the HBM-resident packed reference, so the recorded target
sequences are concatenated into a scratch genome and indexed with the repo's builder."""
import ctypes as C
import os
import numpy as np
import pytest
import simdata
from biscuit_amd.api import Index, Device, default_opt, EXT_DT, SW_DT, GLB_DT, SEED_DT, SA_DT
from biscuit_amd import _lib as B

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def V():
    return np.load(os.path.join(HERE, "golden", "ref_vectors.npz"))


def _genome_of_targets(tmp, name, tgt, toff):
    """concatenate the N-free targets into one contig; returns (Index, start offset per case or -1)"""
    start = np.full(len(toff) - 1, -1, np.int64)
    parts, at = [], 0
    for i in range(len(toff) - 1):
        t = tgt[toff[i]:toff[i + 1]]
        if len(t) == 0 or (t > 3).any():
            continue
        start[i] = at
        parts.append(t)
        at += len(t)
    g = np.concatenate(parts)
    fa = os.path.join(tmp, name + ".fa")
    simdata.write_genome(fa, [("t", g)])
    return Index.build(fa, os.path.join(tmp, name)), start


def _opt(a, b, gp, zdrop=100):
    o = default_opt()
    o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins, o.zdrop = a, b, gp[0], gp[1], gp[2], gp[3], zdrop
    B.lib().bsx_opt_fill_matrices(C.byref(o))
    return o


@pytest.mark.parametrize("form", ["wavefront_per_job", "quarter_wave_per_job", "lane_per_job"])
def test_extend_kernel_vs_reference_vectors(V, tmp_path, form, tune):
    """ksw_extend2 vectors recorded from the reference against k_extend (a wavefront per job, rows in LDS) and against k_ext4
    (k_ext4.hip: a row of 16 lanes per job, the form the regions path runs; it holds queries up to 255 bases)"""
    quarter = form != "wavefront_per_job"
    if quarter:      # "2": then k_extl (a lane per job) over the same jobs, its answers replacing k_ext4's
        tune("ext4", "1" if form == "quarter_wave_per_job" else "2")
    idx, start = _genome_of_targets(str(tmp_path), "ext", V["ext_t"], V["ext_toff"])
    dev = Device(0); dev.upload_index(idx)
    qo = V["ext_qoff"]
    dev.set_reads(V["ext_q"])
    par = V["ext_par"]
    groups = {}
    for i in range(len(par)):
        if start[i] < 0 or (V["ext_q"][qo[i]:qo[i + 1]] > 4).any():
            continue
        if quarter and qo[i + 1] - qo[i] > 255:
            continue
        a, b, which, od, ed, oi, ei, w, eb, zd, h0 = [int(x) for x in par[i]]
        groups.setdefault((a, b, od, ed, oi, ei, zd), []).append(i)
    n = 0
    for key, ids in groups.items():
        dev.set_opt(_opt(key[0], key[1], key[2:6], key[6]))
        jobs = np.zeros(len(ids), dtype=EXT_DT)
        for k, i in enumerate(ids):
            a, b, which, od, ed, oi, ei, w, eb, zd, h0 = [int(x) for x in par[i]]
            jobs[k] = (start[i], qo[i], qo[i + 1] - qo[i], V["ext_toff"][i + 1] - V["ext_toff"][i], h0, w, eb, 1, 1, 1 if which == 1 else 0, 0)
        res = dev.extend(jobs)
        got = np.stack([res[f] for f in ("score", "qle", "tle", "gtle", "gscore", "max_off")], 1)
        assert (got == V["ext_out"][ids]).all(), key
        n += len(ids)
    assert n > (200 if quarter else 300)
    dev.close()


def test_sw_kernel_vs_reference_vectors(V, tmp_path):
    idx, start = _genome_of_targets(str(tmp_path), "sw", V["sw_t"], V["sw_toff"])
    dev = Device(0); dev.upload_index(idx)
    qo = V["sw_qoff"]
    dev.set_reads(V["sw_q"])
    par = V["sw_par"]
    groups = {}
    for i in range(len(par)):
        if start[i] < 0:
            continue
        a, b, which, od, ed, oi, ei, xtra = [int(x) for x in par[i]]
        groups.setdefault((a, b, od, ed, oi, ei), []).append(i)
    n = 0
    for key, ids in groups.items():
        dev.set_opt(_opt(key[0], key[1], key[2:6]))
        jobs = np.zeros(len(ids), dtype=SW_DT)
        for k, i in enumerate(ids):
            a, b, which, od, ed, oi, ei, xtra = [int(x) for x in par[i]]
            jobs[k] = (start[i], qo[i], qo[i + 1] - qo[i], V["sw_toff"][i + 1] - V["sw_toff"][i], xtra, 1, 1, 0, 1 if which == 1 else 0)
        res = dev.sw(jobs)
        got = np.stack([res[f] for f in ("score", "te", "qe", "score2", "te2", "tb", "qb")], 1)
        bad = np.nonzero((got != V["sw_out"][ids]).any(1))[0]
        assert len(bad) == 0, (key, got[bad[:3]], V["sw_out"][ids][bad[:3]])
        n += len(ids)
    assert n > 300
    dev.close()


def test_global_kernel_vs_reference_vectors(V, tmp_path):
    """k_global (K6) against ksw_global2 outputs recorded from the reference (gl_score / gl_cigar): the kernel folds the band set-up of
    bis_bwa_gen_cigar2 in (bwa.c:325-333: w = max(min((max_gap + |d| + 1) >> 1, w_), |d| + 3)), so the vectors used are the ones whose
    recorded band is the band that rule gives when it is passed as w_."""
    idx, start = _genome_of_targets(str(tmp_path), "gl", V["gl_t"], V["gl_toff"])
    dev = Device(0); dev.upload_index(idx)
    qo, to, co = V["gl_qoff"], V["gl_toff"], V["gl_coff"]
    dev.set_reads(V["gl_q"])
    par = V["gl_par"]
    groups = {}
    for i in range(len(par)):
        if start[i] < 0 or (V["gl_q"][qo[i]:qo[i + 1]] > 3).any():
            continue
        a, b, which, od, ed, oi, ei, w, wc = [int(x) for x in par[i]]
        lq, lt = int(qo[i + 1] - qo[i]), int(to[i + 1] - to[i])
        max_ins = int(float(((lq + 1) >> 1) * a - oi) / ei + 1.)
        max_del = int(float(((lq + 1) >> 1) * a - od) / ed + 1.)
        max_gap = max(max_ins, max_del, 1)
        if w > (max_gap + abs(lt - lq) + 1) >> 1:
            continue    # bis_bwa_gen_cigar2 would narrow this band
        groups.setdefault((a, b, od, ed, oi, ei), []).append(i)
    n = n_cig = 0
    for key, ids in groups.items():
        dev.set_opt(_opt(key[0], key[1], key[2:6]))
        jobs = np.zeros(len(ids), dtype=GLB_DT)
        cig_off = 0
        for k, i in enumerate(ids):
            a, b, which, od, ed, oi, ei, w, wc = [int(x) for x in par[i]]
            cap = int(co[i + 1] - co[i]) + 8
            jobs[k] = (start[i], qo[i], qo[i + 1] - qo[i], to[i + 1] - to[i], w, 1 << 20, 0, 1, cig_off, cap, 1, 1, 1 if which == 1 else 0, wc)
            cig_off += cap
        res, pool = dev.global_(jobs, cig_off)
        for k, i in enumerate(ids):
            assert int(res[k]["score"]) == int(V["gl_score"][i]), (key, i, res[k], V["gl_score"][i])
            if int(par[i][8]):
                nc = int(res[k]["n_cigar"])
                off = int(jobs[k]["cigar_off"])
                assert list(pool[off:off + nc]) == list(V["gl_cigar"][co[i]:co[i + 1]]), (key, i)
                n_cig += 1
        n += len(ids)
    assert n > 120 and n_cig > 80, (n, n_cig)
    dev.close()


def test_fm_kernels_vs_reference_vectors(V, tmp_path):
    """K1-K3 on the committed 24 kb genome against vectors recorded from the real reference: the seeding kernel's interval
    lists == the lists the reference's bwt_smem1a / bwt_seed_strategy1 give when driven as mem_collect_intv drives them
    (memchain.c:50-106; tests/golden/make_vectors.py), interval for interval; every SMEM recorded for a single bwt_smem1a
    call with min_intv 1 is among them; bwt_sa agrees."""
    idx = Index.build(os.path.join(HERE, "golden", "g24k.fa"), str(tmp_path / "g"))
    dev = Device(0); dev.upload_index(idx)
    ro = V["fm_roff"]
    reads = [V["fm_reads"][ro[i]:ro[i + 1]] for i in range(len(ro) - 1)]
    buf, offs = simdata.read_buffer(reads)
    tasks = np.zeros(len(reads), dtype=SEED_DT)
    for i, r in enumerate(reads):
        tasks[i] = (offs[i], len(r), int(V["fm_par"][i][0]))
    opt = default_opt()
    dev.set_opt(opt)
    dev.set_reads(buf)
    iv, off = dev.seed(opt, tasks)
    co = V["fm_coff"] // 4
    assert (np.asarray(off) == co).all()
    assert (iv.reshape(-1) == V["fm_collect"]).all()
    assert int(co[-1]) > 1000
    so, n_in = V["fm_soff"], 0
    for i, par in enumerate(V["fm_par"]):
        if int(par[2]) != 1:      # min_intv of the recorded call: pass 1 runs with 1
            continue
        mine = {tuple(int(v) for v in row) for row in iv[off[i]:off[i + 1]]}
        sm = V["fm_smem"][so[i]:so[i + 1]].reshape(-1, 4)
        for row in sm:
            if (int(row[3]) & 0xffffffff) - (int(row[3]) >> 32) >= 19:
                assert tuple(int(v) for v in row) in mine, (i, row)
                n_in += 1
    assert n_in > 100
    ks = V["fm_k"][V["fm_k"] >= 1]
    for p in (0, 1):
        jobs = np.zeros(len(ks), dtype=SA_DT)
        jobs["k"] = ks; jobs["parent"] = p
        assert (dev.sa(jobs) == V["fm_sa_%d" % p]).all()
    dev.close()
