"""-m gpu parity tests: every HIP kernel, called through the C ABI, against the CPU restatement in
oracle/ (itself pinned to the real reference functions by test_oracle_vs_ref.py).  Bit-exact."""
import numpy as np
import pytest
import simdata
from biscuit_amd.api import default_opt, SEED_DT, SA_DT

pytestmark = pytest.mark.gpu


def _reads(small_index, n_pairs=400, read_len=150, seed=5, **kw):
    import os
    contigs = _contigs(small_index)
    pairs = simdata.make_pairs(contigs, n_pairs, read_len, seed, sub=0.01, indel=0.004, pbat_frac=0.2, chimera_frac=0.05,
                               bad_mate_frac=0.05, n_frac=0.05, **kw)
    seqs = []
    for _, r1, r2 in pairs:
        seqs += [r1, r2]
    return seqs


def _contigs(index):
    """decode the forward genome back from <base>.bis.pac + .ann"""
    pac = np.fromfile(index.base + ".bis.pac", dtype=np.uint8)
    l = index.l_pac
    i = np.arange(l)
    g = (pac[i >> 2] >> ((~i & 3) << 1)) & 3
    out = []
    with open(index.base + ".bis.ann") as f:
        f.readline()
        while True:
            h = f.readline()
            if not h:
                break
            name = h.split()[1]
            off, ln, _ = [int(x) for x in f.readline().split()]
            out.append((name, g[off:off + ln].astype(np.uint8)))
    return out


def _tasks(seqs, offs):
    t = np.zeros(2 * len(seqs), dtype=SEED_DT)
    for i, s in enumerate(seqs):
        for p in (0, 1):
            t[2 * i + p] = (offs[i], len(s), p)
    return t


def test_seed_and_sa_match_oracle(small_index, port, device):
    opt = default_opt()
    seqs = _reads(small_index) + [np.zeros(10, np.uint8), np.full(40, 4, np.uint8), np.zeros(0, np.uint8)]
    seqs[-1] = np.array([0, 1, 2], np.uint8)
    buf, offs = simdata.read_buffer(seqs)
    tasks = _tasks(seqs, offs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    pi, po = port.seed(opt, tasks)
    port.counters(reset=True)
    pi, po = port.seed(opt, tasks)
    device.counters(reset=True)
    di, do = device.seed(opt, tasks)
    assert (po == do).all()
    assert pi.shape == di.shape and (pi == di).all()
    assert int(po[-1]) > len(seqs)          # the test is not vacuous
    pc, dc = port.counters(), device.counters()
    looks, depth = device.seed_table(reset=True)
    # the table of k-mer intervals (seed_tab.hpp) takes most FM extensions off the path: fewer blocks than the reference touches
    assert depth >= 8 and looks > 0 and dc[0] + dc[1] < 0.7 * (pc[0] + pc[1]), (pc, dc, looks, depth)
    # the same lists from every table depth (0: none, every step an FM extension) and from the kernel without the table,
    # whose FM-block touches are the reference's own (the algorithmic-bytes figure of SURVEY 8(d))
    from biscuit_amd import _lib as B_
    try:
        for k in (0, 2, 5, depth - 1):
            B_.tune("seed_tab_k", k)
            device.upload_index(small_index)
            assert device.seed_table()[1] == k
            di, do = device.seed(opt, tasks)
            assert (po == do).all() and (pi == di).all(), k
        B_.tune("seed_form", "classic")
        device.counters(reset=True)
        di, do = device.seed(opt, tasks)
        assert (po == do).all() and (pi == di).all()
        dc = device.counters()
        assert pc[0] == dc[0] and pc[1] == dc[1], (pc, dc)
    finally:
        B_.tune("seed_tab_k", None)
        B_.tune("seed_form", None)
        device.upload_index(small_index)
    assert device.seed_table()[1] == depth
    # K3 on every occurrence the chaining step would look up (capped like memchain.c:325)
    jobs = []
    for t in range(len(tasks)):
        for k in range(po[t], po[t + 1]):
            x0, _, x2, _ = [int(v) for v in pi[k]]
            for j in range(min(x2, 50)):
                jobs.append((x0 + j, int(tasks[t]["parent"]), 0))
    jobs = np.array(jobs, dtype=SA_DT)
    assert (port.sa(jobs) == device.sa(jobs)).all()


def test_seed_overflow_path(small_index, port, device):
    """a poly-A read against a genome with a long A run is pathological; capacity retry must agree"""
    opt = default_opt()
    opt.max_mem_intv = 1 << 30
    seqs = _reads(small_index, n_pairs=50, read_len=100, seed=9)
    buf, offs = simdata.read_buffer(seqs)
    tasks = _tasks(seqs, offs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    pi, po = port.seed(opt, tasks)
    di, do = device.seed(opt, tasks)
    assert (po == do).all() and (pi == di).all()


def _rand_ext_jobs(small_index, seqs, offs, rng, n, long_band=False):
    from biscuit_amd.api import EXT_DT
    l_pac = small_index.l_pac
    jobs = np.zeros(n, dtype=EXT_DT)
    for k in range(n):
        r = int(rng.integers(0, len(seqs)))
        L = len(seqs[r])
        qlen = int(rng.integers(1, L))
        left = rng.random() < 0.5
        tlen = int(rng.integers(1, qlen + 220))
        tpos = int(rng.integers(tlen + 1, 2 * l_pac - tlen - 1))
        if tpos < l_pac <= tpos + tlen or tpos - tlen < l_pac <= tpos:
            tpos = int(rng.integers(tlen + 1, l_pac - tlen - 1))
        jobs[k]["tpos"] = tpos
        jobs[k]["qoff"] = offs[r] + (qlen - 1 if left else L - qlen)
        jobs[k]["qlen"] = qlen
        jobs[k]["tlen"] = tlen
        jobs[k]["h0"] = int(rng.integers(1, 160))
        jobs[k]["w"] = int(rng.choice([100, 100, 200, 5, 30] + ([400, 1000] if long_band else [])))
        jobs[k]["end_bonus"] = int(rng.choice([10, 5, 0]))
        jobs[k]["qdir"] = -1 if left else 1
        jobs[k]["tdir"] = -1 if left else 1
        jobs[k]["parent"] = int(rng.integers(0, 2))
    return jobs


@pytest.mark.parametrize("form", ["wavefront_per_job", "quarter_wave_per_job", "lane_per_job"])
def test_extend_matches_oracle(small_index, port, device, form, tune):
    if form != "wavefront_per_job":      # the rows of 16 lanes the regions path extends with (k_ext4.hip), on plain jobs; "2": then the lane-per-job
        tune("ext4", "1" if form == "quarter_wave_per_job" else "2")   # kernel (k_extl.hip) over the same jobs, its answers replacing the others'
    opt = default_opt()
    rng = np.random.default_rng(11)
    seqs = _reads(small_index, n_pairs=200, read_len=150, seed=12)
    buf, offs = simdata.read_buffer(seqs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    jobs = _rand_ext_jobs(small_index, seqs, offs, rng, 3000)
    # make a third of the jobs "real": target = the read's own sequence region found via seeding
    tasks = _tasks(seqs, offs)
    pi, po = port.seed(opt, tasks)
    k = 0
    from biscuit_amd.api import SA_DT
    for t in range(0, len(tasks), 3):
        if po[t + 1] > po[t] and k < len(jobs):
            x0, _, x2, info = [int(v) for v in pi[po[t]]]
            par = int(tasks[t]["parent"])
            pos = int(port.sa(np.array([(x0, par, 0)], dtype=SA_DT))[0])
            qb, qe = info >> 32, info & 0xffffffff
            L = int(tasks[t]["len"])
            if qe < L and pos + (qe - qb) + (L - qe) + 100 < 2 * small_index.l_pac and not (pos < small_index.l_pac <= pos + L + 120):
                jobs[k]["tpos"] = pos + (qe - qb)
                jobs[k]["qoff"] = int(tasks[t]["qoff"]) + qe
                jobs[k]["qlen"] = L - qe
                jobs[k]["tlen"] = min(L - qe + 100, 2 * small_index.l_pac - (pos + qe - qb) - 1)
                jobs[k]["h0"] = qe - qb
                jobs[k]["w"] = 100
                jobs[k]["end_bonus"] = 10
                jobs[k]["qdir"] = 1
                jobs[k]["tdir"] = 1
                jobs[k]["parent"] = par
                k += 3
    pr = port.extend(jobs)
    dr = device.extend(jobs)
    bad = np.nonzero(pr != dr)[0]
    assert len(bad) == 0, (len(bad), jobs[bad[:3]], pr[bad[:3]], dr[bad[:3]])
    assert (pr["score"] > jobs["h0"]).sum() > 50   # real extensions happened


def test_extend_window_follows_the_band(small_index, port, device, tune):
    """ext_dp_win (ext_dp.hpp; round 6): ksw_extend2 with its rows in five register slots of 64 columns that follow the band -- what the chains ->
    regions launch extends a kilobase read with -- through bsx_extend_batch (ext4=3: a wavefront per job) on queries of up to 1 000 bases: random
    jobs (bands of 11, 61, 201 and 255 columns, h0 up to 600 so that the first row's non-zero entries reach past the first window) and the reads'
    own loci (extensions that run for hundreds of rows and move the window a dozen times), against the CPU restatement job by job."""
    tune("ext4", "3")
    opt = default_opt()
    rng = np.random.default_rng(2026)
    seqs = _reads(small_index, n_pairs=60, read_len=1000, seed=21, frag=(1100, 1500))
    buf, offs = simdata.read_buffer(seqs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    jobs = _rand_ext_jobs(small_index, seqs, offs, rng, 1500)
    for k in range(len(jobs)):
        jobs[k]["w"] = int(rng.choice([100, 100, 5, 30, 127]))
        if k % 3 == 0:
            jobs[k]["h0"] = int(rng.integers(1, 600))
    # a third of the jobs "real": the target is the read's own locus, found by seeding (both directions)
    tasks = _tasks(seqs, offs)
    pi, po = port.seed(opt, tasks)
    from biscuit_amd.api import SA_DT
    k = 1
    for t in range(len(tasks)):
        if po[t + 1] > po[t] and k < len(jobs):
            # the longest interval of the strand search with a single occurrence
            best = None
            for q in range(po[t], po[t + 1]):
                x0, _, x2, info = [int(v) for v in pi[q]]
                if x2 == 1 and (best is None or (info & 0xffffffff) - (info >> 32) > best[2] - best[1]):
                    best = (x0, info >> 32, info & 0xffffffff)
            if best is None:
                continue
            x0, qb, qe = best
            par = int(tasks[t]["parent"])
            pos = int(port.sa(np.array([(x0, par, 0)], dtype=SA_DT))[0])
            L = int(tasks[t]["len"])
            l_pac = small_index.l_pac
            if qe < L and pos + L + 150 < 2 * l_pac and not (pos - qb - 150 < l_pac <= pos + L + 150) and pos - qb - 150 > 0:
                right = k % 2 == 1
                if right:
                    jobs[k]["tpos"] = pos + (qe - qb); jobs[k]["qoff"] = int(tasks[t]["qoff"]) + qe; jobs[k]["qlen"] = L - qe
                    jobs[k]["tlen"] = L - qe + 100; jobs[k]["qdir"] = 1; jobs[k]["tdir"] = 1
                elif qb > 0:
                    jobs[k]["tpos"] = pos - 1; jobs[k]["qoff"] = int(tasks[t]["qoff"]) + qb - 1; jobs[k]["qlen"] = qb
                    jobs[k]["tlen"] = qb + 100; jobs[k]["qdir"] = -1; jobs[k]["tdir"] = -1
                else:
                    continue
                jobs[k]["h0"] = qe - qb
                jobs[k]["w"] = 100
                jobs[k]["end_bonus"] = 10
                jobs[k]["parent"] = par
                k += 3
    pr = port.extend(jobs)
    dr = device.extend(jobs)
    bad = np.nonzero(pr != dr)[0]
    assert len(bad) == 0, (len(bad), jobs[bad[:3]], pr[bad[:3]], dr[bad[:3]])
    long_rows = (pr["tle"] > 300).sum()
    assert long_rows > 20, long_rows     # extensions that moved the window several times
    assert (jobs["qlen"] > 320).sum() > 500


@pytest.mark.parametrize("form", ["quarter_wave_per_job", "lane_per_job"])
def test_extend_narrow_jobs_quarter_wave(small_index, port, device, tune, form, capfd):
    """k_ext4 (k_ext4.hip: a row of 16 lanes per job) on the jobs the regions path is full of -- extensions from chance matches of
    19..26 bases: scores that decay, bands of a dozen columns, queries of any length, both directions and strands, targets on either side of
    the forward-reverse boundary, a few real continuations (the read's own locus) that run past the 48 rows of reference bases a row holds --
    against the CPU restatement job by job.  lane_per_job: k_extl (k_extl.hip: a lane per job, the row a window of 64 columns in registers), which
    has to answer most of them itself."""
    import re
    tune("ext4", "1" if form == "quarter_wave_per_job" else "2")
    opt = default_opt()
    rng = np.random.default_rng(77)
    seqs = _reads(small_index, n_pairs=300, read_len=150, seed=13)
    buf, offs = simdata.read_buffer(seqs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    jobs = _rand_ext_jobs(small_index, seqs, offs, rng, 20000)
    jobs["h0"] = rng.integers(19, 27, len(jobs))
    jobs["w"] = rng.choice([100, 100, 100, 200, 40], len(jobs))
    jobs["end_bonus"] = rng.choice([10, 10, 5], len(jobs))
    # some with a target that continues the read for a while
    loci = _locus_of(small_index, port, opt, seqs, offs)
    for k, (r, par, pos, qb) in enumerate(loci[:400]):
        L = len(seqs[r])
        qe = qb + 20
        if qe + 5 >= L or pos + L + 130 >= 2 * small_index.l_pac or pos < small_index.l_pac <= pos + L + 130:
            continue
        jobs[k]["tpos"] = pos + 20; jobs[k]["qoff"] = offs[r] + qe; jobs[k]["qlen"] = L - qe
        jobs[k]["tlen"] = L - qe + 100; jobs[k]["h0"] = 20; jobs[k]["w"] = 100; jobs[k]["end_bonus"] = 10
        jobs[k]["qdir"] = 1; jobs[k]["tdir"] = 1; jobs[k]["parent"] = par
    # and continuations with mismatches (a diverged copy of a repeat: the scores neither take off nor die, the band stays a few dozen columns
    # wide over many rows and the lane kernel's window has to follow it): the reads' buffer is mutated once the loci are known
    mut = buf.copy()
    pick = rng.random(len(mut)) < 0.06
    mut[pick] = (mut[pick] + rng.integers(1, 4, int(pick.sum()))) & 3
    for be in (port, device):
        be.set_reads(mut)
    for k, (r, par, pos, qb) in enumerate(loci[400:1400]):
        k += 400
        L = len(seqs[r])
        qe = qb + int(rng.integers(19, 30))
        if qe + 5 >= L or pos + L + 130 >= 2 * small_index.l_pac or pos < small_index.l_pac <= pos + L + 130 or qb < 8:
            continue
        if k & 1:   # to the right of the seed
            jobs[k]["tpos"] = pos + (qe - qb); jobs[k]["qoff"] = offs[r] + qe; jobs[k]["qlen"] = L - qe
            jobs[k]["tlen"] = L - qe + 100; jobs[k]["qdir"] = 1; jobs[k]["tdir"] = 1
        else:       # to its left
            jobs[k]["tpos"] = pos - 1; jobs[k]["qoff"] = offs[r] + qb - 1; jobs[k]["qlen"] = qb
            jobs[k]["tlen"] = min(qb + 100, pos - (small_index.l_pac if pos >= small_index.l_pac else 0)); jobs[k]["qdir"] = -1; jobs[k]["tdir"] = -1
        jobs[k]["h0"] = qe - qb; jobs[k]["w"] = 100; jobs[k]["end_bonus"] = 10; jobs[k]["parent"] = par
    jobs = jobs[jobs["tlen"] > 0]
    pr = port.extend(jobs)
    capfd.readouterr()
    dr = device.extend(jobs)
    err = capfd.readouterr().err
    bad = np.nonzero(pr != dr)[0]
    assert len(bad) == 0, (len(bad), jobs[bad[:3]], pr[bad[:3]], dr[bad[:3]])
    assert (pr["tle"] > 48).sum() > 20    # extensions that refill their reference bases
    if form == "lane_per_job":
        m = re.search(r"(\d+) jobs, (\d+) left to k_ext4", err)
        assert m and int(m.group(1)) == len(jobs) and int(m.group(2)) < len(jobs) // 3, err[-300:]


def _locus_of(small_index, port, opt, seqs, offs):
    """(read index, parent, forward-reverse position of the read's first SMEM, qb) for reads that seed"""
    from biscuit_amd.api import SA_DT
    tasks = _tasks(seqs, offs)
    pi, po = port.seed(opt, tasks)
    out = []
    for t in range(len(tasks)):
        if po[t + 1] > po[t]:
            x0, _, x2, info = [int(v) for v in pi[po[t]]]
            pos = int(port.sa(np.array([(x0, int(tasks[t]["parent"]), 0)], dtype=SA_DT))[0])
            out.append((t // 2, int(tasks[t]["parent"]), pos, info >> 32))
    return out


def test_sw_matches_oracle(small_index, port, device):
    from biscuit_amd.api import SW_DT
    opt = default_opt()
    rng = np.random.default_rng(21)
    seqs = _reads(small_index, n_pairs=150, read_len=150, seed=22) + [s for _, s in simdata.make_single(_contigs(small_index), 20, 700, 23)]
    buf, offs = simdata.read_buffer(seqs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    loci = _locus_of(small_index, port, opt, seqs, offs)
    l_pac = small_index.l_pac
    jobs = []
    for (r, par, pos, qb) in loci:
        L = len(seqs[r])
        for rep in range(2):
            # window around the read's locus, as in mate rescue / seed SW
            tlen = int(rng.integers(L // 2, L + 1100))
            tpos = max(pos - qb - int(rng.integers(0, 600)), 0 if pos < l_pac else l_pac)
            lim = l_pac if pos < l_pac else 2 * l_pac
            tlen = min(tlen, lim - tpos)
            if tlen < 20:
                continue
            xtra = 0x80000 | (0x40000 if rng.random() < 0.8 else 0) | 19
            if L < 250 and rng.random() < 0.8:
                xtra |= 0x10000
            qlen = L if rng.random() < 0.7 else int(rng.integers(20, min(L, 199)))
            jobs.append((tpos, offs[r], qlen, tlen, xtra, 1, 1, 0, par))
            if rep == 0:  # reverse-complemented query against the other strand's coordinates
                jobs.append((2 * l_pac - (tpos + tlen), offs[r] + L - 1, L, tlen, xtra, -1, 1, 1, 1 - par))
    jobs = np.array(jobs, dtype=SW_DT)
    assert len(jobs) > 300
    pr = port.sw(jobs)
    dr = device.sw(jobs)
    bad = np.nonzero(pr != dr)[0]
    assert len(bad) == 0, (len(bad), jobs[bad[:3]], pr[bad[:3]], dr[bad[:3]])
    assert (pr["score"] > 60).sum() > 100 and (pr["score2"] > 0).sum() > 5 and (pr["qb"] >= 0).sum() > 100


def test_global_matches_oracle(small_index, port, device):
    from biscuit_amd.api import GLB_DT
    opt = default_opt()
    rng = np.random.default_rng(31)
    # 150 bp, 900 bp and a few 2.2 kb reads: the three size classes of the kernel (the last one walks its MD with one lane from HBM)
    seqs = _reads(small_index, n_pairs=200, read_len=150, seed=32) + [s for _, s in simdata.make_single(_contigs(small_index), 30, 900, 33)] \
        + [s for _, s in simdata.make_single(_contigs(small_index), 6, 2200, 34)]
    buf, offs = simdata.read_buffer(seqs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    loci = _locus_of(small_index, port, opt, seqs, offs)
    l_pac = small_index.l_pac
    jobs = []
    cig_off = 0
    for (r, par, pos, qb) in loci:
        L = len(seqs[r])
        lo = 0 if pos < l_pac else l_pac
        hi = l_pac if pos < l_pac else 2 * l_pac
        rb = pos - qb
        for rep in range(2):
            d0, d1 = (0, 0) if rep == 0 else (int(rng.integers(-6, 7)), int(rng.integers(-6, 7)))
            b, e = max(rb + d0, lo), min(rb + L + d1, hi)
            if e - b < 10:
                continue
            want = int(rng.random() < 0.85)
            w0 = int(rng.choice([0, 0, 3, 10, 40, 100]))
            ntry = 3 if want else 1
            cap = 64
            rev = b >= l_pac  # reverse-strand hits are aligned on reversed sequences (bwa.c:307-312)
            jobs.append((e - 1 if rev else b, offs[r] + (L - 1 if rev else 0), L, e - b, w0, 400, int(rng.integers(L - 40, L + 1)), ntry,
                         cig_off, cap, -1 if rev else 1, -1 if rev else 1, par, want))
            cig_off += cap
    jobs = np.array(jobs, dtype=GLB_DT)
    assert len(jobs) > 300
    pr, pp = port.global_(jobs, cig_off)
    dr, dp = device.global_(jobs, cig_off)
    bad = np.nonzero(pr != dr)[0]
    assert len(bad) == 0, (len(bad), jobs[bad[:3]], pr[bad[:3]], dr[bad[:3]])
    for k in range(len(jobs)):
        n = int(pr[k]["n_cigar"])
        if n > 0:
            o = int(jobs[k]["cigar_off"])
            assert (pp[o:o + n] == dp[o:o + n]).all(), k
    assert (pr["n_cigar"] > 1).sum() > 20
    # the second half of bis_bwa_gen_cigar2 (lib/aln/bwa.c:342-418) on the device: NM, MD, ZC, ZR of the same jobs against a
    # restatement of that loop written here (the CIGARs were just shown to be the checker's)
    tr, tp, tags, mds = device.global_tags(jobs, cig_off)
    assert (tr == dr).all() and (tp == dp).all()
    pac = np.fromfile(small_index.base + ".bis.pac", dtype=np.uint8)

    def ref_base(p):
        if p >= l_pac:
            p = 2 * l_pac - 1 - p
            return 3 - int((pac[p >> 2] >> ((~p & 3) << 1)) & 3)
        return int((pac[p >> 2] >> ((~p & 3) << 1)) & 3)
    n_checked = n_conv = n_del = 0
    for k in range(len(jobs)):
        J = jobs[k]
        n = int(dr[k]["n_cigar"])
        if not J["want_cigar"] or n <= 0:
            assert mds[k] is None
            continue
        o = int(J["cigar_off"])
        q = [int(buf[int(J["qoff"]) + i * int(J["qdir"])]) for i in range(int(J["qlen"]))]
        t = [ref_base(int(J["tpos"]) + i * int(J["tdir"])) for i in range(int(J["tlen"]))]
        int2base = "TGCAN" if J["tdir"] < 0 else "ACGTN"     # bwa.c:345: the forward-strand base of a reversed alignment
        parent = int(J["use_ct"])
        x = y = u = n_mm = n_gap = zc = zr = 0
        md = ""
        for c in range(n):
            op, ln = int(dp[o + c]) & 0xf, int(dp[o + c]) >> 4
            if op == 0:
                for i in range(ln):
                    a, b = q[x + i], t[y + i]
                    if a == b:
                        zr += (a == 1) if parent else (a == 2)
                        u += 1
                    else:
                        md += "%d%s" % (u, int2base[b])
                        u = 0
                        if (parent and a == 3 and b == 1) or (not parent and a == 0 and b == 2):
                            zc += 1
                        else:
                            n_mm += 1
                x += ln
                y += ln
            elif op == 2:
                if 0 < c < n - 1:
                    md += "%d^%s" % (u, "".join(int2base[b] for b in t[y:y + ln]))
                    u = 0
                    n_gap += ln
                    n_del += 1
                y += ln
            elif op == 1:
                x += ln
                n_gap += ln
        md += "%d" % u
        assert mds[k] == md.encode() + b"\0", (k, mds[k], md)
        assert (int(tags[k]["NM"]), int(tags[k]["ZC"]), int(tags[k]["ZR"]), int(tags[k]["bss_u"])) == (n_mm + n_gap, zc, zr, int(zc == 0)), k
        n_checked += 1
        n_conv += zc
    assert n_checked > 250 and n_conv > 1000 and n_del > 5


def test_oversize_jobs_take_the_large_classes(small_index, port, device):
    """Jobs beyond the LDS-resident classes do not fail their batch: extension and global alignment of queries longer than
    16384 bases keep their DP rows in HBM, local alignment of queries up to 3072 bases runs with 48 register slots per lane.
    Same results as the CPU restatement, mixed into batches with ordinary jobs."""
    from biscuit_amd.api import EXT_DT, SW_DT, GLB_DT
    opt = default_opt()
    rng = np.random.default_rng(77)
    contigs = _contigs(small_index)
    big = [s for _, s in simdata.make_single(contigs, 3, 20000, 71, sub=0.02, indel=0.002)]
    mid = [s for _, s in simdata.make_single(contigs, 6, 2500, 72)]
    seqs = big + mid + _reads(small_index, n_pairs=20, read_len=150, seed=73)
    buf, offs = simdata.read_buffer(seqs)
    for be in (port, device):
        be.set_opt(opt)
        be.set_reads(buf)
    loci = {r: (par, pos, qb) for (r, par, pos, qb) in _locus_of(small_index, port, opt, seqs, offs)}
    l_pac = small_index.l_pac
    # ---- extension: the whole long read to the right of its first seed, plus ordinary jobs around it
    jobs = list(_rand_ext_jobs(small_index, seqs[len(big) + len(mid):], offs[len(big) + len(mid):], rng, 50))
    n_big = 0
    for r in range(len(big)):
        if r not in loci:
            continue
        par, pos, qb = loci[r]
        L = len(seqs[r])
        hi = l_pac if pos < l_pac else 2 * l_pac
        q0 = qb + 19
        if pos + 19 + (L - q0) + 300 >= hi:
            continue
        j = np.zeros(1, dtype=EXT_DT)[0]
        j["tpos"] = pos + 19; j["qoff"] = offs[r] + q0; j["qlen"] = L - q0; j["tlen"] = L - q0 + 200
        j["h0"] = 19; j["w"] = 100; j["end_bonus"] = 5; j["qdir"] = 1; j["tdir"] = 1; j["parent"] = par
        assert j["qlen"] > 16384
        jobs.append(j)
        n_big += 1
    assert n_big >= 1
    jobs = np.array(jobs, dtype=EXT_DT)
    pr, dr = port.extend(jobs), device.extend(jobs)
    assert (pr == dr).all(), (jobs[pr != dr][:2], pr[pr != dr][:2], dr[pr != dr][:2])
    assert (pr["score"][-n_big:] > 5000).all()
    # ---- global alignment with traceback and tags of the long reads against their loci
    gj, cig_off = [], 0
    for r in list(range(len(big))) + list(range(len(big) + len(mid), len(big) + len(mid) + 10)):
        if r not in loci:
            continue
        par, pos, qb = loci[r]
        L = len(seqs[r])
        lo, hi = (0, l_pac) if pos < l_pac else (l_pac, 2 * l_pac)
        b, e = max(pos - qb, lo), min(pos - qb + L + 3, hi)
        rev = b >= l_pac
        cap = 4096
        gj.append((e - 1 if rev else b, offs[r] + (L - 1 if rev else 0), L, e - b, 100, 400, L, 3, cig_off, cap, -1 if rev else 1, -1 if rev else 1, par, 1))
        cig_off += cap
    gj = np.array(gj, dtype=GLB_DT)
    assert (gj["qlen"] > 16384).sum() >= 1
    pr, pp = port.global_(gj, cig_off)
    tr, tp, tags, mds = device.global_tags(gj, cig_off)
    assert (pr == tr).all(), (pr[pr != tr][:2], tr[pr != tr][:2])
    for k in range(len(gj)):
        n, o = int(pr[k]["n_cigar"]), int(gj[k]["cigar_off"])
        assert n > 0 and (pp[o:o + n] == tp[o:o + n]).all(), k
    dr, dp = device.global_(gj, cig_off)
    assert (dr == pr).all()
    # ---- local alignment: queries of 2500 bases (mate rescue of long paired reads)
    sj = []
    for r in range(len(big), len(big) + len(mid)):
        if r not in loci:
            continue
        par, pos, qb = loci[r]
        L = len(seqs[r])
        lo, hi = (0, l_pac) if pos < l_pac else (l_pac, 2 * l_pac)
        tpos = max(pos - qb - 300, lo)
        tlen = min(L + 700, hi - tpos)
        sj.append((tpos, offs[r], L, tlen, 0x80000 | 0x40000 | 19, 1, 1, 0, par))
    for r in range(len(big) + len(mid), len(big) + len(mid) + 10):
        if r in loci:
            par, pos, qb = loci[r]
            lo, hi = (0, l_pac) if pos < l_pac else (l_pac, 2 * l_pac)
            tpos = max(pos - qb - 200, lo)
            sj.append((tpos, offs[r], len(seqs[r]), min(600, hi - tpos), 0x80000 | 0x40000 | 0x10000 | 19, 1, 1, 0, par))
    sj = np.array(sj, dtype=SW_DT)
    assert (sj["qlen"] > 1024).sum() >= 3
    pr, dr = port.sw(sj), device.sw(sj)
    assert (pr == dr).all(), (sj[pr != dr][:2], pr[pr != dr][:2], dr[pr != dr][:2])
    assert (pr["score"][sj["qlen"] > 1024] > 1000).all()
