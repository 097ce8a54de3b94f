"""oracle/backhalf.py -- TEST INFRASTRUCTURE, not product code.

A second, independent restatement of host-logic functions of the reference's back half, written from the reference's
source lines (not from csrc/host/region.c) in plain Python for small cases:

    sort_dedup        mem_sort_deduplicate + mem_test_reg_concatenation up to its alignment   lib/aln/mem_alnreg.c:63-202
    matesw            mem_alnreg_matesw + mem_alnreg_matesw_core (SW by the real ksw_align2)   lib/aln/mem_alnreg.c:385-513
    format_sam        mem_alnreg_formatSAM with mem_alnreg_tagSA and mem_alnreg_tagXAXB: one SAM line     lib/aln/mem_alnreg_format.c:126-436
    setsam_post       mem_alnreg_setSAM after its alignment: position, strand, squeezed deletions, clips   lib/aln/mem_alnreg_format.c:79-120
    strand_searches   bis_worker1: which converted index a read is searched against, in order               lib/aln/bwamem.c:311-376
    flt_chained_seeds mem_flt_chained_seeds + mem_seed_sw (SW by the real ksw_align2)                                lib/aln/memchain.c:501-568
    reg2sam_pe        mem_reg2sam_pe, mem_reg2sam_pe_nopairing, mem_alnreg_select_format up to the text (which records are written,
                      with which flag / mapq / mate; mem_approx_mapq_se is the real function)          lib/aln/mem_alnreg_format.c:445-696
    mark_primary_se   mem_mark_primary_se + mem_mark_primary_se_core   lib/aln/mem_alnreg.c:252-380
    pestat            cal_sub + mem_pestat                             lib/aln/mem_pair.c:41-146
    pair              mem_pair                                         lib/aln/mem_pair.c:149-270

The reference files themselves cannot be compiled here (they include wzmisc.h from a repository that is not under
/root/reference), so these rows stay "unpinned"; what this file adds is that the product's
implementation (a histogram instead of a sort in pestat, qsort-style generic introsort, its own key packing) is no longer
compared only with itself.  The sorts of the last three functions order by keys that are unique (hash_64 of distinct ids; keys
that embed the element's index), so klib's introsort and Python's sort give the same permutation; sort_dedup's keys are not,
and it takes the permutation from the real klib template (oracle/_ref) through a callback.

Regions are dicts with the mem_alnreg_t fields the functions read or write: rb re qb qe rid score is_alt bss (+ sub sub_n
alt_sc secondary secondary_all hash, written here).  opt is a dict of the mem_opt_t fields used.  Floats: opt["mask_level"]
is the C float; products with ints are formed in single precision where the C code does (float * int), in double elsewhere.
"""
import math
import struct

M64 = (1 << 64) - 1
INT_MAX = 2147483647


def hash_64(key):   # lib/aln/utils.h:107-117
    key &= M64
    key = (key + (~(key << 32) & M64)) & M64
    key ^= key >> 22
    key = (key + (~(key << 13) & M64)) & M64
    key ^= key >> 8
    key = (key + (key << 3)) & M64
    key ^= key >> 15
    key = (key + (~(key << 27) & M64)) & M64
    key ^= key >> 31
    return key


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def _sig_overlap(a, b, mask_level):
    """the overlap test shared by mem_mark_primary_se_core (mem_alnreg.c:268-272) and cal_sub (mem_pair.c:49-54):
    `e_min - b_max >= min_l * opt->mask_level` is int >= int * float, evaluated in single precision"""
    b_max = max(a["qb"], b["qb"])
    e_min = min(a["qe"], b["qe"])
    if e_min > b_max:
        min_l = min(a["qe"] - a["qb"], b["qe"] - b["qb"])
        return f32(float(e_min - b_max)) >= f32(f32(float(min_l)) * f32(mask_level))
    return False


def _mark_core(opt, n_mark, regs, z):   # mem_alnreg.c:252-288
    tmp = max(opt["a"] + opt["b"], opt["o_del"] + opt["e_del"], opt["o_ins"] + opt["e_ins"])
    del z[:]
    z.append(0)
    for i in range(1, n_mark):
        a = regs[i]
        hit = None
        for k in z:
            b = regs[k]
            if _sig_overlap(a, b, opt["mask_level"]):
                if b["sub"] == 0:
                    b["sub"] = a["score"]
                if b["score"] - a["score"] <= tmp and (b["is_alt"] or not a["is_alt"]):
                    b["sub_n"] += 1
                hit = k
                break
        if hit is None:
            z.append(i)
        else:
            a["secondary"] = hit


def mark_primary_se(opt, regs, rid):   # mem_alnreg.c:290-380; returns n_pri, regs reordered in place
    n = len(regs)
    if n == 0:
        return 0
    n_pri = 0
    for i, p in enumerate(regs):
        p["sub"] = p["alt_sc"] = 0
        p["secondary"] = p["secondary_all"] = -1
        p["hash"] = hash_64(rid + i)
        if not p["is_alt"]:
            n_pri += 1
    regs.sort(key=lambda p: (-p["score"], p["is_alt"], p["hash"]))      # alnreg_hlt
    z = []
    _mark_core(opt, n, regs, z)
    for i, p in enumerate(regs):
        p["secondary_all"] = i
        if not p["is_alt"] and p["secondary"] >= 0 and regs[p["secondary"]]["is_alt"]:
            p["alt_sc"] = regs[p["secondary"]]["score"]
    if 0 < n_pri < n:
        regs.sort(key=lambda p: (p["is_alt"], -p["score"], p["hash"]))  # alnreg_hlt2
        zmap = [0] * n
        for i, p in enumerate(regs):
            zmap[p["secondary_all"]] = i
        for p in regs:
            if p["secondary"] >= 0:
                p["secondary_all"] = zmap[p["secondary"]]
                if p["is_alt"]:
                    p["secondary"] = INT_MAX
            else:
                p["secondary_all"] = -1
        for i in range(n_pri):
            regs[i]["sub"] = 0
            regs[i]["secondary"] = -1
        _mark_core(opt, n_pri, regs, z)
    else:
        for p in regs:
            p["secondary_all"] = p["secondary"]
    return n_pri


# ---------------------------------------------------------------------------------------------------------------------
def infer_isize(pos1, pos2, isrev1, isrev2, len1, len2):   # mem_alnreg.h:75-83
    if isrev1 and not isrev2:
        return pos1 - pos2 + len1
    if isrev2 and not isrev1:
        return pos2 - pos1 + len2
    return None


def alnreg_isize(l_pac, r1, r2):   # mem_alnreg.h:86-93 (note: strictly greater than l_pac)
    if r1["rid"] != r2["rid"]:
        return None
    isrev1, isrev2 = r1["rb"] > l_pac, r2["rb"] > l_pac
    pos1 = (l_pac << 1) - 1 - r1["rb"] if isrev1 else r1["rb"]
    pos2 = (l_pac << 1) - 1 - r2["rb"] if isrev2 else r2["rb"]
    return infer_isize(pos1, pos2, isrev1, isrev2, r1["qe"] - r1["qb"], r2["qe"] - r2["qb"])


def cal_sub(opt, regs):   # mem_pair.c:41-57
    best = regs[0]
    for p in regs[1:]:
        if _sig_overlap(p, best, opt["mask_level"]):
            return p["score"]
    return opt["min_seed_len"] * opt["a"]


def pestat(opt, l_pac, reads):   # mem_pair.c:60-146; reads = list of region lists, pairs (2i, 2i+1)
    MIN_RATIO, MIN_DIR_CNT, OUTLIER_BOUND, MAPPING_BOUND, MAX_STDDEV = 0.8, 10, 2.0, 3.0, 4.0
    isize = []
    for i in range(len(reads) >> 1):
        r0, r1 = reads[2 * i], reads[2 * i + 1]
        if not r0 or not r1:
            continue
        b0, b1 = r0[0], r1[0]
        if cal_sub(opt, r0) > MIN_RATIO * b0["score"]:
            continue
        if cal_sub(opt, r1) > MIN_RATIO * b1["score"]:
            continue
        if b0["rid"] != b1["rid"] or b0["bss"] != b1["bss"]:
            continue
        ins = alnreg_isize(l_pac, b0, b1)
        if ins is not None and -opt["max_ins"] <= ins <= opt["max_ins"]:
            isize.append(ins)
    pes = {"low": 0, "high": 0, "failed": 0, "avg": 0.0, "std": 0.0}
    if len(isize) < MIN_DIR_CNT:
        pes["failed"] = 1
        return pes
    isize.sort()
    n = len(isize)
    p25, p50, p75 = isize[int(.25 * n + .499)], isize[int(.50 * n + .499)], isize[int(.75 * n + .499)]
    low = int(p25 - OUTLIER_BOUND * (p75 - p25) + .499)
    high = int(p75 + OUTLIER_BOUND * (p75 - p25) + .499)
    avg, x = 0.0, 0
    for v in isize:
        if low <= v <= high:
            avg += v
            x += 1
    avg /= x
    std = 0.0
    for v in isize:
        if low <= v <= high:
            std += (v - avg) * (v - avg)
    std = math.sqrt(std / x)
    low = int(p25 - MAPPING_BOUND * (p75 - p25) + .499)
    high = int(p75 + MAPPING_BOUND * (p75 - p25) + .499)
    if low > avg - MAX_STDDEV * std:
        low = int(avg - MAX_STDDEV * std + .499)
    if high < avg + MAX_STDDEV * std:
        high = int(avg + MAX_STDDEV * std + .499)
    pes.update(low=low, high=high, avg=avg, std=std)
    return pes


# ---------------------------------------------------------------------------------------------------------------------
def region_depos(l_pac, ann_offset, reg):   # mem_alnreg.h:139-144 over bns_depos
    pos = reg["rb"] if reg["rb"] < l_pac else reg["re"] - 1
    if pos >= l_pac:
        pos = (l_pac << 1) - 1 - pos
    return pos - ann_offset[reg["rid"]]


def pair(opt, l_pac, ann_offset, pes, regs_pair, n_pri, rid):   # mem_pair.c:149-270 -> (score, sub, n_sub, z0, z1)
    v = []
    for r in range(2):
        for i in range(n_pri[r]):
            p = regs_pair[r][i]
            x = ((p["bss"] << 63) | (p["rid"] << 32) | (region_depos(l_pac, ann_offset, p) & M64)) & M64
            y = ((p["score"] << 32) | (i << 2) | ((1 if p["rb"] >= l_pac else 0) << 1) | r) & M64
            v.append((x, y, p["qe"] - p["qb"]))
    v.sort(key=lambda t: (t[0], t[1]))
    proper = []
    hi_lo = max(pes["low"], pes["high"])
    for i in range(len(v)):
        for k in range(i - 1, -1, -1):
            if v[i][0] >> 32 != v[k][0] >> 32:
                break
            if v[i][0] >> 63 != v[k][0] >> 63:
                break
            if (v[i][0] & 0xffffffff) - (v[k][0] & 0xffffffff) > hi_lo:
                break
            if (v[i][1] & 1) == (v[k][1] & 1):
                break
            ins = infer_isize(v[k][0], v[i][0], (v[k][1] >> 1) & 1, (v[i][1] >> 1) & 1, v[k][2], v[i][2])
            if ins is not None and pes["low"] <= ins <= pes["high"]:
                zscore = (ins - pes["avg"]) / pes["std"]
                sc = max(0, int((v[i][1] >> 32) + (v[k][1] >> 32) + .721 * math.log(2. * math.erfc(abs(zscore) * math.sqrt(0.5))) * opt["a"] + .499))
                y = (k << 32) | i
                proper.append((((sc << 32) | (hash_64(y ^ (rid << 8)) & 0xffffffff)), y))
    if not proper:
        return 0, 0, 0, -1, -1
    proper.sort()
    i, k = proper[-1][1] >> 32, proper[-1][1] & 0xffffffff
    z = [-1, -1]
    z[v[i][1] & 1] = (v[i][1] & 0xffffffff) >> 2
    z[v[k][1] & 1] = (v[k][1] & 0xffffffff) >> 2
    score = proper[-1][0] >> 32
    sub = proper[-2][0] >> 32 if len(proper) > 1 else 0
    tmp = max(opt["a"] + opt["b"], opt["o_del"] + opt["e_del"], opt["o_ins"] + opt["e_ins"])
    n_sub = sum(1 for t in proper[:-1] if sub - (t[0] >> 32) <= tmp)
    return score, sub, n_sub, z[0], z[1]


# ---------------------------------------------------------------------------------------------------------------------
def sort_dedup(opt, l_pac, regs, klib_order, can_merge=True):   # mem_sort_deduplicate + mem_test_reg_concatenation up to its alignment, mem_alnreg.c:63-202
    """regs: dicts with rb re qb qe rid score.  klib_order(keys) -> the permutation ks_introsort leaves for records compared by
    these integer keys alone (the real klib template, through oracle/_ref: the keys here are NOT unique, so the algorithm's own
    order for equal keys is part of the result).  Returns the indices of the regions kept, in their final order, or None when two
    regions would have to be aligned across their gap (mem_alnreg.c:92-97: the score decides, and a merge rewrites a region)."""
    n = len(regs)
    if n <= 1:
        return list(range(n))
    order = klib_order([r["re"] for r in regs])            # alnreg_slt2: by END
    a = [dict(regs[i], idx=i) for i in order]
    mlr = opt["mask_level_redun"]
    for i in range(1, n):
        p = a[i]
        j = i - 1
        while j >= 0 and p["rid"] == a[j]["rid"] and p["rb"] < a[j]["re"] + opt["max_chain_gap"]:
            q = a[j]
            j -= 1
            if q["qe"] == q["qb"]:
                continue
            orr = q["re"] - p["rb"]
            oq = q["qe"] - p["qb"] if q["qb"] < p["qb"] else p["qe"] - q["qb"]
            mr = min(q["re"] - q["rb"], p["re"] - p["rb"])
            mq = min(q["qe"] - q["qb"], p["qe"] - p["qb"])
            # int64 > float * int64: both sides in single precision
            if f32(float(orr)) > f32(f32(mlr) * f32(float(mr))) and f32(float(oq)) > f32(f32(mlr) * f32(float(mq))):
                if p["score"] < q["score"]:
                    p["qe"] = p["qb"]
                    break
                q["qe"] = q["qb"]
            elif can_merge and q["rb"] < p["rb"]:   # bns == 0 (the call after a mate rescue): mem_test_reg_concatenation returns 0 at once
                if q["rb"] < l_pac <= p["rb"]:
                    continue
                if q["qb"] >= p["qb"] or q["qe"] >= p["qe"] or q["re"] >= p["re"]:
                    continue
                w = abs((q["re"] - p["rb"]) - (q["qe"] - p["qb"]))
                r = abs((q["re"] - p["rb"]) / (p["re"] - q["rb"]) - (q["qe"] - p["qb"]) / (p["qe"] - q["qb"]))
                if q["re"] < p["rb"] or q["qe"] < p["qb"]:
                    if w > opt["w"] << 1 or r >= f32(0.05):
                        continue
                elif w > opt["w"] << 2 or r >= f32(f32(0.05) * 2):
                    continue
                return None
    a = [r for r in a if r["qe"] > r["qb"]]
    # alnreg_slt: score descending, then rb, then qb -- one ascending integer key with the same order
    order = klib_order([((1 << 19) - r["score"]) << 44 | r["rb"] << 10 | r["qb"] for r in a])
    a = [a[i] for i in order]
    dead = [False] * len(a)
    for i in range(1, len(a)):
        if a[i]["score"] == a[i - 1]["score"] and a[i]["rb"] == a[i - 1]["rb"] and a[i]["qb"] == a[i - 1]["qb"]:
            dead[i] = True
    return [a[i]["idx"] for i in range(len(a)) if i == 0 or not dead[i]]


# ---------------------------------------------------------------------------------------------------------------------
KSW_XBYTE, KSW_XSUBO, KSW_XSTART = 0x10000, 0x40000, 0x80000   # lib/aln/ksw.h:6-9


def _pos2rid(anns, pos_f):   # bns_pos2rid (bntseq.c:356-369): the contig holding forward position pos_f
    for i, (off, ln) in enumerate(anns):
        if off <= pos_f < off + ln:
            return i
    return -1


def fetch_seq(l_pac, anns, get_base, beg, mid, end):   # bns_fetch_seq over bns_get_seq (bntseq.c:402-452) -> (seq, beg, end, rid)
    if end < beg:
        beg, end = end, beg
    is_rev = mid >= l_pac
    rid = _pos2rid(anns, (l_pac << 1) - 1 - mid if is_rev else mid)
    far_beg, far_end = anns[rid][0], anns[rid][0] + anns[rid][1]
    if is_rev:
        far_beg, far_end = (l_pac << 1) - far_end, (l_pac << 1) - far_beg
    beg, end = max(beg, far_beg), min(end, far_end)
    if beg >= l_pac:       # reverse strand: the complement, read backwards on the forward strand
        seq = [3 - get_base((l_pac << 1) - 1 - k) for k in range(beg, end)]
    else:
        seq = [get_base(k) for k in range(beg, end)]
    return seq, beg, end, rid


def _matesw_core(opt, l_pac, anns, get_base, pes, reg, ms, mregs, ksw_align2, klib_order):   # mem_alnreg.c:395-491
    l_ms = len(ms)
    for m in mregs:
        ins = alnreg_isize(l_pac, reg, m)
        if ins is not None and pes["low"] <= ins <= pes["high"]:
            return
    rev = [0] * l_ms
    for i in range(l_ms):
        rev[l_ms - 1 - i] = 3 - ms[i] if ms[i] < 4 else 4
    rb = max(0, reg["rb"] + pes["low"] - l_ms)
    re = min(l_pac << 1, reg["rb"] + pes["high"])
    rid, ref = -1, None
    if rb < re:
        ref, rb, re, rid = fetch_seq(l_pac, anns, get_base, rb, (rb + re) >> 1, re)
    if reg["rid"] != rid or re - rb < opt["min_seed_len"]:
        return
    parent = reg["bss"] ^ (1 if reg["rb"] < l_pac else 0)
    xtra = KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if l_ms * opt["a"] < 250 else 0) | (opt["min_seed_len"] * opt["a"])
    aln = ksw_align2(rev, ref, opt["gamat"] if parent else opt["ctmat"], xtra)   # the mate is on the other converted strand
    if aln["score"] >= opt["min_seed_len"] and aln["qb"] >= 0:
        b = {"rid": reg["rid"], "is_alt": reg["is_alt"], "qb": l_ms - (aln["qe"] + 1), "qe": l_ms - aln["qb"],
             "rb": (l_pac << 1) - (rb + aln["te"] + 1), "re": (l_pac << 1) - (rb + aln["tb"]), "score": aln["score"], "csub": aln["score2"],
             "secondary": -1, "bss": reg["bss"], "parent": 1 - parent}
        b["seedcov"] = min(b["re"] - b["rb"], b["qe"] - b["qb"]) >> 1
        at = len(mregs)
        for i, m in enumerate(mregs):
            if m["score"] < b["score"]:
                at = i
                break
        mregs.insert(at, b)
        keep = sort_dedup(opt, l_pac, mregs, klib_order, can_merge=False)
        mregs[:] = [mregs[i] for i in keep]


def matesw(opt, l_pac, anns, get_base, pes, seqs, regs_pair, ksw_align2, klib_order):   # mem_alnreg.c:494-513
    good = [[], []]
    for i in range(2):
        for r in regs_pair[i]:
            if r["score"] >= regs_pair[i][0]["score"] - opt["pen_unpaired"]:
                good[i].append(dict(r))
    for i in range(2):
        for j in range(min(len(good[i]), opt["max_matesw"])):
            _matesw_core(opt, l_pac, anns, get_base, pes, good[i][j], seqs[1 - i], regs_pair[1 - i], ksw_align2, klib_order)


# ---------------------------------------------------------------------------------------------------------------------
MEM_F_NOPAIRING, MEM_F_ALL, MEM_F_NO_MULTI, MEM_F_KEEP_SUPP_MAPQ = 0x4, 0x8, 0x10, 0x1000   # lib/aln/bwamem.h:42-52


def _select_format(opt, regs, mapq_se, trace_setsam):   # mem_alnreg_select_format, mem_alnreg_format.c:445-488
    out = []
    for k, p in enumerate(regs):
        if p["rb"] < 0 or p["re"] < 0:
            continue
        if p["score"] < opt["T"]:
            continue
        if p["secondary"] >= 0 and (p["is_alt"] or not (opt["flag"] & MEM_F_ALL)):
            continue
        if p["secondary"] >= 0 and p["secondary"] < INT_MAX and f32(float(p["score"])) < f32(f32(float(regs[p["secondary"]]["score"])) * f32(opt["drop_ratio"])):
            continue
        if out and p["secondary"] < 0:
            p["flag"] |= 0x10000 if opt["flag"] & MEM_F_NO_MULTI else 0x800
        if p["secondary"] >= 0:
            p["flag"] |= 0x100
        p["mapq"] = mapq_se(p) if p["secondary"] < 0 else 0
        if not (opt["flag"] & MEM_F_KEEP_SUPP_MAPQ) and out and not p["is_alt"]:
            p["mapq"] = min(p["mapq"], regs[0]["mapq"])
        out.append(k)
    return out


def _nopairing(opt, regs_pair, mapq_se, trace):   # mem_reg2sam_pe_nopairing, mem_alnreg_format.c:519-559
    sel = [_select_format(opt, regs_pair[i], mapq_se, None) for i in range(2)]
    best = [sel[i][0] if sel[i] else -1 for i in range(2)]           # -1: the unmapped stand-in (flag 0x40<<i | 0x1 | 0x4)
    for i in range(2):
        if sel[i]:
            for j, k in enumerate(sel[i]):
                p = regs_pair[i][k]
                trace.append((i, k, best[1 - i], p["flag"], p["mapq"], int(j == 0)))
        else:
            trace.append((i, -1, best[1 - i], 0x40 << i | 0x1 | 0x4, 0, 1))


def raw_mapq(diff, a):
    return int(6.02 * diff / a + .499)


def reg2sam_pe(opt, l_pac, ann_offset, pes, rid, regs_pair, n_pri, mapq_se):   # mem_reg2sam_pe, mem_alnreg_format.c:562-696 -> the records written
    trace = []
    for i in range(2):
        for p in regs_pair[i]:
            p["flag"] |= (0x40 << i) | 1
    if opt["flag"] & MEM_F_NOPAIRING or n_pri[0] == 0 or n_pri[1] == 0:
        _nopairing(opt, regs_pair, mapq_se, trace)
        return trace
    for i in range(2):
        for j in range(1, n_pri[i]):
            if regs_pair[i][j]["secondary"] < 0 and regs_pair[i][j]["score"] >= opt["T"]:
                _nopairing(opt, regs_pair, mapq_se, trace)
                return trace
    pscore, sub_pscore, n_sub, z0, z1 = pair(opt, l_pac, ann_offset, pes, regs_pair, n_pri, rid)
    z = [z0, z1]
    if pscore <= 0:
        _nopairing(opt, regs_pair, mapq_se, trace)
        return trace
    score_unpaired = regs_pair[0][0]["score"] + regs_pair[1][0]["score"] - opt["pen_unpaired"]
    if pscore > score_unpaired:
        sub_pscore = max(sub_pscore, score_unpaired)
        q_pe = raw_mapq(pscore - sub_pscore, opt["a"])
        if n_sub > 0:
            q_pe -= int(4.343 * math.log(n_sub + 1) + .499)
        q_pe = max(0, min(60, q_pe))
        q_pe = int(q_pe * (1. - .5 * f32(regs_pair[0][0]["frac_rep"] + regs_pair[1][0]["frac_rep"])) + .499)   # float + float, then double
        c = [regs_pair[0][z[0]], regs_pair[1][z[1]]]
        q_se = [0, 0]
        for i in range(2):
            if c[i]["secondary"] >= 0:
                c[i]["sub"] = regs_pair[i][c[i]["secondary"]]["score"]
                c[i]["secondary"] = -2
            q_se[i] = mapq_se(c[i])
        for i in range(2):
            q_se[i] = max(q_se[i], min(q_pe, q_se[i] + 40))
            c[i]["mapq"] = min(q_se[i], raw_mapq(c[i]["score"] - c[i]["csub"], opt["a"]))
    else:
        z = [0, 0]
        for i in range(2):
            regs_pair[i][0]["mapq"] = mapq_se(regs_pair[i][0])
    for i in range(2):
        regs = regs_pair[i]
        k = regs[z[i]]["secondary_all"]
        if 0 <= k < n_pri[i]:
            for j, r in enumerate(regs):
                if r["secondary_all"] == k or j == k:
                    r["secondary_all"] = z[i]
            regs[z[i]]["secondary_all"] = -1
    for i in range(2):
        regs = regs_pair[i]
        p = regs[z[i]]
        trace.append((i, z[i], z[1 - i], p["flag"], p["mapq"], 1))
        if n_pri[i] < len(regs):
            q = regs[n_pri[i]]
            if q["score"] >= opt["T"] and q["secondary"] < 0:
                q["flag"] |= 0x800
                trace.append((i, n_pri[i], -2, q["flag"], q["mapq"], 0))
    return trace


# ---------------------------------------------------------------------------------------------------------------------
MEM_F_REF_HDR, MEM_F_SOFTCLIP = 0x100, 0x200


def get_rlen(cigar):   # bwamem.h:200-208
    return sum(c >> 4 for c in cigar if (c & 0xf) in (0, 2))


def _cigar_text(opt, r, is_primary, alphabet="MIDSH"):   # the clip letter depends on the record (mem_alnreg_format.c:281-287)
    out = []
    for c in r["cigar"]:
        op = c & 0xf
        if not (opt["flag"] & MEM_F_SOFTCLIP) and not r["is_alt"] and op in (3, 4):
            op = 3 if is_primary else 4
        out.append("%d%s" % (c >> 4, alphabet[op]))
    return "".join(out)


def _pri_idx(opt, regs, i):   # get_pri_idx, mem_alnreg.h:127-131: int >= int * double
    k = regs[i]["secondary_all"]
    if k >= 0 and regs[i]["score"] >= regs[k]["score"] * float(f32(opt["XA_drop_ratio"])):
        return k
    return -1


def _tag_sa(names, p_idx, regs0):   # mem_alnreg_format.c:194-228
    if regs0 is None or regs0[p_idx]["flag"] & 0x100:
        return ""
    out = ""
    for i, q in enumerate(regs0):
        if i == p_idx or not q["cigar"] or q["flag"] & 0x100:
            continue
        out += "%s,%d,%s,%s,%d,%d;" % (names[q["rid"]], q["pos"] + 1, "+-"[q["is_rev"]], "".join("%d%s" % (c >> 4, "MIDSH"[c & 0xf]) for c in q["cigar"]), q["mapq"], q["NM"])
    return "\tSA:Z:" + out if out else ""


def _tag_xaxb(opt, names, p_idx, regs0):   # mem_alnreg_format.c:126-191
    if regs0 is None or opt["flag"] & MEM_F_ALL:
        return ""
    mine = [i for i in range(len(regs0)) if _pri_idx(opt, regs0, i) == p_idx]
    cnt_alt = sum(1 for i in mine if regs0[i]["is_alt"])
    cnt_pri = len(mine) - cnt_alt
    out = ""
    if cnt_pri <= opt["max_XA_hits"] and cnt_alt <= opt["max_XA_hits_alt"]:
        parts = []
        for i in mine:
            q = regs0[i]
            if not q["cigar"]:
                continue
            parts.append("%s,%s%d,%s,%d" % (names[q["rid"]], "+-"[q["is_rev"]], q["pos"] + 1, "".join("%d%s" % (c >> 4, "MIDSHN"[c & 0xf]) for c in q["cigar"]), q["NM"]))
        if parts:
            out += "\tXA:Z:" + ";".join(parts)
    if cnt_pri > 0 or cnt_alt > 0:
        out += "\tXB:Z:%d,%d" % (cnt_pri, cnt_alt)
    return out


def format_sam(opt, l_pac, names, annos, s, p0, m0, regs0, p_idx, is_primary, pes, rg_id):   # mem_alnreg_format.c:237-436 -> one line
    """s: the read (name comment seq0 qual l_seq barcode umi); p0 / m0 / regs0[*]: regions with their SAM side (pos is_rev cigar md NM ZC ZR
    bss_u mapq flag ...); m0 may be None; regs0 None or the read's list with p0 = regs0[p_idx]"""
    p = dict(p0)
    m = dict(m0) if m0 is not None else {"rid": 0, "pos": 0, "is_rev": 0, "cigar": [], "mapq": 0, "flag": 0, "is_alt": 0, "bss_u": 0}
    has_m = m0 is not None
    if has_m:
        p["flag"] |= 0x1
        if m["rid"] < 0:
            p["flag"] |= 0x8
        if m0["bss_u"] == 0:
            p["bss_u"] = 0
    if p["rid"] >= 0 and has_m and m["rid"] >= 0 and pes is not None:
        ins = alnreg_isize(l_pac, p, m)
        if ins is not None and pes["low"] <= ins <= pes["high"]:
            p["flag"] |= 2
    if p["rid"] < 0 and has_m and m["rid"] >= 0:
        p.update(rid=m["rid"], pos=m["pos"], is_rev=m["is_rev"], cigar=[])
    if has_m and m["rid"] < 0 and p["rid"] >= 0:
        m.update(rid=p["rid"], pos=p["pos"], is_rev=p["is_rev"], cigar=[])
    if has_m and m["is_rev"]:
        p["flag"] |= 0x20
    f = [s["name"] + ("_" + s["comment"] if s.get("comment") else ""), str((p["flag"] & 0xffff) | (0x100 if p["flag"] & 0x10000 else 0))]
    if p["rid"] >= 0:
        f += [names[p["rid"]], str(p["pos"] + 1), str(p["mapq"]), _cigar_text(opt, p, is_primary) if p["cigar"] else "*"]
    else:
        f += ["*", "0", "0", "*"]
    if has_m and m["rid"] >= 0:
        f += ["=" if p["rid"] == m["rid"] else names[m["rid"]], str(m["pos"] + 1)]
        tlen = "0"
        if p["rid"] == m["rid"]:
            q0 = q1 = -1
            if p["is_rev"]:
                q1 = p["pos"] + get_rlen(p["cigar"]) - 1
            else:
                q0 = p["pos"]
            if m["is_rev"]:
                q1 = m["pos"] + get_rlen(m["cigar"]) - 1
            else:
                q0 = m["pos"]
            if p["cigar"] and m["cigar"] and q0 >= 0 and q1 >= 0:
                tlen = str(q1 - q0 + 1)
        f.append(tlen)
    else:
        f += ["*", "0", "0"]
    if p["flag"] & 0x100:
        f += ["*", "*"]
    else:
        qb, qe = 0, len(s["seq0"])
        if p["cigar"] and not is_primary and not (opt["flag"] & MEM_F_SOFTCLIP) and not p["is_alt"]:
            c0, c1 = p["cigar"][0], p["cigar"][-1]
            if p["is_rev"]:
                if (c0 & 0xf) in (3, 4):
                    qe -= c0 >> 4
                if (c1 & 0xf) in (3, 4):
                    qb += c1 >> 4
            else:
                if (c0 & 0xf) in (3, 4):
                    qb += c0 >> 4
                if (c1 & 0xf) in (3, 4):
                    qe -= c1 >> 4
        if p["is_rev"]:
            f.append("".join("TGCAN"[s["seq0"][i]] for i in range(qe - 1, qb - 1, -1)))
            f.append(s["qual"][qb:qe][::-1] if s.get("qual") else "*")
        else:
            f.append("".join("ACGTN"[s["seq0"][i]] for i in range(qb, qe)))
            f.append(s["qual"][qb:qe] if s.get("qual") else "*")
    line = "\t".join(f)
    if p["cigar"]:
        line += "\tNM:i:%d\tMD:Z:%s\tZC:i:%d\tZR:i:%d" % (p["NM"], p["md"], p["ZC"], p["ZR"])
    if p["score"] >= 0:
        line += "\tAS:i:%d" % p["score"]
    if p["sub"] >= 0:
        line += "\tXS:i:%d" % max(p["sub"], p["csub"])
    if rg_id:
        line += "\tRG:Z:" + rg_id
    line += _tag_sa(names, p_idx, regs0) if regs0 is not None else ""
    if is_primary and p["alt_sc"] > 0:
        line += "\tPA:f:%.3f" % (p["score"] / p["alt_sc"])
    line += "\tXL:i:%d" % s["l_seq"]
    line += _tag_xaxb(opt, names, p_idx, regs0) if regs0 is not None else ""
    if opt["flag"] & MEM_F_REF_HDR and p["rid"] >= 0 and annos[p["rid"]]:
        line += "\tXR:Z:" + annos[p["rid"]].replace("\t", " ")
    if s.get("barcode"):
        line += "\tCB:Z:" + s["barcode"]
    if s.get("umi"):
        line += "\tRX:Z:" + s["umi"]
    line += "\tMC:Z:" + (_cigar_text(opt, m, is_primary) if m["cigar"] else "*")
    line += "\tMQ:i:%d" % m["mapq"]
    line += "\tYD:A:" + ("u" if p["bss_u"] else "fr"[p["bss"]])
    return line + "\n"


# ---------------------------------------------------------------------------------------------------------------------
def strand_searches(parent, is_pe, second_of_pair):   # bis_worker1, bwamem.c:311-376: the mem_align1_core calls of one read, in order
    """-> the `parent` argument of each call: 1 = the read as C>T against the parent index, 0 = as G>A against the daughter index"""
    if not is_pe:
        out = []
        if not (parent & 1) or parent >> 1:
            out.append(0)
        if not (parent & 1) or not (parent >> 1):
            out.append(1)
        return out
    if not second_of_pair:
        return [1] + ([0] if not parent else [])
    return [0] + ([1] if not parent else [])


# ---------------------------------------------------------------------------------------------------------------------
def setsam_post(l_pac, ann_offset, s, reg, cigar):   # mem_alnreg_setSAM after the alignment, mem_alnreg_format.c:79-120
    """s: l_seq clip5 clip3 of the read; reg: rb re qb qe rid -> (pos, is_rev, final cigar); the MD string moves with the cigar unchanged"""
    p = reg["rb"] if reg["rb"] < l_pac else reg["re"] - 1
    is_rev = int(p >= l_pac)
    rpos = (l_pac << 1) - 1 - p if is_rev else p
    cigar = list(cigar)
    if cigar:
        if cigar[0] & 0xf == 2:
            rpos += cigar[0] >> 4
            cigar = cigar[1:]
        elif cigar[-1] & 0xf == 2:
            cigar = cigar[:-1]
    if reg["qb"] != 0 or reg["qe"] != s["l_seq"] or s["clip5"] or s["clip3"]:
        clip5 = s["l_seq"] - reg["qe"] + s["clip3"] if is_rev else reg["qb"] + s["clip5"]
        clip3 = reg["qb"] + s["clip5"] if is_rev else s["l_seq"] - reg["qe"] + s["clip3"]
        if clip5:
            cigar = [clip5 << 4 | 3] + cigar
        if clip3:
            cigar = cigar + [clip3 << 4 | 3]
    return rpos - ann_offset[reg["rid"]], is_rev, cigar


# ---------------------------------------------------------------------------------------------------------------------
MEM_SHORT_EXT, MEM_SHORT_LEN = 50, 200   # memchain.c:494-498


def _seed_sw(opt, l_pac, anns, get_base, query, seed, parent, ksw_align2):   # mem_seed_sw, memchain.c:501-537
    rbeg, qbeg, ln = seed
    if ln >= MEM_SHORT_LEN:
        return -1
    qb, qe, rb, re = qbeg, qbeg + ln, rbeg, rbeg + ln
    mid = (rb + re) >> 1
    qb = max(qb - MEM_SHORT_EXT, 0)
    qe = min(qe + MEM_SHORT_EXT, len(query))
    rb = max(rb - MEM_SHORT_EXT, 0)
    re = min(re + MEM_SHORT_EXT, l_pac << 1)
    if rb < l_pac < re:
        if mid < l_pac:
            re = l_pac
        else:
            rb = l_pac
    if qe - qb >= MEM_SHORT_LEN or re - rb >= MEM_SHORT_LEN:
        return -1
    rseq, rb, re, _ = fetch_seq(l_pac, anns, get_base, rb, mid, re)
    return ksw_align2(query[qb:qe], rseq, opt["ctmat"] if parent else opt["gamat"], KSW_XSTART)["score"]


def flt_chained_seeds(opt, l_pac, anns, get_base, query, seeds, parent, ksw_align2):   # memchain.c:541-568 -> [(seed index, score)] kept
    l_query = len(query)
    min_l = f32(f32(1.1) * opt["min_chain_weight"]) if opt["min_chain_weight"] else float(f32(5.5)) * math.log(l_query)   # float * int; float -> double * double
    if min_l > f32(f32(0.05) * l_query):
        return [(i, None) for i in range(len(seeds))]     # short read: the chain is left as it is
    min_hsp = int(opt["a"] * min_l + .499)
    out = []
    for i, sd in enumerate(seeds):
        sc = _seed_sw(opt, l_pac, anns, get_base, query, sd, parent, ksw_align2)
        if sc < 0 or sc >= min_hsp:
            out.append((i, sd[2] * opt["a"] if sc < 0 else sc))
    return out
