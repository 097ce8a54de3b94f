"""oracle/e2e.py -- TEST INFRASTRUCTURE, not product code.

An end-to-end restatement of `biscuit align` (FASTQ + index files in, SAM text out) that shares NO host code with the product:
nothing under biscuit_amd/ is imported, loaded or executed here.  The kernels are the reference's own functions, compiled from
the sources where they lie (oracle/_ref/libbiscuit_ref.so: bwt_smem1a, bwt_seed_strategy1, bwt_sa, ksw_extend2, ksw_align2,
ksw_global2, the klib introsort and B-tree templates, kseq_read, read_clipping, mem_approx_mapq_se, bwa_print_sam_hdr); every
function of the reference that cannot be compiled here (memchain.c, mem_alnreg.c, mem_pair.c, mem_alnreg_format.c, bntseq.c and
align.c include wzmisc.h, which is not under /root/reference) is restated below in plain Python from its source lines, cited at
each function.  The restatements of mem_mark_primary_se, mem_pestat and mem_pair are the ones of oracle/backhalf.py.

    index files            bwa_idx_load / bns_restore_core / bwt_restore_*      lib/aln/bwa.c:490-560, bntseq.c:99-214
    read_chunk             bis_bseq_read, bis_kseq2bseq1, trim_readno           lib/aln/bwa.c:58-63,764-850
    collect_intv           mem_collect_intv                                     lib/aln/memchain.c:50-106
    chain                  mem_chain, merge_seed_to_chain                       lib/aln/memchain.c:227-393
    chain_flt              mem_chain_flt, mem_chain_weight                      lib/aln/memchain.c:158-180,406-488
    flt_chained_seeds      mem_flt_chained_seeds, mem_seed_sw                   lib/aln/memchain.c:501-568
    chain2region           mem_chain2region, mem_chain2region1, extensions      lib/aln/memchain.c:576-904
    sort_deduplicate       mem_sort_deduplicate, mem_test_reg_concatenation     lib/aln/mem_alnreg.c:63-202
    merge_regions          mem_merge_regions                                    lib/aln/mem_alnreg.c:214-238
    matesw                 mem_alnreg_matesw, mem_alnreg_matesw_core            lib/aln/mem_alnreg.c:385-513
    gen_cigar2             bis_bwa_gen_cigar2 (DP by the real ksw_global2)      lib/aln/bwa.c:290-428
    set_sam                mem_alnreg_setSAM                                    lib/aln/mem_alnreg_format.c:40-123
    format_sam, tags       mem_alnreg_formatSAM, _tagSA, _tagXAXB               lib/aln/mem_alnreg_format.c:126-436
    select_format, reg2sam mem_alnreg_select_format, mem_reg2sam_se/_pe/_pe_nopairing   lib/aln/mem_alnreg_format.c:445-696
    worker1 / worker2      bis_worker1, bis_worker2, mem_align1_core            lib/aln/bwamem.c:183-407
    process_seqs           mem_process_seqs                                     lib/aln/bwamem.c:432-476
    parse_args, align      main_align, process, update_a                        lib/aln/align.c:70-182,319-598

Plain Python, one read at a time: meant for thousands of reads against genomes of a few Mbp (the whole forward genome is held
as one byte per base).  `python oracle/e2e.py [biscuit align options] <index base> <in1.fq> [in2.fq]` writes SAM to stdout.
"""
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from backhalf import (INT_MAX, M64, alnreg_isize, f32, hash_64, mark_primary_se, pair, pestat)  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PATH = os.path.join(HERE, "_ref", "libbiscuit_ref.so")

# lib/aln/bwamem.h:42-52
MEM_F_PE, MEM_F_NOPAIRING, MEM_F_ALL, MEM_F_NO_MULTI, MEM_F_NO_RESCUE = 0x2, 0x4, 0x8, 0x10, 0x20
MEM_F_SELF_OVLP, MEM_F_ALN_REG, MEM_F_REF_HDR, MEM_F_SOFTCLIP, MEM_F_SMARTPE, MEM_F_KEEP_SUPP_MAPQ = 0x40, 0x80, 0x100, 0x200, 0x400, 0x1000
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000   # lib/aln/ksw.h:6-9

# nst_nt4_table (lib/aln/bntseq.c:49-66): A/a 0, C/c 1, G/g 2, T/t 3, '-' 5, everything else 4
NT4 = np.full(256, 4, np.uint8)
for _c, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("a", 0), ("c", 1), ("g", 2), ("t", 3), ("-", 5)):
    NT4[ord(_c)] = _v

_u8p = C.POINTER(C.c_uint8)


def _p(a):
    return a.ctypes.data_as(_u8p)


class Ref:
    """the reference's own functions (oracle/_ref), nothing else"""

    def __init__(self):
        if not os.path.exists(REF_PATH):
            raise RuntimeError("oracle/_ref/libbiscuit_ref.so is missing: run `make ref` where /root/reference exists")
        R = self.R = C.CDLL(REF_PATH)
        R.ref_bwt_load.restype = C.c_void_p
        R.ref_bwt_load.argtypes = [C.c_char_p, C.c_char_p]
        R.ref_bwt_sa.restype = C.c_uint64
        R.ref_bwt_sa.argtypes = [C.c_void_p, C.c_uint64]
        R.ref_bwt_smem1a.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]
        R.ref_bwt_seed_strategy1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        R.ref_ksw_extend2.argtypes = [C.c_int, _u8p, C.c_int, _u8p, C.POINTER(C.c_int8)] + [C.c_int] * 8 + [C.POINTER(C.c_int)]
        R.ref_ksw_align2.argtypes = [C.c_int, _u8p, C.c_int, _u8p, C.POINTER(C.c_int8)] + [C.c_int] * 5 + [C.POINTER(C.c_int)]
        R.ref_ksw_global2.argtypes = [C.c_int, _u8p, C.c_int, _u8p, C.POINTER(C.c_int8)] + [C.c_int] * 6 + [C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.c_int]
        R.ref_fill_scmat.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int8)]
        R.ref_approx_mapq_se.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int] + [C.c_int] * 6 + [C.c_int64, C.c_int64, C.c_int, C.c_float]
        R.ref_infer_bw.argtypes = [C.c_int] * 6
        R.ref_introsort_kv.argtypes = [C.c_int64, C.POINTER(C.c_int64)]
        R.ref_introsort_kv_desc.argtypes = [C.c_int64, C.POINTER(C.c_int64)]
        R.ref_introsort_64.argtypes = [C.c_int64, C.POINTER(C.c_uint64)]
        R.ref_bt_new.restype = C.c_void_p
        R.ref_bt_free.argtypes = [C.c_void_p]
        R.ref_bt_put.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        R.ref_bt_interval.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        R.ref_bt_traverse.restype = C.c_int64
        R.ref_bt_traverse.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int64]
        R.ref_kseq_open.restype = C.c_void_p
        R.ref_kseq_open.argtypes = [C.c_char_p]
        R.ref_kseq_close.argtypes = [C.c_void_p]
        R.ref_kseq_next.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        R.ref_read_clipping.argtypes = [C.c_int, _u8p, C.c_char_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        R.ref_sam_hdr.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]

    # --- klib sorts: the permutation ks_introsort leaves for records compared by `key` alone (ksort.h:184-236) ---
    def order(self, keys, desc=False):
        n = len(keys)
        kv = (C.c_int64 * (2 * n))()
        for i, k in enumerate(keys):
            kv[2 * i], kv[2 * i + 1] = k, i
        (self.R.ref_introsort_kv_desc if desc else self.R.ref_introsort_kv)(n, kv)
        return [kv[2 * i + 1] for i in range(n)]

    def order_tuples(self, tuples):
        """records compared by a lexicographic `<` over a tuple: the comparison outcomes are those of the tuples' dense ranks"""
        rank = {t: r for r, t in enumerate(sorted(set(tuples)))}
        return self.order([rank[t] for t in tuples])

    def sort_u64(self, a):
        n = len(a)
        v = (C.c_uint64 * n)(*a)
        self.R.ref_introsort_64(n, v)
        return list(v)

    # --- DP kernels (ksw.c) ---
    def extend2(self, q, t, mat, opt, w, end_bonus, h0):
        out = (C.c_int * 6)()
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        self.R.ref_ksw_extend2(len(q), _p(q), len(t), _p(t), mat, opt["o_del"], opt["e_del"], opt["o_ins"], opt["e_ins"], w, end_bonus, opt["zdrop"], h0, out)
        return tuple(out)      # score qle tle gtle gscore max_off

    def align2(self, q, t, mat, opt, xtra):
        out = (C.c_int * 7)()
        q = np.array(q, np.uint8)     # copies: the reference reverses them in place (and restores them)
        t = np.array(t, np.uint8)
        self.R.ref_ksw_align2(len(q), _p(q), len(t), _p(t), mat, opt["o_del"], opt["e_del"], opt["o_ins"], opt["e_ins"], xtra, out)
        return dict(zip(("score", "te", "qe", "score2", "te2", "tb", "qb"), out))

    def global2(self, q, t, mat, opt, w, want_cigar):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        cap = len(q) + len(t) + 8
        cg = (C.c_uint32 * cap)()
        n = C.c_int(0)
        sc = self.R.ref_ksw_global2(len(q), _p(q), len(t), _p(t), mat, opt["o_del"], opt["e_del"], opt["o_ins"], opt["e_ins"], w, int(want_cigar), C.byref(n), cg, cap)
        assert n.value <= cap
        return sc, list(cg[:n.value])

    def mapq_se(self, opt, r):     # mem_approx_mapq_se, bwamem.c:134-157
        return self.R.ref_approx_mapq_se(opt["a"], opt["b"], opt["min_seed_len"], opt["mapQ_coef_len"], opt["mapQ_coef_fac"], r["score"], r["sub"], r["csub"],
                                         r["sub_n"], r["qb"], r["qe"], r["rb"], r["re"], r["seedcov"], r["frac_rep"])


_REF = None


def ref():
    global _REF
    if _REF is None:
        _REF = Ref()
    return _REF


# =====================================================================================================================
# options (mem_opt_init, bwamem.c:77-128; main_align's getopt ladder, align.c:339-462; update_a, align.c:169-182)
def opt_init():
    o = dict(flag=0, a=1, b=2, o_del=6, o_ins=6, e_del=1, e_ins=1, w=100, T=30, zdrop=100, pen_unpaired=17, pen_clip5=10, pen_clip3=10,
             max_mem_intv=20, min_seed_len=19, split_width=10, max_occ=500, max_chain_gap=10000, max_ins=5000, mask_level=f32(0.50), drop_ratio=f32(0.50),
             XA_drop_ratio=f32(0.80), split_factor=f32(1.5), chunk_size=10000000, n_threads=1, max_XA_hits=5, max_XA_hits_alt=5, max_matesw=50,
             mask_level_redun=f32(0.95), min_chain_weight=0, max_chain_extend=1 << 30, mapQ_coef_len=f32(50.0), mapQ_coef_fac=int(math.log(50)),
             bsstrand=0, parent=0, clip5=0, clip3=0, min_base_qual=0, has_bc=0, adaptor1=None, adaptor2=None)
    return o


def _atoi(s):       # C atoi / the leading integer strtol reads
    s = s.strip()
    k = 1 if s[:1] in "+-" else 0
    while k < len(s) and s[k].isdigit():
        k += 1
    try:
        return int(s[:k])
    except ValueError:
        return 0


def _two(s):        # "INT[,INT]" as align.c reads -O -E -L -g: strtol, then one punctuation mark and a digit start the second
    a = _atoi(s)
    k = 1 if s[:1] in "+-" else 0
    while k < len(s) and s[k].isdigit():
        k += 1
    b = a
    if k < len(s) and not s[k].isalnum() and not s[k].isspace() and k + 1 < len(s) and s[k + 1].isdigit():
        b = _atoi(s[k + 1:])
    return a, b


def _strtod_prefix(s):
    k = 0
    best = None
    while k <= len(s):
        try:
            best = (float(s[:k]), k)
        except ValueError:
            pass
        k += 1
    return best if best else (0.0, 0)


def parse_args(argv):
    """-> (opt, pes0, hdr_line, rg_id, copy_comment, positional).  The option -> field table of align.c:339-462."""
    opt = opt_init()
    opt["flag"] |= MEM_F_NO_MULTI         # align.c:335
    opt0 = set()
    pes0 = None
    hdr_line = None
    rg_line = None
    rg_id = ""
    copy_comment = False
    ignore_alt = False
    infer_alt = True
    with_arg = set("@1235bcdfgkmrsvwxyzABDEGHIJKLNOQRTUWX")
    pos = []
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if not a.startswith("-") or a == "-":
            pos.append(a)           # GNU getopt permutes: options may follow operands
            continue
        if a == "--":
            pos.extend(argv[i:])
            break
        k = 1
        while k < len(a):
            c = a[k]
            k += 1
            v = None
            if c in with_arg:
                if k < len(a):
                    v = a[k:]
                else:
                    v = argv[i]
                    i += 1
                k = len(a)
            if c == "k": opt["min_seed_len"] = _atoi(v); opt0.add("min_seed_len")
            elif c == "b": opt["parent"] = _atoi(v)
            elif c == "f": opt["bsstrand"] = _atoi(v)
            elif c == "i": infer_alt = False
            elif c == "w": opt["w"] = _atoi(v); opt0.add("w")
            elif c == "A": opt["a"] = _atoi(v); opt0.add("a")
            elif c == "B": opt["b"] = _atoi(v); opt0.add("b")
            elif c == "T": opt["T"] = _atoi(v); opt0.add("T")
            elif c == "U": opt["pen_unpaired"] = _atoi(v); opt0.add("pen_unpaired")
            elif c == "@": opt["n_threads"] = max(1, _atoi(v))
            elif c == "P": opt["flag"] |= MEM_F_NOPAIRING
            elif c == "a": opt["flag"] |= MEM_F_ALL
            elif c == "p": opt["flag"] |= MEM_F_PE | MEM_F_SMARTPE
            elif c == "q": opt["flag"] |= MEM_F_KEEP_SUPP_MAPQ
            elif c == "M": opt["flag"] |= MEM_F_NO_MULTI
            elif c == "S": opt["flag"] |= MEM_F_NO_RESCUE
            elif c == "e": opt["flag"] |= MEM_F_SELF_OVLP
            elif c == "F": opt["flag"] |= MEM_F_ALN_REG
            elif c == "Y": opt["flag"] |= MEM_F_SOFTCLIP
            elif c == "V": opt["flag"] |= MEM_F_REF_HDR
            elif c == "c": opt["max_occ"] = _atoi(v); opt0.add("max_occ")
            elif c == "d": opt["zdrop"] = _atoi(v); opt0.add("zdrop")
            elif c == "v": pass
            elif c == "j": ignore_alt = True
            elif c == "r": opt["split_factor"] = f32(float(v)); opt0.add("split_factor")
            elif c == "D": opt["drop_ratio"] = f32(float(v)); opt0.add("drop_ratio")
            elif c == "m": opt["max_matesw"] = _atoi(v)
            elif c == "s": opt["split_width"] = _atoi(v)
            elif c == "G": opt["max_chain_gap"] = _atoi(v)
            elif c == "N": opt["max_chain_extend"] = _atoi(v)
            elif c == "W": opt["min_chain_weight"] = _atoi(v)
            elif c == "y": opt["max_mem_intv"] = _atoi(v)
            elif c == "C": copy_comment = True
            elif c == "J": opt["adaptor1"] = NT4[np.frombuffer(v.encode(), np.uint8)].copy()
            elif c == "K": opt["adaptor2"] = NT4[np.frombuffer(v.encode(), np.uint8)].copy()
            elif c == "z": opt["min_base_qual"] = _atoi(v)
            elif c == "5": opt["clip5"] = _atoi(v)
            elif c == "3": opt["clip3"] = _atoi(v)
            elif c == "9": opt["has_bc"] = 1
            elif c == "X": opt["mask_level"] = f32(float(v))
            elif c == "g": opt["max_XA_hits"], opt["max_XA_hits_alt"] = _two(v)
            elif c == "Q":
                opt["mapQ_coef_len"] = f32(float(_atoi(v)))
                opt["mapQ_coef_fac"] = int(math.log(opt["mapQ_coef_len"])) if opt["mapQ_coef_len"] > 0 else 0
            elif c == "O": opt["o_del"], opt["o_ins"] = _two(v); opt0.update(("o_del", "o_ins"))
            elif c == "E": opt["e_del"], opt["e_ins"] = _two(v); opt0.update(("e_del", "e_ins"))
            elif c == "L": opt["pen_clip5"], opt["pen_clip3"] = _two(v); opt0.update(("pen_clip5", "pen_clip3"))
            elif c == "R":      # bwa_set_rg, bwa.c:700-730 (escapes: bwa_escape, bwa.c:686-698)
                rg_line = v.replace("\\t", "\t").replace("\\n", "\n").replace("\\r", "\r").replace("\\\\", "\\")
                assert rg_line.startswith("@RG") and "\tID:" in rg_line
                rg_id = rg_line.split("\tID:", 1)[1].split("\t")[0].split("\n")[0]
            elif c == "H":
                assert v.startswith("@"), "-H FILE is not restated"
                h = v.replace("\\t", "\t").replace("\\n", "\n").replace("\\r", "\r").replace("\\\\", "\\")
                hdr_line = h if hdr_line is None else hdr_line + "\n" + h
            elif c == "I":      # align.c:433-452
                avg, k1 = _strtod_prefix(v)
                rest = v[k1:]
                std = avg * .1
                vals = []
                while rest and not rest[0].isalnum() and not rest[0].isspace() and len(rest) > 1 and rest[1].isdigit():
                    x, k1 = _strtod_prefix(rest[1:])
                    vals.append(x)
                    rest = rest[1 + k1:]
                if vals:
                    std = vals[0]
                high = int(avg + 4. * std + .499)
                low = int(avg - 4. * std + .499)
                if len(vals) > 1:
                    high = int(vals[1] + .499)
                if len(vals) > 2:
                    low = int(vals[2] + .499)
                pes0 = {"low": low, "high": high, "failed": 0, "avg": avg, "std": std}
            elif c in "12x":
                raise NotImplementedError("-%s is not restated" % c)
            else:
                raise ValueError("unknown option -%s" % c)
    if rg_line:
        hdr_line = rg_line if hdr_line is None else hdr_line + "\n" + rg_line
    if "a" in opt0:        # update_a
        for f in ("b", "T", "o_del", "e_del", "o_ins", "e_ins", "zdrop", "pen_clip5", "pen_clip3", "pen_unpaired"):
            if f not in opt0:
                opt[f] *= opt["a"]
    fill_mats(opt)
    return opt, pes0, hdr_line, rg_id, copy_comment, pos, ignore_alt, infer_alt


def fill_mats(opt):        # bwa_fill_scmat_ct / _ga (bwa.c:146-182), the real functions
    for which, key in ((1, "ctmat"), (2, "gamat")):
        m = (C.c_int8 * 25)()
        ref().R.ref_fill_scmat(which, opt["a"], opt["b"], m)
        opt[key] = m


# =====================================================================================================================
class Index:
    """bwa_idx_load (bwa.c:490-560): bwt[0] = daughter (.dau.bwt/.dau.sa), bwt[1] = parent (.par.*); bns_restore_core (bntseq.c:99-175)"""

    def __init__(self, prefix, ignore_alt=False, infer_alt=True):
        R = ref().R
        self.bwt = [R.ref_bwt_load((prefix + ".dau.bwt").encode(), (prefix + ".dau.sa").encode()),
                    R.ref_bwt_load((prefix + ".par.bwt").encode(), (prefix + ".par.sa").encode())]
        assert self.bwt[0] and self.bwt[1]
        with open(prefix + ".bis.ann") as f:
            lines = f.read().split("\n")
        l_pac, n_seqs, _seed = lines[0].split()
        self.l_pac = int(l_pac)
        self.anns = []
        for i in range(int(n_seqs)):
            head = lines[1 + 2 * i]
            gi_name = head.split(" ", 2)          # "%u%s" then the rest of the line is the comment
            name = gi_name[1]
            rest = head[len(gi_name[0]) + 1 + len(name):]
            anno = rest[1:] if len(rest) > 1 and rest != " (null)" else ""
            off, ln, _n_ambs = lines[2 + 2 * i].split()
            self.anns.append({"name": name, "anno": anno, "offset": int(off), "len": int(ln), "is_alt": 0})
        if os.path.exists(prefix + ".alt"):      # bntseq.c:189-214
            names = {a["name"]: k for k, a in enumerate(self.anns)}
            for line in open(prefix + ".alt"):
                w = line.split("\t")[0].rstrip("\r\n")
                if not w.startswith("@") and w in names:
                    self.anns[names[w]]["is_alt"] = 1
        if infer_alt:
            self._infer_alt()
        if ignore_alt:
            for a in self.anns:
                a["is_alt"] = 0
        raw = np.fromfile(prefix + ".bis.pac", np.uint8)[:(self.l_pac >> 2) + 1]
        self.bases = ((raw[:, None] >> np.array([6, 4, 2, 0], np.uint8)) & 3).astype(np.uint8).reshape(-1)[:self.l_pac]   # _get_pac, bntseq.c:233
        self.offsets = [a["offset"] for a in self.anns]

    def _infer_alt(self):      # infer_alt_chromosomes, align.c:184-223
        if any(a["is_alt"] for a in self.anns):
            return
        found = [0] * 25
        for a in self.anns:
            nm = a["name"]
            if nm.startswith("chr"):
                if len(nm) == 4:
                    u = nm[3].upper()
                    if u == "X": found[22] = 1
                    elif u == "Y": found[23] = 1
                    elif u == "M": found[24] = 1
                    elif nm[3].isdigit() and 0 < int(nm[3]) <= 22: found[int(nm[3]) - 1] = 1
                elif len(nm) == 5 and nm[3].isdigit() and nm[4].isdigit() and 0 < int(nm[3:]) <= 22:
                    found[int(nm[3:]) - 1] = 1
        if sum(found) < 20:
            return
        for a in self.anns:
            nm = a["name"]
            if nm.startswith("chrUn") or "_random" in nm or "_hap" in nm or "_alt" in nm:
                a["is_alt"] = 1

    # --- bntseq.c:356-452, bntseq.h:92-94 ---
    def depos(self, pos):
        is_rev = pos >= self.l_pac
        return ((self.l_pac << 1) - 1 - pos if is_rev else pos), int(is_rev)

    def pos2rid(self, pos_f):
        if pos_f >= self.l_pac:
            return -1
        left, mid, right = 0, 0, len(self.anns)
        while left < right:
            mid = (left + right) >> 1
            if pos_f >= self.anns[mid]["offset"]:
                if mid == len(self.anns) - 1:
                    break
                if pos_f < self.anns[mid + 1]["offset"]:
                    break
                left = mid + 1
            else:
                right = mid
        return mid

    def intv2rid(self, rb, re):
        if rb < self.l_pac < re:
            return -2
        rid_b = self.pos2rid(self.depos(rb)[0])
        rid_e = self.pos2rid(self.depos(re - 1)[0]) if rb < re else rid_b
        return rid_b if rid_b == rid_e else -1

    def get_seq(self, beg, end):
        l_pac = self.l_pac
        if end < beg:
            beg, end = end, beg
        end = min(end, l_pac << 1)
        beg = max(beg, 0)
        if beg >= l_pac or end <= l_pac:
            if beg >= l_pac:
                beg_f, end_f = (l_pac << 1) - 1 - end, (l_pac << 1) - 1 - beg
                return (3 - self.bases[beg_f + 1:end_f + 1][::-1]).astype(np.uint8)
            return self.bases[beg:end].copy()
        return np.zeros(0, np.uint8)

    def fetch_seq(self, beg, mid, end):     # -> seq, beg, end, rid
        if end < beg:
            beg, end = end, beg
        assert beg <= mid < end
        pos_f, is_rev = self.depos(mid)
        rid = self.pos2rid(pos_f)
        far_beg = self.anns[rid]["offset"]
        far_end = far_beg + self.anns[rid]["len"]
        if is_rev:
            far_beg, far_end = (self.l_pac << 1) - far_end, (self.l_pac << 1) - far_beg
        beg, end = max(beg, far_beg), min(end, far_end)
        seq = self.get_seq(beg, end)
        assert len(seq) == end - beg
        return seq, beg, end, rid


# =====================================================================================================================
# seeding
def collect_intv(opt, idx, parent, seq):      # mem_collect_intv, memchain.c:50-106 -> [(x0, x1, x2, info)] sorted by info
    R = ref().R
    bwt, bwtc = idx.bwt[parent], idx.bwt[1 - parent]
    ln = len(seq)
    q = np.ascontiguousarray(seq, np.uint8)
    cap = ln + 8
    buf = (C.c_uint64 * (4 * cap))()
    ret = C.c_int(0)
    start_width = 2 if opt["flag"] & MEM_F_SELF_OVLP else 1
    split_len = int(f32(opt["min_seed_len"] * opt["split_factor"]) + .499)      # int * float -> float; + double
    mem = []

    def smem1(x, min_intv):     # bwt_smem1 = bwt_smem1a(..., max_intv 0, ...), bwt.c:372-375
        n = R.ref_bwt_smem1a(bwt, bwtc, ln, _p(q), x, min_intv, 0, buf, cap, C.byref(ret))
        assert n <= cap
        for i in range(n):
            info = buf[4 * i + 3]
            if (info & 0xffffffff) - (info >> 32) >= opt["min_seed_len"]:
                mem.append((buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], info))
        return ret.value

    x = 0
    while x < ln:
        if q[x] < 4:
            x = smem1(x, start_width)
        else:
            x += 1
    for k in range(len(mem)):
        p = mem[k]
        start, end = p[3] >> 32, p[3] & 0xffffffff
        if end - start < split_len or p[2] > opt["split_width"]:
            continue
        smem1((start + end) >> 1, p[2] + 1)
    if opt["max_mem_intv"] > 0:
        x = 0
        m = (C.c_uint64 * 4)()
        while x < ln:
            if q[x] < 4:
                x = R.ref_bwt_seed_strategy1(bwt, bwtc, ln, _p(q), x, opt["min_seed_len"], opt["max_mem_intv"], m)
                if m[2] > 0:
                    mem.append((m[0], m[1], m[2], m[3]))
            else:
                x += 1
    order = ref().order([p[3] for p in mem])
    return [mem[i] for i in order]


# =====================================================================================================================
# chaining.  A seed is [rbeg, qbeg, len, score]; a chain {pos, seeds, seeds_extra, rid, is_alt, frac_rep, w, kept, first}
def _merge_seed_to_chain(opt, l_pac, c, s, seed_rid):     # memchain.c:227-256
    last = c["seeds"][-1]
    first = c["seeds"][0]
    if seed_rid != c["rid"]:
        return 0
    if s[1] >= first[1] and s[1] + s[2] <= last[1] + last[2] and s[0] >= first[0] and s[0] + s[2] <= last[0] + last[2]:
        c["seeds_extra"].append(s)
        return 1
    if (last[0] < l_pac or first[0] < l_pac) and s[0] >= l_pac:
        return 0
    qdist = s[1] - last[1]
    rdist = s[0] - last[0]
    if rdist >= 0 and qdist - rdist <= opt["w"] and rdist - qdist <= opt["w"] and qdist - last[2] < opt["max_chain_gap"] and rdist - last[2] < opt["max_chain_gap"]:
        c["seeds"].append(s)
        return 1
    return 0


def _getbss(parent, l_pac, rb):      # mem_getbss, memchain.c:265
    return 1 if (rb > l_pac) == bool(parent) else 0


def chain(opt, idx, s, parent):      # mem_chain, memchain.c:268-393
    if s["l_seq"] < opt["min_seed_len"]:
        return []
    R = ref().R
    l_pac = idx.l_pac
    intvs = collect_intv(opt, idx, parent, s["bisseq"][parent])
    b = e = l_rep = 0
    for (x0, x1, x2, info) in intvs:
        if x2 <= opt["max_occ"]:
            continue
        sb, se = info >> 32, info & 0xffffffff
        if sb > e:
            l_rep += e - b
            b, e = sb, se
        else:
            e = max(e, se)
    l_rep += e - b
    tree = R.ref_bt_new()
    chains = []           # by id, the payload of the B-tree's keys
    lu = (C.c_int64 * 2)()
    for (x0, x1, x2, info) in intvs:
        slen = (info & 0xffffffff) - (info >> 32)
        k = count = 0
        while k < x2 and count < opt["max_occ"] and ((count > 5 and k < opt["max_occ"]) or count <= 5):
            rbeg = R.ref_bwt_sa(idx.bwt[parent], x0 + k)
            k += 1
            sd = [rbeg, info >> 32, slen, slen]
            rid = idx.intv2rid(rbeg, rbeg + slen)
            if rid < 0:
                continue
            if (opt["bsstrand"] & 1) and _getbss(parent, l_pac, rbeg) != opt["bsstrand"] >> 1:
                continue
            to_add = 0
            if chains:
                R.ref_bt_interval(tree, rbeg, lu)
                if lu[0] < 0 or not _merge_seed_to_chain(opt, l_pac, chains[lu[0]], sd, rid):
                    to_add = 1
            else:
                to_add = 1
            if to_add:
                count += 1
                chains.append({"pos": rbeg, "seeds": [sd], "seeds_extra": [], "rid": rid, "is_alt": int(bool(idx.anns[rid]["is_alt"]))})
                R.ref_bt_put(tree, rbeg, len(chains) - 1)
    ids = (C.c_int64 * max(1, len(chains)))()
    n = R.ref_bt_traverse(tree, ids, len(chains))
    R.ref_bt_free(tree)
    # kb_putp of a key equal to one already in the tree still inserts (kbtree.h:283-316 has no equality test), so n == len(chains)
    assert n == len(chains), (n, len(chains))
    out = [chains[ids[i]] for i in range(n)]
    frac_rep = f32(f32(float(l_rep)) / s["l_seq"])
    for c in out:
        c["frac_rep"] = frac_rep
    return out


def chain_weight(c):      # mem_chain_weight, memchain.c:158-180
    w = end = 0
    for s in c["seeds"]:
        if s[1] >= end:
            w += s[2]
        elif s[1] + s[2] > end:
            w += s[1] + s[2] - end
        end = max(end, s[1] + s[2])
    tmp, w, end = w, 0, 0
    for s in c["seeds"]:
        if s[0] >= end:
            w += s[2]
        elif s[0] + s[2] > end:
            w += s[0] + s[2] - end
        end = max(end, s[0] + s[2])
    w = min(w, tmp)
    return w if w < 1 << 30 else (1 << 30) - 1


def chain_flt(opt, chns):      # mem_chain_flt, memchain.c:406-488
    if not chns:
        return chns
    a = []
    for c in chns:
        c["first"], c["kept"] = -1, 0
        c["w"] = chain_weight(c)
        if c["w"] >= opt["min_chain_weight"]:
            a.append(c)
    if not a:
        # the reference would read chns->a[0] of an empty vector here (memchain.c:428); nothing to keep
        return a
    a = [a[i] for i in ref().order([c["w"] for c in a], desc=True)]
    beg = lambda c: c["seeds"][0][1]                        # noqa: E731
    end = lambda c: c["seeds"][-1][1] + c["seeds"][-1][2]   # noqa: E731
    a[0]["kept"] = 3
    to_keep = [0]
    for i in range(1, len(a)):
        large_overlap = 0
        ci = a[i]
        broke = False
        for k in to_keep:
            ck = a[k]
            b_max = max(beg(ck), beg(ci))
            e_min = min(end(ck), end(ci))
            if e_min > b_max and (not ck["is_alt"] or ci["is_alt"]):
                li, lj = end(ci) - beg(ci), end(ck) - beg(ck)
                min_l = min(li, lj)
                if f32(float(e_min - b_max)) >= f32(f32(float(min_l)) * opt["mask_level"]) and min_l < opt["max_chain_gap"]:
                    large_overlap = 1
                    if ck["first"] < 0:
                        ck["first"] = i
                    if f32(float(ci["w"])) < f32(f32(float(ck["w"])) * opt["drop_ratio"]) and ck["w"] - ci["w"] >= opt["min_seed_len"] << 1:
                        broke = True
                        break
        if not broke:
            to_keep.append(i)
            ci["kept"] = 2 if large_overlap else 3
    for k in to_keep:
        if a[k]["first"] >= 0:
            a[a[k]["first"]]["kept"] = 1
    i = k = 0
    while i < len(a):
        if not (a[i]["kept"] == 0 or a[i]["kept"] == 3):
            k += 1
            if k >= opt["max_chain_extend"]:
                break
        i += 1
    while i < len(a):
        if a[i]["kept"] < 3:
            a[i]["kept"] = 0
        i += 1
    return [c for c in a if c["kept"] != 0]


MEM_SHORT_EXT, MEM_SHORT_LEN = 50, 200      # memchain.c:494-498


def _seed_sw(opt, idx, l_query, query, s, parent):      # mem_seed_sw, memchain.c:501-535
    l_pac = idx.l_pac
    if s[2] >= MEM_SHORT_LEN:
        return -1
    qb, qe, rb, re = s[1], s[1] + s[2], s[0], s[0] + s[2]
    mid = (rb + re) >> 1
    qb = max(qb - MEM_SHORT_EXT, 0)
    qe = min(qe + MEM_SHORT_EXT, l_query)
    rb = max(rb - MEM_SHORT_EXT, 0)
    re = min(re + MEM_SHORT_EXT, l_pac << 1)
    if rb < l_pac < re:
        if mid < l_pac:
            re = l_pac
        else:
            rb = l_pac
    if qe - qb >= MEM_SHORT_LEN or re - rb >= MEM_SHORT_LEN:
        return -1
    rseq, rb, re, _ = idx.fetch_seq(rb, mid, re)
    return ref().align2(query[qb:qe], rseq, opt["ctmat"] if parent else opt["gamat"], opt, KSW_XSTART)["score"]


def flt_chained_seeds(opt, idx, s, chns, parent):      # mem_flt_chained_seeds, memchain.c:539-568
    l_query = s["l_seq"]
    if not chns:      # nothing below has an effect without chains (and a read clipped to nothing would take log(0))
        return
    if opt["min_chain_weight"]:
        min_l = float(f32(f32(1.1) * opt["min_chain_weight"]))
    else:
        min_l = float(f32(5.5)) * math.log(l_query)
    if min_l > float(f32(f32(0.05) * l_query)):
        return
    min_hsp = int(opt["a"] * min_l + .499)
    for c in chns:
        kept = []
        for sd in c["seeds"]:
            sd[3] = _seed_sw(opt, idx, l_query, s["seq"], sd, parent)
            if sd[3] < 0 or sd[3] >= min_hsp:
                if sd[3] < 0:
                    sd[3] = sd[2] * opt["a"]
                kept.append(sd)
        c["seeds"] = kept


# =====================================================================================================================
# chains -> regions
def new_reg():      # memset(reg, 0, sizeof(mem_alnreg_t)), mem_alnreg.h:34-68
    return dict(rb=0, re=0, qb=0, qe=0, rid=0, score=0, truesc=0, sub=0, alt_sc=0, csub=0, sub_n=0, w=0, seedcov=0, secondary=0, secondary_all=0,
                seedlen0=0, n_comp=0, is_alt=0, hash=0, flag=0, mapq=0, frac_rep=0.0, bss=0, parent=0, pos=0, cigar=[], md="", NM=0, ZC=0, ZR=0,
                bss_u=0, is_rev=0)


def cal_max_gap(opt, qlen):      # memchain.c:576-582
    l_del = int((qlen * opt["a"] - opt["o_del"]) / opt["e_del"] + 1.)
    l_ins = int((qlen * opt["a"] - opt["o_ins"]) / opt["e_ins"] + 1.)
    ln = max(l_del, l_ins, 1)
    return min(ln, opt["w"] << 1)


def _reference_span(opt, l_query, l_pac, c):      # mem_chain_reference_span, memchain.c:585-605
    r0, r1 = l_pac << 1, 0
    for s in c["seeds"]:
        b = s[0] - (s[1] + cal_max_gap(opt, s[1]))
        e = s[0] + s[2] + ((l_query - s[1] - s[2]) + cal_max_gap(opt, l_query - s[1] - s[2]))
        r0, r1 = min(r0, b), max(r1, e)
    r0, r1 = max(r0, 0), min(r1, l_pac << 1)
    if r0 < l_pac < r1:
        if c["seeds"][0][0] < l_pac:
            r1 = l_pac
        else:
            r0 = l_pac
    return r0, r1


def _asymmetric_flt_seed(rseq, query, s, rbeg):      # memchain.c:138-149
    r = rseq[s[0] - rbeg:s[0] - rbeg + s[2]]
    q = query[s[1]:s[1] + s[2]]
    return bool((((r == 3) & (q == 1)) | ((r == 0) & (q == 2))).any())


def _chain2region1(opt, idx, rseq, rmax, rid, l_query, query, seeds, regs, parent, reg0, frac_rep):      # memchain.c:742-871
    X = ref()
    l_pac = idx.l_pac
    mat = opt["ctmat"] if parent else opt["gamat"]
    n = len(seeds)
    srt = X.sort_u64([(seeds[i][3] << 32 | i) & M64 for i in range(n)])
    for k in range(n - 1, -1, -1):
        s = seeds[srt[k] & 0xffffffff]
        if _asymmetric_flt_seed(rseq, query, s, rmax[0]):
            continue
        u = reg0
        while u < len(regs):
            reg = regs[u]
            if not (s[0] < reg["rb"] or s[0] + s[2] > reg["re"] or s[1] < reg["qb"] or s[1] + s[2] > reg["qe"]):
                if not (s[2] - reg["seedlen0"] > .1 * l_query):
                    qd, rd = s[1] - reg["qb"], s[0] - reg["rb"]
                    w = min(cal_max_gap(opt, min(qd, rd)), reg["w"])
                    if qd - rd < w and rd - qd < w:
                        break
                    qd, rd = reg["qe"] - (s[1] + s[2]), reg["re"] - (s[0] + s[2])
                    w = min(cal_max_gap(opt, min(qd, rd)), reg["w"])
                    if qd - rd < w and rd - qd < w:
                        break
            u += 1
        if u < len(regs):
            i = k + 1
            while i < n:
                if srt[i] != 0:
                    t = seeds[srt[i] & 0xffffffff]
                    if not (t[2] < s[2] * .95):
                        if s[1] <= t[1] and s[1] + s[2] - t[1] >= s[2] >> 2 and t[1] - s[1] != t[0] - s[0]:
                            break
                        if t[1] <= s[1] and t[1] + t[2] - s[1] >= s[2] >> 2 and s[1] - t[1] != s[0] - t[0]:
                            break
                i += 1
            if i == n:
                srt[k] = 0
                continue
        reg = new_reg()
        aw = [opt["w"], opt["w"]]
        reg["w"] = opt["w"]
        reg["score"] = reg["truesc"] = -1
        reg["rid"] = rid
        # left_extend_seed_set_align_beg, memchain.c:613-672
        if s[1] == 0:
            reg["score"] = reg["truesc"] = s[2] * opt["a"]
            reg["qb"], reg["rb"] = 0, s[0]
        else:
            qs = query[:s[1]][::-1]
            tmp = s[0] - rmax[0]
            rs = rseq[:tmp][::-1]
            for i in range(2):
                prev = reg["score"]
                aw[0] = opt["w"] << i
                reg["score"], qle, tle, gtle, gscore, max_off = X.extend2(qs, rs, mat, opt, aw[0], opt["pen_clip5"], s[2] * opt["a"])
                if reg["score"] == prev or max_off < (aw[0] >> 1) + (aw[0] >> 2):
                    break
            if gscore <= 0 or gscore <= reg["score"] - opt["pen_clip5"]:
                reg["qb"], reg["rb"], reg["truesc"] = s[1] - qle, s[0] - tle, reg["score"]
            else:
                reg["qb"], reg["rb"], reg["truesc"] = 0, s[0] - gtle, gscore
        # right_extend_seed_set_align_end, memchain.c:677-730
        if s[1] + s[2] == l_query:
            reg["qe"], reg["re"] = l_query, s[0] + s[2]
        else:
            sc0 = reg["score"]
            qe = s[1] + s[2]
            re = s[0] + s[2] - rmax[0]
            assert re >= 0
            for i in range(2):
                prev = reg["score"]
                aw[1] = opt["w"] << i
                reg["score"], qle, tle, gtle, gscore, max_off = X.extend2(query[qe:], rseq[re:rmax[1] - rmax[0]], mat, opt, aw[1], opt["pen_clip3"], sc0)
                if reg["score"] == prev or max_off < (aw[1] >> 1) + (aw[1] >> 2):
                    break
            if gscore <= 0 or gscore <= reg["score"] - opt["pen_clip3"]:
                reg["qe"], reg["re"] = qe + qle, rmax[0] + re + tle
                reg["truesc"] += reg["score"] - sc0
            else:
                reg["qe"], reg["re"] = l_query, rmax[0] + re + gtle
                reg["truesc"] += gscore - sc0
        reg["bss"] = _getbss(parent, l_pac, reg["rb"])
        reg["parent"] = parent
        if _getbss(parent, l_pac, reg["re"]) != reg["bss"]:
            continue
        reg["seedcov"] = sum(t[2] for t in seeds if t[1] >= reg["qb"] and t[1] + t[2] <= reg["qe"] and t[0] >= reg["rb"] and t[0] + t[2] <= reg["re"])
        reg["w"] = max(aw)
        reg["seedlen0"] = s[2]
        reg["frac_rep"] = frac_rep
        regs.append(reg)


def chain2region(opt, idx, s, parent, chns, regs):      # mem_chain2region, memchain.c:873-904
    reg0 = len(regs)
    for c in chns:
        if not c["seeds"]:
            continue
        r0, r1 = _reference_span(opt, s["l_seq"], idx.l_pac, c)
        rseq, r0, r1, rid = idx.fetch_seq(r0, c["seeds"][0][0], r1)
        n0 = len(regs)
        _chain2region1(opt, idx, rseq, (r0, r1), rid, s["l_seq"], s["seq"], c["seeds"], regs, parent, reg0, c["frac_rep"])
        if len(regs) == n0 and c["seeds_extra"]:
            _chain2region1(opt, idx, rseq, (r0, r1), rid, s["l_seq"], s["seq"], c["seeds_extra"], regs, parent, reg0, c["frac_rep"])


# =====================================================================================================================
def gen_cigar2(opt, idx, mat, w_, query, rb, re, parent, want_cigar):      # bis_bwa_gen_cigar2, bwa.c:290-428
    """-> None (rejected) or dict(score [, cigar, NM, md, ZC, ZR, bss_u])"""
    l_pac = idx.l_pac
    l_query = len(query)
    if l_query <= 0 or rb >= re or (rb < l_pac < re):
        return None
    rseq = idx.get_seq(rb, re)
    rlen = len(rseq)
    if re - rb != rlen:
        return None
    if rb >= l_pac:
        query = query[::-1]
        rseq = rseq[::-1]
    query = np.ascontiguousarray(query)
    rseq = np.ascontiguousarray(rseq)
    out = {}
    cigar = []
    if l_query == re - rb and w_ == 0:
        if want_cigar:
            cigar = [l_query << 4]
        out["score"] = int(sum(mat[int(rseq[i]) * 5 + int(query[i])] for i in range(l_query)))
    else:
        max_ins = int((((l_query + 1) >> 1) * mat[0] - opt["o_ins"]) / opt["e_ins"] + 1.)
        max_del = int((((l_query + 1) >> 1) * mat[0] - opt["o_del"]) / opt["e_del"] + 1.)
        max_gap = max(max_ins, max_del, 1)
        w = (max_gap + abs(rlen - l_query) + 1) >> 1
        w = min(w, w_)
        w = max(w, abs(rlen - l_query) + 3)
        out["score"], cigar = ref().global2(query, rseq, mat, opt, w, want_cigar)
    if not want_cigar:
        return out
    int2base = "ACGTN" if rb < l_pac else "TGCAN"
    md = []
    x = y = u = n_mm = n_gap = n_conv_ct = n_ret_c = n_conv_ga = n_ret_g = 0
    for k, cg in enumerate(cigar):
        op, ln = cg & 0xf, cg >> 4
        if op == 0:
            for i in range(ln):
                q_, r_ = int(query[x + i]), int(rseq[y + i])
                if q_ == r_:
                    if q_ == 1: n_ret_c += 1
                    if q_ == 2: n_ret_g += 1
                    u += 1
                else:
                    md.append("%d%s" % (u, int2base[r_]))
                    u = 0
                    if parent and q_ == 3 and r_ == 1:
                        n_conv_ct += 1
                    elif not parent and q_ == 0 and r_ == 2:
                        n_conv_ga += 1
                    else:
                        n_mm += 1
            x += ln
            y += ln
        elif op == 2:
            if 0 < k < len(cigar) - 1:
                md.append("%d^%s" % (u, "".join(int2base[int(rseq[y + i])] for i in range(ln))))
                u = 0
                n_gap += ln
            y += ln
        elif op == 1:
            x += ln
            n_gap += ln
    md.append("%d" % u)
    out.update(cigar=cigar, md="".join(md), NM=n_mm + n_gap, ZC=n_conv_ct if parent else n_conv_ga, ZR=n_ret_c if parent else n_ret_g,
               bss_u=1 if n_conv_ct == 0 and n_conv_ga == 0 else 0)
    return out


def _test_reg_concatenation(opt, idx, query, a, b):      # mem_test_reg_concatenation, mem_alnreg.c:63-117 -> (score, w)
    l_pac = idx.l_pac
    assert a["rid"] == b["rid"] and a["rb"] <= b["rb"]
    if a["rb"] < l_pac <= b["rb"]:
        return 0, 0
    if a["qb"] >= b["qb"] or a["qe"] >= b["qe"] or a["re"] >= b["re"]:
        return 0, 0
    w = abs((a["re"] - b["rb"]) - (a["qe"] - b["qb"]))
    r = abs((a["re"] - b["rb"]) / (b["re"] - a["rb"]) - (a["qe"] - b["qb"]) / (b["qe"] - a["qb"]))
    if a["re"] < b["rb"] or a["qe"] < b["qb"]:
        if w > opt["w"] << 1 or r >= f32(0.05):
            return 0, 0
    elif w > opt["w"] << 2 or r >= f32(f32(0.05) * 2):
        return 0, 0
    w += a["w"] + b["w"]
    w = min(w, opt["w"] << 2)
    g = gen_cigar2(opt, idx, opt["ctmat"] if a["parent"] else opt["gamat"], w, query[a["qb"]:b["qe"]], a["rb"], b["re"], a["parent"], False)
    assert g is not None      # the reference would read an unset score here
    score = g["score"]
    q_s = int((b["qe"] - a["qb"]) / ((b["qe"] - b["qb"]) + (a["qe"] - a["qb"])) * (b["score"] + a["score"]) + .499)
    r_s = int((b["re"] - a["rb"]) / ((b["re"] - b["rb"]) + (a["re"] - a["rb"])) * (b["score"] + a["score"]) + .499)
    if score / max(q_s, r_s) < f32(0.90):
        return 0, 0
    return score, w


def sort_deduplicate(opt, idx, query, regs):      # mem_sort_deduplicate, mem_alnreg.c:121-202; idx None = the call without merging
    if len(regs) <= 1:
        return
    X = ref()
    regs[:] = [regs[i] for i in X.order([r["re"] for r in regs])]      # alnreg_slt2: by END
    for r in regs:
        r["n_comp"] = 1
    mlr = opt["mask_level_redun"]
    for i in range(1, len(regs)):
        p = regs[i]
        j = i - 1
        while j >= 0 and p["rid"] == regs[j]["rid"] and p["rb"] < regs[j]["re"] + opt["max_chain_gap"]:
            q = regs[j]
            j -= 1
            if q["qe"] == q["qb"]:
                continue
            orr = q["re"] - p["rb"]
            oq = q["qe"] - p["qb"] if q["qb"] < p["qb"] else p["qe"] - q["qb"]
            mr = min(q["re"] - q["rb"], p["re"] - p["rb"])
            mq = min(q["qe"] - q["qb"], p["qe"] - p["qb"])
            if f32(float(orr)) > f32(mlr * f32(float(mr))) and f32(float(oq)) > f32(mlr * f32(float(mq))):
                if p["score"] < q["score"]:
                    p["qe"] = p["qb"]
                    break
                q["qe"] = q["qb"]
            elif q["rb"] < p["rb"] and idx is not None:
                score, w = _test_reg_concatenation(opt, idx, query, q, p)
                if score > 0:
                    p["n_comp"] += q["n_comp"] + 1
                    p["seedcov"] = max(p["seedcov"], q["seedcov"])
                    p["sub"] = max(p["sub"], q["sub"])
                    p["csub"] = max(p["csub"], q["csub"])
                    p["truesc"] = p["score"] = score
                    p["qb"], p["rb"], p["w"] = q["qb"], q["rb"], w
                    q["qb"] = q["qe"]
    regs[:] = [r for r in regs if r["qe"] > r["qb"]]
    regs[:] = [regs[i] for i in X.order_tuples([(-r["score"], r["rb"], r["qb"]) for r in regs])]      # alnreg_slt
    for i in range(1, len(regs)):
        if regs[i]["score"] == regs[i - 1]["score"] and regs[i]["rb"] == regs[i - 1]["rb"] and regs[i]["qb"] == regs[i - 1]["qb"]:
            regs[i]["qe"] = regs[i]["qb"]
    regs[:] = [r for i, r in enumerate(regs) if i == 0 or r["qe"] > r["qb"]]


def merge_regions(opt, idx, s, regs):      # mem_merge_regions, mem_alnreg.c:214-238
    sort_deduplicate(opt, idx, s["seq"], regs)
    if opt["flag"] & MEM_F_SELF_OVLP and regs and regs[0]["truesc"] == s["l_seq"] * opt["a"]:      # mem_test_and_remove_exact
        del regs[0]
    for p in regs:
        if p["rid"] >= 0 and idx.anns[p["rid"]]["is_alt"]:
            p["is_alt"] = 1


# =====================================================================================================================
def _matesw_core(opt, idx, pes, reg, ms, mregs):      # mem_alnreg_matesw_core, mem_alnreg.c:395-491
    l_pac = idx.l_pac
    l_ms = len(ms)
    for m in mregs:
        ins = alnreg_isize(l_pac, reg, m)
        if ins is not None and pes["low"] <= ins <= pes["high"]:
            return
    rev = np.where(ms < 4, 3 - ms, 4).astype(np.uint8)[::-1]
    rb = max(0, reg["rb"] + pes["low"] - l_ms)
    re = min(l_pac << 1, reg["rb"] + pes["high"])
    rid, rseq = -1, None
    if rb < re:
        rseq, rb, re, rid = idx.fetch_seq(rb, (rb + re) >> 1, re)
    if reg["rid"] != rid or re - rb < opt["min_seed_len"]:
        return
    parent = reg["bss"] ^ (1 if reg["rb"] < l_pac else 0)
    xtra = KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if l_ms * opt["a"] < 250 else 0) | (opt["min_seed_len"] * opt["a"])
    aln = ref().align2(rev, rseq, opt["gamat"] if parent else opt["ctmat"], opt, xtra)
    if aln["score"] >= opt["min_seed_len"] and aln["qb"] >= 0:
        b = new_reg()
        b.update(rid=reg["rid"], is_alt=reg["is_alt"], qb=l_ms - (aln["qe"] + 1), qe=l_ms - aln["qb"], rb=(l_pac << 1) - (rb + aln["te"] + 1),
                 re=(l_pac << 1) - (rb + aln["tb"]), score=aln["score"], csub=aln["score2"], secondary=-1, bss=reg["bss"], parent=1 - parent)
        b["seedcov"] = min(b["re"] - b["rb"], b["qe"] - b["qb"]) >> 1
        at = len(mregs)
        for i, m in enumerate(mregs):
            if m["score"] < b["score"]:
                at = i
                break
        mregs.insert(at, b)
        sort_deduplicate(opt, None, None, mregs)


def matesw(opt, idx, pes, s, regs_pair):      # mem_alnreg_matesw, mem_alnreg.c:494-513
    good = [[dict(r) for r in regs_pair[i] if r["score"] >= regs_pair[i][0]["score"] - opt["pen_unpaired"]] for i in range(2)]
    for i in range(2):
        for j in range(min(len(good[i]), opt["max_matesw"])):
            _matesw_core(opt, idx, pes, good[i][j], s[1 - i]["seq"], regs_pair[1 - i])


# =====================================================================================================================
def set_sam(opt, idx, s, reg):      # mem_alnreg_setSAM, mem_alnreg_format.c:40-123
    if reg["cigar"]:
        return
    R = ref().R
    query = np.where(s["seq"] < 5, s["seq"], 4).astype(np.uint8)
    w1 = R.ref_infer_bw(reg["qe"] - reg["qb"], reg["re"] - reg["rb"], reg["truesc"], opt["a"], opt["o_del"], opt["e_del"])
    w2 = R.ref_infer_bw(reg["qe"] - reg["qb"], reg["re"] - reg["rb"], reg["truesc"], opt["a"], opt["o_ins"], opt["e_ins"])
    w = max(w1, w2)
    if w > opt["w"]:
        w = min(w, reg["w"])
    last_sc = -(1 << 30)
    g = None
    for i in range(3):
        w = min(w, opt["w"] << 2)
        g = gen_cigar2(opt, idx, opt["ctmat"] if reg["parent"] else opt["gamat"], w, query[reg["qb"]:reg["qe"]], reg["rb"], reg["re"], reg["parent"], True)
        assert g is not None
        score = g["score"]
        reg.update(NM=g["NM"], ZC=g["ZC"], ZR=g["ZR"], bss_u=g["bss_u"])
        if score == last_sc or w == opt["w"] << 2 or score >= reg["truesc"] - opt["a"]:
            break
        w <<= 1
        last_sc = score
    cigar, md = list(g["cigar"]), g["md"]
    rpos, is_rev = idx.depos(reg["rb"] if reg["rb"] < idx.l_pac else reg["re"] - 1)
    reg["is_rev"] = is_rev
    if is_rev:
        reg["flag"] |= 0x10
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
    reg["cigar"], reg["md"] = cigar, md
    assert idx.pos2rid(rpos) == reg["rid"]
    reg["pos"] = rpos - idx.anns[reg["rid"]]["offset"]


def get_rlen(cigar):      # bwamem.h:200-208
    return sum(c >> 4 for c in cigar if (c & 0xf) in (0, 2))


def _pri_idx(opt, regs, i):      # get_pri_idx, mem_alnreg.h:127-131 (XA_drop_ratio arrives as a double)
    k = regs[i]["secondary_all"]
    if k >= 0 and regs[i]["score"] >= regs[k]["score"] * float(opt["XA_drop_ratio"]):
        return k
    return -1


def _cg(cigar, alphabet):
    return "".join("%d%s" % (c >> 4, alphabet[c & 0xf]) for c in cigar)


def _tag_xaxb(opt, idx, s, p0, regs0):      # mem_alnreg_tagXAXB, mem_alnreg_format.c:126-191
    if regs0 is None or opt["flag"] & MEM_F_ALL:
        return ""
    mine = [i for i in range(len(regs0)) if (lambda r: r >= 0 and regs0[r] is p0)(_pri_idx(opt, regs0, i))]
    cnt_alt = sum(1 for i in mine if regs0[i]["is_alt"])
    cnt_pri = len(mine) - cnt_alt
    out = ""
    if cnt_pri <= opt["max_XA_hits"] and cnt_alt <= opt["max_XA_hits_alt"]:
        parts = []
        for i in mine:
            q = regs0[i]
            if not q["cigar"]:
                set_sam(opt, idx, s, q)
                if not q["cigar"]:
                    continue
            parts.append("%s,%s%d,%s,%d" % (idx.anns[q["rid"]]["name"], "+-"[q["is_rev"]], q["pos"] + 1, _cg(q["cigar"], "MIDSHN"), q["NM"]))
        if parts:
            out += "\tXA:Z:" + ";".join(parts)
    if cnt_pri > 0 or cnt_alt > 0:
        out += "\tXB:Z:%d,%d" % (cnt_pri, cnt_alt)
    return out


def _tag_sa(idx, p0, regs0):      # mem_alnreg_tagSA, mem_alnreg_format.c:194-228
    if regs0 is None or p0["flag"] & 0x100:
        return ""
    out = ""
    for q in regs0:
        if q is p0 or not q["cigar"] or q["flag"] & 0x100:
            continue
        out += "%s,%d,%s,%s,%d,%d;" % (idx.anns[q["rid"]]["name"], q["pos"] + 1, "+-"[q["is_rev"]], _cg(q["cigar"], "MIDSH"), q["mapq"], q["NM"])
    return "\tSA:Z:" + out if out else ""


def format_sam(opt, idx, s, p0, m0, regs0, is_primary, pes, rg_id):      # mem_alnreg_formatSAM, mem_alnreg_format.c:237-436
    l_pac = idx.l_pac
    p = dict(p0)
    m = dict(m0) if m0 is not None else new_reg()
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

    def cigar_text(r):
        out = []
        for c in r["cigar"]:
            op = c & 0xf
            if not (opt["flag"] & MEM_F_SOFTCLIP) and not r["is_alt"] and op in (3, 4):
                op = 3 if is_primary else 4
            out.append("%d%s" % (c >> 4, "MIDSH"[op]))
        return "".join(out)

    f = [s["name"] + ("_" + s["comment"] if s["comment"] else ""), str((p["flag"] & 0xffff) | (0x100 if p["flag"] & 0x10000 else 0))]
    if p["rid"] >= 0:
        f += [idx.anns[p["rid"]]["name"], str(p["pos"] + 1), str(p["mapq"]), cigar_text(p) if p["cigar"] else "*"]
    else:
        f += ["*", "0", "0", "*"]
    if has_m and m["rid"] >= 0:
        f += ["=" if p["rid"] == m["rid"] else idx.anns[m["rid"]]["name"], str(m["pos"] + 1)]
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
        qb, qe = 0, s["l_seq0"]
        if p["cigar"] and not is_primary and not (opt["flag"] & MEM_F_SOFTCLIP) and not p["is_alt"]:
            c0, c1 = p["cigar"][0], p["cigar"][-1]
            if p["is_rev"]:
                if (c0 & 0xf) in (3, 4): qe -= c0 >> 4
                if (c1 & 0xf) in (3, 4): qb += c1 >> 4
            else:
                if (c0 & 0xf) in (3, 4): qb += c0 >> 4
                if (c1 & 0xf) in (3, 4): qe -= c1 >> 4
        if p["is_rev"]:
            f.append("".join("TGCAN"[v] for v in s["seq0"][qb:qe][::-1]))
            f.append(s["qual"][qb:qe][::-1] if s["qual"] is not None else "*")
        else:
            f.append("".join("ACGTN"[v] for v in s["seq0"][qb:qe]))
            f.append(s["qual"][qb:qe] if s["qual"] is not None else "*")
    line = "\t".join(f)
    if p["cigar"]:
        line += "\tNM:i:%d\tMD:Z:%s\tZC:i:%d\tZR:i:%d" % (p["NM"], p["md"], p["ZC"], p["ZR"])
    if p["score"] >= 0:
        line += "\tAS:i:%d" % p["score"]
    if p["sub"] >= 0:
        line += "\tXS:i:%d" % max(p["sub"], p["csub"])
    if rg_id:
        line += "\tRG:Z:" + rg_id
    line += _tag_sa(idx, p0, regs0)
    if is_primary and p["alt_sc"] > 0:
        line += "\tPA:f:%.3f" % (p["score"] / p["alt_sc"])
    line += "\tXL:i:%d" % s["l_seq"]
    line += _tag_xaxb(opt, idx, s, p0, regs0)
    if opt["flag"] & MEM_F_REF_HDR and p["rid"] >= 0 and idx.anns[p["rid"]]["anno"]:
        line += "\tXR:Z:" + idx.anns[p["rid"]]["anno"].replace("\t", " ")
    if s["barcode"]:
        line += "\tCB:Z:" + s["barcode"]
    if s["umi"]:
        line += "\tRX:Z:" + s["umi"]
    line += "\tMC:Z:" + (cigar_text(m) if m["cigar"] else "*")
    line += "\tMQ:i:%d" % m["mapq"]
    line += "\tYD:A:" + ("u" if p["bss_u"] else "fr"[p["bss"]])
    return line + "\n"


def select_format(opt, idx, s, regs):      # mem_alnreg_select_format, mem_alnreg_format.c:445-488
    out = []
    for k, p in enumerate(regs):
        if p["rb"] < 0 or p["re"] < 0:
            continue
        if p["score"] < opt["T"]:
            continue
        if p["secondary"] >= 0 and (p["is_alt"] or not (opt["flag"] & MEM_F_ALL)):
            continue
        if 0 <= p["secondary"] < INT_MAX and f32(float(p["score"])) < f32(f32(float(regs[p["secondary"]]["score"])) * opt["drop_ratio"]):
            continue
        if out and p["secondary"] < 0:
            p["flag"] |= 0x10000 if opt["flag"] & MEM_F_NO_MULTI else 0x800
        if p["secondary"] >= 0:
            p["flag"] |= 0x100
        p["mapq"] = ref().mapq_se(opt, p) if p["secondary"] < 0 else 0
        if not (opt["flag"] & MEM_F_KEEP_SUPP_MAPQ) and out and not p["is_alt"]:
            p["mapq"] = min(p["mapq"], regs[0]["mapq"])
        set_sam(opt, idx, s, p)
        out.append(k)
    return out


def reg2sam_se(opt, idx, s, regs, rg_id):      # mem_reg2sam_se, mem_alnreg_format.c:492-516
    sel = select_format(opt, idx, s, regs)
    if sel:
        return "".join(format_sam(opt, idx, s, regs[k], None, regs, j == 0, None, rg_id) for j, k in enumerate(sel))
    reg = new_reg()
    reg.update(rid=-1, flag=0x4)
    return format_sam(opt, idx, s, reg, None, regs, True, None, rg_id)


def _reg2sam_pe_nopairing(opt, idx, s, regs_pair, pes, rg_id):      # mem_alnreg_format.c:519-559
    best = [None, None]
    sel = [None, None]
    for i in range(2):
        sel[i] = select_format(opt, idx, s[i], regs_pair[i])
        if sel[i]:
            best[i] = regs_pair[i][sel[i][0]]
        else:
            best[i] = new_reg()
            best[i].update(rid=-1, flag=0x40 << i | 0x1 | 0x4)
    out = []
    for i in range(2):
        if sel[i]:
            out.append("".join(format_sam(opt, idx, s[i], regs_pair[i][k], best[1 - i], regs_pair[i], j == 0, pes, rg_id) for j, k in enumerate(sel[i])))
        else:
            out.append(format_sam(opt, idx, s[i], best[i], best[1 - i], None, True, pes, rg_id))
    return out


def _raw_mapq(diff, a):
    return int(6.02 * diff / a + .499)


def reg2sam_pe(opt, idx, pid, s, regs_pair, n_pri, pes, rg_id):      # mem_reg2sam_pe, mem_alnreg_format.c:562-696
    X = ref()
    for i in range(2):
        for p in regs_pair[i]:
            p["flag"] |= (0x40 << i) | 1
    if opt["flag"] & MEM_F_NOPAIRING or n_pri[0] == 0 or n_pri[1] == 0:
        return _reg2sam_pe_nopairing(opt, idx, s, regs_pair, pes, rg_id)
    for i in range(2):
        for j in range(1, n_pri[i]):
            if regs_pair[i][j]["secondary"] < 0 and regs_pair[i][j]["score"] >= opt["T"]:
                return _reg2sam_pe_nopairing(opt, idx, s, regs_pair, pes, rg_id)
    pscore, sub_pscore, n_sub, z0, z1 = pair(opt, idx.l_pac, idx.offsets, pes, regs_pair, n_pri, pid)
    z = [z0, z1]
    if pscore <= 0:
        return _reg2sam_pe_nopairing(opt, idx, s, regs_pair, pes, rg_id)
    score_unpaired = regs_pair[0][0]["score"] + regs_pair[1][0]["score"] - opt["pen_unpaired"]
    if pscore > score_unpaired:
        sub_pscore = max(sub_pscore, score_unpaired)
        q_pe = _raw_mapq(pscore - sub_pscore, opt["a"])
        if n_sub > 0:
            q_pe -= int(4.343 * math.log(n_sub + 1) + .499)
        q_pe = max(0, min(60, q_pe))
        q_pe = int(q_pe * (1. - .5 * f32(regs_pair[0][0]["frac_rep"] + regs_pair[1][0]["frac_rep"])) + .499)
        c = [regs_pair[0][z[0]], regs_pair[1][z[1]]]
        q_se = [0, 0]
        for i in range(2):
            if c[i]["secondary"] >= 0:
                c[i]["sub"] = regs_pair[i][c[i]["secondary"]]["score"]
                c[i]["secondary"] = -2
            q_se[i] = X.mapq_se(opt, c[i])
        for i in range(2):
            q_se[i] = max(q_se[i], min(q_pe, q_se[i] + 40))
            c[i]["mapq"] = min(q_se[i], _raw_mapq(c[i]["score"] - c[i]["csub"], opt["a"]))
    else:
        z = [0, 0]
        for i in range(2):
            regs_pair[i][0]["mapq"] = X.mapq_se(opt, regs_pair[i][0])
    for i in range(2):
        regs = regs_pair[i]
        k = regs[z[i]]["secondary_all"]
        if 0 <= k < n_pri[i]:
            assert regs[k]["secondary_all"] < 0
            for j, r in enumerate(regs):
                if r["secondary_all"] == k or j == k:
                    r["secondary_all"] = z[i]
            regs[z[i]]["secondary_all"] = -1
    for i in range(2):
        set_sam(opt, idx, s[i], regs_pair[i][z[i]])
    out = []
    for i in range(2):
        regs = regs_pair[i]
        txt = format_sam(opt, idx, s[i], regs[z[i]], regs_pair[1 - i][z[1 - i]], regs, True, pes, rg_id)
        if n_pri[i] < len(regs):
            p = regs[n_pri[i]]
            if p["score"] >= opt["T"] and p["secondary"] < 0:
                p["flag"] |= 0x800
                set_sam(opt, idx, s[i], p)
                txt += format_sam(opt, idx, s[i], p, None, regs, False, pes, rg_id)
        out.append(txt)
    return out


# =====================================================================================================================
def bsconvert(seq, parent):      # bseq_bsconvert, bwamem.c:161-181
    out = seq.copy()
    if parent:
        out[seq == 1] = 3
    else:
        out[seq == 2] = 0
    return out


def clip_read(opt, s, adaptor):      # read_clipping, bwamem.c:286-303 -- the real function
    out = (C.c_int * 5)()
    ad = adaptor if adaptor is not None else None
    seq0 = np.ascontiguousarray(s["seq0"])
    ref().R.ref_read_clipping(len(seq0), _p(seq0), s["qual"].encode() if s["qual"] is not None else None, _p(ad) if ad is not None else None,
                              len(ad) if ad is not None else 0, opt["clip5"], opt["clip3"], opt["min_base_qual"], out)
    s["l_adaptor"], s["clip5"], s["clip3"], s["l_seq"] = out[0], out[1], out[2], out[3]
    s["seq"] = seq0[out[4]:out[4] + out[3]] if out[3] > 0 else seq0[:0]
    s["bisseq"] = [None, None]


def align1_core(opt, idx, s, regs, parent):      # mem_align1_core, bwamem.c:183-208
    if s["bisseq"][parent] is None:
        s["bisseq"][parent] = bsconvert(s["seq"], parent)
    chns = chain(opt, idx, s, parent)
    chns = chain_flt(opt, chns)
    flt_chained_seeds(opt, idx, s, chns, parent)
    chain2region(opt, idx, s, parent, chns, regs)


def _check_names(n1, n2):      # check_paired_read_names, bwamem.c:210-216
    if n1 == n2:
        return
    if n1[-1:] == "1" and n2[len(n1) - 1:len(n1)] == "2" and n1[:-1] == n2[:len(n1) - 1]:
        return
    raise RuntimeError('paired reads have different names: "%s", "%s"' % (n1, n2))


def worker1(opt, idx, seqs, i):      # bis_worker1, bwamem.c:311-376 -> the region list(s) of read i (SE) or of pair i (PE)
    if not (opt["flag"] & MEM_F_PE):
        s = seqs[i]
        clip_read(opt, s, opt["adaptor1"])
        regs = []
        if not (opt["parent"] & 1) or opt["parent"] >> 1:
            align1_core(opt, idx, s, regs, 0)
        if not (opt["parent"] & 1) or not (opt["parent"] >> 1):
            align1_core(opt, idx, s, regs, 1)
        merge_regions(opt, idx, s, regs)
        return [regs]
    s0, s1 = seqs[i << 1], seqs[i << 1 | 1]
    _check_names(s0["name"], s1["name"])
    clip_read(opt, s0, opt["adaptor1"])
    clip_read(opt, s1, opt["adaptor2"])
    r0 = []
    align1_core(opt, idx, s0, r0, 1)
    if not opt["parent"]:
        align1_core(opt, idx, s0, r0, 0)
    merge_regions(opt, idx, s0, r0)
    r1 = []
    align1_core(opt, idx, s1, r1, 0)
    if not opt["parent"]:
        align1_core(opt, idx, s1, r1, 1)
    merge_regions(opt, idx, s1, r1)
    return [r0, r1]


def worker2(opt, idx, seqs, regs, i, n_processed, pes, rg_id):      # bis_worker2, bwamem.c:382-424 -> SAM text of read i / pair i
    if not (opt["flag"] & MEM_F_PE):
        mark_primary_se(opt, regs[i], n_processed + i)
        for r in regs[i]:
            r["flag"] = 0
        return [reg2sam_se(opt, idx, seqs[i], regs[i], rg_id)]
    pr = [regs[i << 1], regs[i << 1 | 1]]
    sp = [seqs[i << 1], seqs[i << 1 | 1]]
    if not (opt["flag"] & MEM_F_NO_RESCUE):
        matesw(opt, idx, pes, sp, pr)
    n_pri = [mark_primary_se(opt, pr[0], i << 1), mark_primary_se(opt, pr[1], i << 1 | 1)]
    for r in pr[0] + pr[1]:
        r["flag"] = 0
    return reg2sam_pe(opt, idx, (n_processed >> 1) + i, sp, pr, n_pri, pes, rg_id)


_G = {}


def _w1(i):
    regs = worker1(_G["opt"], _G["idx"], _G["seqs"], i)
    pe = bool(_G["opt"]["flag"] & MEM_F_PE)
    ss = [_G["seqs"][i << 1], _G["seqs"][i << 1 | 1]] if pe else [_G["seqs"][i]]
    return regs, [{k: s[k] for k in ("l_adaptor", "clip5", "clip3", "l_seq")} for s in ss]


def _w2(i):
    return worker2(_G["opt"], _G["idx"], _G["seqs"], _G["regs"], i, _G["n_processed"], _G["pes"], _G["rg_id"])


def _reclip(s, c):
    s.update(c)
    off = c["clip5"]
    s["seq"] = s["seq0"][off:off + c["l_seq"]] if c["l_seq"] > 0 else s["seq0"][:0]
    s["bisseq"] = [None, None]


def process_seqs(opt, idx, n_processed, seqs, pes0, rg_id, procs=1):      # mem_process_seqs, bwamem.c:432-476 -> SAM text per read
    pe = bool(opt["flag"] & MEM_F_PE)
    n = len(seqs)
    n_units = n >> 1 if pe else n
    regs = [None] * n
    if procs > 1 and n_units > 8:
        import multiprocessing as mp
        _G.update(opt=opt, idx=idx, seqs=seqs)
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_w1, range(n_units), chunksize=max(1, n_units // (procs * 8)))
        for i, (rr, clips) in enumerate(res):
            for j, (r, c) in enumerate(zip(rr, clips)):
                k = (i << 1 | j) if pe else i
                regs[k] = r
                _reclip(seqs[k], c)
    else:
        for i in range(n_units):
            rr = worker1(opt, idx, seqs, i)
            for j, r in enumerate(rr):
                regs[(i << 1 | j) if pe else i] = r
    pes = None
    if pe:
        pes = dict(pes0) if pes0 is not None else pestat(opt, idx.l_pac, regs)
    if procs > 1 and n_units > 8:
        _G.update(regs=regs, n_processed=n_processed, pes=pes, rg_id=rg_id)
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_w2, range(n_units), chunksize=max(1, n_units // (procs * 8)))
    else:
        res = [worker2(opt, idx, seqs, regs, i, n_processed, pes, rg_id) for i in range(n_units)]
    return [t for r in res for t in r]


# =====================================================================================================================
def _kseq_next(R, ks, buf):
    r = R.ref_kseq_next(ks, buf, len(buf))
    if r < 0:
        assert r != -3
        return None
    name, comment, seq, qual = buf.value.decode("latin-1").rstrip("\n").split("\t")
    return name, (None if comment == "*" else comment), seq, (None if qual == "*" else qual)


def _bseq1(rec, has_bc):      # trim_readno + bis_kseq2bseq1, bwa.c:58-63,764-815
    name, comment, seq, qual = rec
    if len(name) > 2 and name[-2] == "/" and name[-1].isdigit():
        name = name[:-2]
    bc = umi = None
    if has_bc:
        toks = [t for t in name.split("_") if t]      # strtok skips empty fields
        bc = toks[1] if len(toks) > 1 else None
        umi = toks[2] if len(toks) > 2 else None
        for t in toks[3:]:
            bc, umi = umi, t
    seq0 = NT4[np.frombuffer(seq.encode("latin-1"), np.uint8)].copy()
    return {"name": name, "comment": comment, "seq0": seq0, "l_seq0": len(seq0), "qual": qual, "barcode": bc, "umi": umi}


def read_chunk(R, chunk_size, has_bc, ks1, ks2, buf):      # bis_bseq_read, bwa.c:817-850
    seqs = []
    size = 0
    while True:
        r1 = _kseq_next(R, ks1, buf)
        if r1 is None:
            break
        r2 = None
        if ks2:
            r2 = _kseq_next(R, ks2, buf)
            if r2 is None:
                break
        s = _bseq1(r1, has_bc)
        seqs.append(s)
        size += s["l_seq0"]
        if ks2:
            s = _bseq1(r2, has_bc)
            seqs.append(s)
            size += s["l_seq0"]
        if size >= chunk_size and (len(seqs) & 1) == 0:
            break
    return seqs


def _classify(seqs):      # bseq_classify, bwa.c:119-138 -> indices of the single reads and of the paired ones
    se, pe = [], []
    has_last = True
    i = 1
    while i < len(seqs):
        if has_last:
            if seqs[i]["name"] == seqs[i - 1]["name"]:
                pe += [i - 1, i]
                has_last = False
            else:
                se.append(i - 1)
        else:
            has_last = True
        i += 1
    if has_last and seqs:
        se.append(i - 1)
    return se, pe


def sam_header(idx, hdr_line):      # bwa_print_sam_hdr (bwa.c:654-684), the real function; bwa_pg unset so no @PG line
    n = len(idx.anns)
    names = (C.c_char_p * n)(*[a["name"].encode() for a in idx.anns])
    lens = (C.c_int * n)(*[a["len"] for a in idx.anns])
    buf = C.create_string_buffer(1 << 20)
    r = ref().R.ref_sam_hdr(n, names, lens, hdr_line.encode() if hdr_line else None, None, buf, len(buf))
    assert r >= 0
    return buf.raw[:r].decode()


def align(argv, out, procs=1):      # main_align + process, align.c:70-167,319-598
    opt, pes0, hdr_line, rg_id, copy_comment, pos, ignore_alt, infer_alt = parse_args(argv)
    assert 2 <= len(pos) <= 3
    R = ref().R
    idx = Index(pos[0], ignore_alt, infer_alt)
    ks1 = R.ref_kseq_open(pos[1].encode())
    assert ks1
    ks2 = None
    if len(pos) > 2 and not (opt["flag"] & MEM_F_PE):
        ks2 = R.ref_kseq_open(pos[2].encode())
        assert ks2
        opt["flag"] |= MEM_F_PE
    if not (opt["flag"] & MEM_F_ALN_REG):
        out.write(sam_header(idx, hdr_line))
    chunk = opt["chunk_size"] * opt["n_threads"]
    buf = C.create_string_buffer(1 << 22)
    n_processed = 0
    while True:
        seqs = read_chunk(R, chunk, opt["has_bc"], ks1, ks2, buf)
        if not seqs:
            break
        if not copy_comment:
            for s in seqs:
                s["comment"] = None
        if opt["flag"] & MEM_F_SMARTPE:
            se, pe = _classify(seqs)
            sam = [None] * len(seqs)
            if se:
                o = dict(opt, flag=opt["flag"] & ~MEM_F_PE)
                for k, t in zip(se, process_seqs(o, idx, n_processed, [seqs[k] for k in se], None, rg_id, procs)):
                    sam[k] = t
            if pe:
                o = dict(opt, flag=opt["flag"] | MEM_F_PE)
                for k, t in zip(pe, process_seqs(o, idx, n_processed + len(se), [seqs[k] for k in pe], pes0, rg_id, procs)):
                    sam[k] = t
        else:
            sam = process_seqs(opt, idx, n_processed, seqs, pes0, rg_id, procs)
        n_processed += len(seqs)
        for t in sam:
            if t:
                out.write(t)
    R.ref_kseq_close(ks1)
    if ks2:
        R.ref_kseq_close(ks2)


if __name__ == "__main__":
    _procs = int(os.environ.get("E2E_PROCS", "1"))
    align(sys.argv[1:], sys.stdout, _procs)
