"""ctypes binding of the C-ABI in include/bsx.h (the in-tree libbiscuit_amd.so).

PyTorch is not involved in the data path: the shared library owns device memory and streams.
This module is plumbing for tests/ and bench.py; the product interface is the C ABI and the
`biscuit_align` command line.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSX_LIB_PATH") or os.path.join(_HERE, "libbiscuit_amd.so")   # ($BSX_LIB_PATH: a debug build of the same library, tools/dbg/)


class Opt(C.Structure):  # bsx_opt_t == mem_opt_t (lib/aln/bwamem.h:54-124)
    _fields_ = [
        ("a", C.c_int), ("b", C.c_int), ("o_del", C.c_int), ("e_del", C.c_int), ("o_ins", C.c_int), ("e_ins", C.c_int),
        ("pen_unpaired", C.c_int), ("pen_clip5", C.c_int), ("pen_clip3", C.c_int), ("w", C.c_int), ("zdrop", C.c_int),
        ("max_mem_intv", C.c_uint64), ("T", C.c_int), ("flag", C.c_int), ("min_seed_len", C.c_int),
        ("min_chain_weight", C.c_int), ("max_chain_extend", C.c_uint32), ("split_factor", C.c_float),
        ("split_width", C.c_int), ("max_occ", C.c_uint32), ("max_chain_gap", C.c_int), ("n_threads", C.c_int),
        ("chunk_size", C.c_int), ("mask_level", C.c_float), ("drop_ratio", C.c_float), ("XA_drop_ratio", C.c_float),
        ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float), ("mapQ_coef_fac", C.c_int), ("max_ins", C.c_int),
        ("max_matesw", C.c_int), ("max_XA_hits", C.c_int), ("max_XA_hits_alt", C.c_int), ("mat", C.c_int8 * 25),
        ("parent", C.c_uint8), ("bsstrand", C.c_uint8), ("ctmat", C.c_int8 * 25), ("gamat", C.c_int8 * 25),
        ("adaptor1", C.POINTER(C.c_uint8)), ("l_adaptor1", C.c_int), ("adaptor2", C.POINTER(C.c_uint8)), ("l_adaptor2", C.c_int),
        ("clip5", C.c_int), ("clip3", C.c_int), ("min_base_qual", C.c_int), ("has_bc", C.c_uint8),
    ]


class PeStat(C.Structure):
    _fields_ = [("low", C.c_int), ("high", C.c_int), ("set", C.c_int), ("failed", C.c_int), ("avg", C.c_double), ("std", C.c_double)]


class Read(C.Structure):  # bsx_read_t
    _fields_ = [("l_seq", C.c_int), ("id", C.c_int), ("name", C.c_char_p), ("comment", C.c_char_p), ("barcode", C.c_char_p),
                ("umi", C.c_char_p), ("qual", C.c_char_p), ("sam", C.c_void_p), ("seq", C.POINTER(C.c_uint8)),
                ("seq0", C.POINTER(C.c_uint8)), ("l_seq0", C.c_int), ("l_adaptor", C.c_int), ("clip5", C.c_int), ("clip3", C.c_int)]


class Intv(C.Structure):
    _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]


class SeedTask(C.Structure):
    _fields_ = [("qoff", C.c_uint32), ("len", C.c_int32), ("parent", C.c_int32)]


class SaJob(C.Structure):
    _fields_ = [("k", C.c_uint64), ("parent", C.c_int32), ("pad", C.c_int32)]


class ExtJob(C.Structure):
    _fields_ = [("tpos", C.c_int64), ("qoff", C.c_uint32), ("qlen", C.c_int32), ("tlen", C.c_int32), ("h0", C.c_int32),
                ("w", C.c_int32), ("end_bonus", C.c_int32), ("qdir", C.c_int8), ("tdir", C.c_int8), ("parent", C.c_uint8), ("pad", C.c_uint8)]


class ExtRes(C.Structure):
    _fields_ = [("score", C.c_int32), ("qle", C.c_int32), ("tle", C.c_int32), ("gtle", C.c_int32), ("gscore", C.c_int32), ("max_off", C.c_int32)]


class SwJob(C.Structure):
    _fields_ = [("tpos", C.c_int64), ("qoff", C.c_uint32), ("qlen", C.c_int32), ("tlen", C.c_int32), ("xtra", C.c_int32),
                ("qdir", C.c_int8), ("tdir", C.c_int8), ("qcomp", C.c_uint8), ("use_ct", C.c_uint8)]


class SwRes(C.Structure):
    _fields_ = [("score", C.c_int32), ("te", C.c_int32), ("qe", C.c_int32), ("score2", C.c_int32), ("te2", C.c_int32), ("tb", C.c_int32), ("qb", C.c_int32)]


class GlbJob(C.Structure):
    _fields_ = [("tpos", C.c_int64), ("qoff", C.c_uint32), ("qlen", C.c_int32), ("tlen", C.c_int32), ("w0", C.c_int32),
                ("w_max", C.c_int32), ("truesc", C.c_int32), ("n_try", C.c_int32), ("cigar_off", C.c_uint32), ("cigar_cap", C.c_uint32),
                ("qdir", C.c_int8), ("tdir", C.c_int8), ("use_ct", C.c_uint8), ("want_cigar", C.c_uint8)]


class GlbRes(C.Structure):
    _fields_ = [("score", C.c_int32), ("n_cigar", C.c_int32), ("w_used", C.c_int32), ("pad", C.c_int32)]


class Region(C.Structure):  # bsx_region_t
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64), ("qb", C.c_int32), ("qe", C.c_int32), ("rid", C.c_int32), ("score", C.c_int32), ("truesc", C.c_int32),
                ("w", C.c_int32), ("seedcov", C.c_int32), ("seedlen0", C.c_int32), ("frac_rep", C.c_float), ("bss", C.c_uint8), ("parent", C.c_uint8), ("pad", C.c_uint8 * 2)]


class GlbTag(C.Structure):
    _fields_ = [("NM", C.c_int32), ("ZC", C.c_int32), ("ZR", C.c_int32), ("l_md", C.c_int32), ("md_off", C.c_uint64), ("bss_u", C.c_uint8), ("pad", C.c_uint8 * 7)]


class Backend(C.Structure):  # bsx_backend_t (csrc/host/bsx_core.h)
    _fields_ = [("ctx", C.c_void_p), ("name", C.c_char_p)] + [(n, C.c_void_p) for n in
                ("set_opt", "set_reads", "seed_batch", "sa_batch", "extend_batch", "sw_batch", "global_batch", "global_batch_tags", "regions_batch", "regions_finish", "regions_dedup")] + [("dedup_cap", C.c_int), ("regions_dedup2", C.c_void_p), ("msw_plan", C.c_void_p)]


class PhaseStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("t_seed", "t_sa", "t_chain", "t_extend", "t_merge", "t_pestat", "t_matesw",
                                          "t_primary", "t_cigar", "t_sam", "t_total", "t_prep", "t_cleanup", "t_regions")] + \
               [(n, C.c_int64) for n in ("n_tasks", "n_intv", "n_sa", "n_ext_jobs", "n_ext_rounds", "n_sw_jobs", "n_glb_jobs", "n_host_tasks", "n_redo_tasks")]


_lib = None


def lib():
    """Load libbiscuit_amd.so; raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `make` (or __graft_entry__.build()) first" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.bsx_version.restype = C.c_char_p
        L.bsx_strerror.restype = C.c_char_p
        L.bsx_device_name.restype = C.c_char_p
        L.bsx_index_l_pac.restype = C.c_int64
        L.bsx_index_l_pac.argtypes = [C.c_void_p]
        L.bsx_index_free.argtypes = [C.c_void_p]
        L.bsx_device_close.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def tune(name, value=None):
    """a setting of the library (csrc/host/tune.c has the table): value None = back to the default"""
    L = lib()
    L.bsx_tune_set.argtypes = [C.c_char_p, C.c_char_p]
    check(L.bsx_tune_set(name.encode(), None if value is None else str(value).encode()), "bsx_tune_set(%s)" % name)


def tune_names():
    L = lib()
    L.bsx_tune_name.restype = C.c_char_p
    out, i = [], 0
    while True:
        n = L.bsx_tune_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def tune_env(settings):
    """An environment for a child process from a dict of settings: names of the library's table (with or without the BSX_ prefix of the
    environment variables they were until round 5, any case) go into ONE variable, $BSX_TUNE="name=value,..."; everything else -- real
    environment variables such as BSX_STREAM_DEPTH -- passes through unchanged."""
    names = set(tune_names())
    env, tuned = {}, []
    for k, v in (settings or {}).items():
        if k == "BSX_TUNE":
            tuned.append(str(v))
            continue
        low = k[4:].lower() if k.upper().startswith("BSX_") else k.lower()
        if low in names and k not in ("BSX_PHASES",):
            tuned.append("%s=%s" % (low, v))
        else:
            env[k] = str(v)
    if tuned:
        env["BSX_TUNE"] = ",".join(tuned)
    return env


class BsxError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise BsxError("%s failed: %s (%d)" % (what, lib().bsx_strerror(rc).decode(), rc))
