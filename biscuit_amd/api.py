"""Thin Python mirror of the C ABI (include/bsx.h) over numpy buffers -- test/bench plumbing."""
import ctypes as C
import numpy as np
from . import _lib as B

SEED_DT = np.dtype(B.SeedTask)
SA_DT = np.dtype(B.SaJob)
EXT_DT = np.dtype(B.ExtJob)
EXTRES_DT = np.dtype(B.ExtRes)
SW_DT = np.dtype(B.SwJob)
SWRES_DT = np.dtype(B.SwRes)
GLB_DT = np.dtype(B.GlbJob)
GLBRES_DT = np.dtype(B.GlbRes)
INTV_DT = np.dtype(B.Intv)


def default_opt():
    o = B.Opt()
    B.lib().bsx_opt_init(C.byref(o))
    return o


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Index:
    """<base>.{par,dau}.{bwt,sa} + <base>.bis.{ann,amb,pac} (bwa_idx_load_from_disk, lib/aln/bwa.c:525)"""

    def __init__(self, base):
        self.base = base
        self.h = C.c_void_p()
        B.check(B.lib().bsx_index_load(base.encode(), C.byref(self.h)), "bsx_index_load(%s)" % base)
        self.l_pac = B.lib().bsx_index_l_pac(self.h)

    @staticmethod
    def build(fasta, base):
        B.check(B.lib().bsx_index_build(fasta.encode(), base.encode()), "bsx_index_build")
        return Index(base)

    @classmethod
    def _wrap(cls, h):
        self = cls.__new__(cls)
        self.base = None
        self.h = h
        self.l_pac = B.lib().bsx_index_l_pac(h)
        return self

    @classmethod
    def from_fasta(cls, fasta):
        """pac + annotation only (bis_bns_fasta2bntseq); the FM indices come from build_host() or Device.build_index()"""
        h = C.c_void_p()
        B.check(B.lib().bsx_index_from_fasta(fasta.encode(), C.byref(h)), "bsx_index_from_fasta")
        return cls._wrap(h)

    @classmethod
    def synthetic(cls, n_bases, seed, n_contigs=8, repeat_frac=0.05, profile=0):
        """seeded synthetic genome (csrc/host/sim.c) straight into an index without FM indices; profile 1 = with the high-copy
        interspersed repeat families of a mammalian genome (~43 % repeats)"""
        L = B.lib()
        L.bsx_sim_genome_index2.argtypes = [C.c_int64, C.c_uint64, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        B.check(L.bsx_sim_genome_index2(n_bases, seed, n_contigs, repeat_frac, profile, C.byref(h)), "bsx_sim_genome_index2")
        return cls._wrap(h)

    def build_host(self):
        B.check(B.lib().bsx_index_build_host(self.h), "bsx_index_build_host")

    def save(self, base):
        B.check(B.lib().bsx_index_save(self.h, base.encode()), "bsx_index_save")

    def close(self):
        if self.h:
            B.lib().bsx_index_free(self.h)
            self.h = None


class Batches:
    """The five kernel-level batch seams, bound either to the HIP device or (tests only) to the
    CPU restatement in oracle/."""

    def __init__(self, fns, ctx):
        self.f = fns
        self.ctx = ctx
        self._keep = None

    def set_opt(self, opt):
        B.check(self.f["set_opt"](self.ctx, C.byref(opt)), "set_opt")

    def set_reads(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._keep = buf
        B.check(self.f["set_reads"](self.ctx, _p(buf), C.c_size_t(buf.size)), "set_reads")

    def seed(self, opt, tasks):
        tasks = np.ascontiguousarray(tasks, dtype=SEED_DT)
        n = len(tasks)
        out = C.c_void_p()
        cap = C.c_int64(0)
        off = np.zeros(n + 1, dtype=np.int64)
        B.check(self.f["seed_batch"](self.ctx, C.byref(opt), C.c_int64(n), _p(tasks), C.byref(out), C.byref(cap), _p(off)), "seed_batch")
        tot = int(off[n])
        if tot:
            arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(tot, 4)).copy()
        else:
            arr = np.zeros((0, 4), dtype=np.uint64)
        if out.value:
            _libc_free(out)
        return arr, off

    def sa(self, jobs):
        jobs = np.ascontiguousarray(jobs, dtype=SA_DT)
        pos = np.zeros(len(jobs), dtype=np.uint64)
        B.check(self.f["sa_batch"](self.ctx, C.c_int64(len(jobs)), _p(jobs), _p(pos)), "sa_batch")
        return pos

    def extend(self, jobs):
        jobs = np.ascontiguousarray(jobs, dtype=EXT_DT)
        res = np.zeros(len(jobs), dtype=EXTRES_DT)
        B.check(self.f["extend_batch"](self.ctx, C.c_int64(len(jobs)), _p(jobs), _p(res)), "extend_batch")
        return res

    def sw(self, jobs):
        jobs = np.ascontiguousarray(jobs, dtype=SW_DT)
        res = np.zeros(len(jobs), dtype=SWRES_DT)
        B.check(self.f["sw_batch"](self.ctx, C.c_int64(len(jobs)), _p(jobs), _p(res)), "sw_batch")
        return res

    def global_(self, jobs, pool_len):
        jobs = np.ascontiguousarray(jobs, dtype=GLB_DT)
        res = np.zeros(len(jobs), dtype=GLBRES_DT)
        pool = np.zeros(max(1, pool_len), dtype=np.uint32)
        B.check(self.f["global_batch"](self.ctx, C.c_int64(len(jobs)), _p(jobs), _p(res), _p(pool), C.c_size_t(pool.size)), "global_batch")
        return res, pool


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _libc_free(p):
    _libc.free(p)


class Device(Batches):
    """bsx_device_* : one HIP device with the index resident in HBM.  Raises if there is no GPU."""

    def __init__(self, ordinal=0):
        L = B.lib()
        self.h = C.c_void_p()
        B.check(L.bsx_device_open(ordinal, C.byref(self.h)), "bsx_device_open")
        names = {"set_opt": "bsx_device_set_opt", "set_reads": "bsx_device_set_reads", "seed_batch": "bsx_seed_batch",
                 "sa_batch": "bsx_sa_batch", "extend_batch": "bsx_extend_batch", "sw_batch": "bsx_sw_batch",
                 "global_batch": "bsx_global_batch"}
        fns = {k: getattr(L, v) for k, v in names.items() if hasattr(L, v)}
        Batches.__init__(self, fns, self.h)
        self.name = L.bsx_device_name(self.h).decode()

    def upload_index(self, index):
        B.check(B.lib().bsx_device_upload_index(self.h, index.h), "bsx_device_upload_index")

    def build_index(self, index, fill_host=False):
        """both FM indices built on the device from index's pac and left resident (csrc/hip/k_index.hip)"""
        B.check(B.lib().bsx_device_build_index(self.h, index.h, int(fill_host)), "bsx_device_build_index")

    def regions(self, opt, tasks):
        """bsx_regions_batch + bsx_regions_finish: seeding through regions on the device -> (regions, offsets, counts per strand search;
        a negative count = declined)"""
        tasks = np.ascontiguousarray(tasks, dtype=SEED_DT)
        n = len(tasks)
        L = B.lib()
        out, cap, di, dc = C.c_void_p(), C.c_int64(0), C.c_void_p(), C.c_int64(0)
        off = np.zeros(n + 1, dtype=np.int64)
        cnt = np.zeros(n + 1, dtype=np.int32)
        doff = np.zeros(n + 1, dtype=np.int64)
        L.bsx_regions_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
        L.bsx_regions_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
        B.check(L.bsx_regions_batch(self.h, C.byref(opt), n, _p(tasks), C.byref(out), C.byref(cap), _p(off), _p(cnt), C.byref(di), C.byref(dc), _p(doff)), "bsx_regions_batch")
        B.check(L.bsx_regions_finish(self.h, C.byref(out), C.byref(cap), _p(off), _p(cnt)), "bsx_regions_finish")
        dt = np.dtype(B.Region)
        tot = int(max([off[i] + cnt[i] for i in range(n) if cnt[i] > 0] + [0]))
        regs = np.frombuffer(C.string_at(out.value, tot * dt.itemsize), dtype=dt).copy() if tot else np.zeros(0, dtype=dt)
        for ptr in (out, di):
            if ptr.value:
                _libc_free(ptr)
        return regs, off[:n], cnt[:n]

    def global_tags(self, jobs, pool_len):
        """bsx_global_batch_tags: K6 plus NM / MD / ZC / ZR of every job with a CIGAR -> (res, pool, tags, [md bytes or None])"""
        jobs = np.ascontiguousarray(jobs, dtype=GLB_DT)
        n = len(jobs)
        res = np.zeros(n, dtype=GLBRES_DT)
        pool = np.zeros(max(1, pool_len), dtype=np.uint32)
        tags = np.zeros(n, dtype=np.dtype(B.GlbTag))
        md = C.c_void_p()
        cap = C.c_int64(0)
        f = B.lib().bsx_global_batch_tags
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        B.check(f(self.h, n, _p(jobs), _p(res), _p(pool), pool.size, _p(tags), C.byref(md), C.byref(cap)), "bsx_global_batch_tags")
        out = []
        for k in range(n):
            if tags[k]["l_md"] < 0:
                out.append(None)
            else:
                out.append(C.string_at(md.value + int(tags[k]["md_off"]), int(tags[k]["l_md"]) + 1))
        if md.value:
            _libc_free(md)
        return res, pool, tags, out

    def counters(self, reset=False):
        c = (C.c_uint64 * 4)()
        B.check(B.lib().bsx_device_counters(self.h, c, int(reset)), "bsx_device_counters")
        return list(c)

    def seed_passes(self, reset=False):
        """([FM blocks, table entries] of the first seeding pass, [FM blocks, table entries, launches, strand searches] of the second pass
        inside the chunk's sequence, [ms first pass, ms second pass]) since the last reset; read before counters()/seed_table() with reset"""
        w = (C.c_uint64 * 6)()
        ms = (C.c_double * 2)()
        B.check(B.lib().bsx_device_seed_passes(self.h, w, ms, int(reset)), "bsx_device_seed_passes")
        return list(w[:2]), list(w[2:]), list(ms)

    def region_work(self, reset=False):
        """[strand searches, SA intervals, occurrences looked up, regions written, read bases] of the region kernels since the last reset"""
        w = (C.c_uint64 * 5)()
        B.check(B.lib().bsx_device_region_work(self.h, w, int(reset)), "bsx_device_region_work")
        return list(w)

    def seed_table(self, reset=False):
        """(table entries read by the seeding kernel since the last reset, depth K of the resident table of k-mer intervals; 0 = none)"""
        n, k = C.c_uint64(), C.c_int()
        B.check(B.lib().bsx_device_seed_table(self.h, C.byref(n), C.byref(k), int(reset)), "bsx_device_seed_table")
        return int(n.value), int(k.value)

    def kernel_time(self, k, reset=False):
        ms = C.c_double()
        n = C.c_int64()
        B.check(B.lib().bsx_device_kernel_time(self.h, k, C.byref(ms), C.byref(n), int(reset)), "bsx_device_kernel_time")
        return ms.value, n.value

    def close(self):
        if self.h:
            B.lib().bsx_device_close(self.h)
            self.h = None
