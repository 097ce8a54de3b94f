#!/usr/bin/env python3
"""Stand-alone kernel times of one unpipelined chunk of the bench workload under different environments, the index built once:
   python tools/chunk_ab.py [--genome-mbp 3100] [--profile 0|1] "" "seed_direct=0" "tier1c=0 x4=0" ...   (settings of the library, csrc/host/tune.c; anything else is set as an environment variable)"""
import argparse
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3100)
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--single-end", action="store_true")
    ap.add_argument("--crc", action="store_true", help="checksum of the chunk's SAM text per configuration (the knobs must not change it)")
    ap.add_argument("cfgs", nargs="*", default=[""])
    a = ap.parse_args()
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    L = B.lib()
    n_bases = int(a.genome_mbp * 1e6)
    idx = Index.synthetic(n_bases, seed=2024, n_contigs=24 if n_bases >= 1_000_000_000 else 8, profile=a.profile)
    dev = Device(0)
    dev.build_index(idx)
    opt = default_opt()
    opt.n_threads = 16
    opt.flag |= 0x10 | (0 if a.single_end else 0x2)
    pairs = (opt.chunk_size * 16) // (2 * a.read_len)
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    C.c_int.in_dll(L, "bsx_verbose").value = 1
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, pairs, a.read_len, 1001, a.read_len if a.single_end else 200, a.read_len + 400 if a.single_end else 500, 0.005, 0.0, C.byref(p)), "sim")
    B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * pairs, p, None), "warm")
    L.bsx_sim_reset_reads(p, 2 * pairs)
    names = ["seed", "occ", "extend", "sw", "global", "tier1", "tiers23", "seed2"]
    import time
    for cfg in a.cfgs:
        kv = dict(x.split("=", 1) for x in cfg.split())
        tnames = set(B.tune_names())
        for k, v in kv.items():   # a setting of the library (name or the BSX_NAME it was as an environment variable), else a real environment variable
            low = k[4:].lower() if k.startswith("BSX_") else k.lower()
            if low in tnames:
                B.tune(low, v)
            else:
                os.environ[k] = v
        for k in range(8):
            dev.kernel_time(k, reset=True)
        t0 = time.time()
        crc = None
        for _ in range(a.reps):
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * pairs, p, None), "chunk")
            if a.crc and crc is None:
                import zlib
                reads = C.cast(p, C.POINTER(B.Read))
                crc = 0
                for i in range(2 * pairs):
                    crc = zlib.crc32(C.string_at(reads[i].sam), crc)
            L.bsx_sim_reset_reads(p, 2 * pairs)
        dt = (time.time() - t0) / a.reps
        print("%-50s %s | chunk %.0f ms" % (cfg or "(defaults)", " ".join("%s %.1f" % (names[k], dev.kernel_time(k)[0] / a.reps) for k in range(8)), dt * 1e3) + (" | sam crc %08x" % crc if crc is not None else ""), flush=True)
        for k in kv:
            low = k[4:].lower() if k.startswith("BSX_") else k.lower()
            if low in tnames:
                B.tune(low, None)
            else:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
