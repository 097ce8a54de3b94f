#!/usr/bin/env python3
"""Device path vs host chaining / host de-duplication on one bench-shaped chunk of 1 066 666 reads (tandem-repeat reads included), SAM compared
by checksum: python tools/parity_big.py [genome_mbp] [seed] [families] [port]   (>= 1000 Mbp: synthetic genome indexed on the device;
"port": a fifth run of the same chunk through the CPU restatement of the kernels, oracle/port.c -- minutes of host time)"""
import ctypes as C, os, sys, tempfile, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, Device, default_opt
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fam = int(sys.argv[3]) if len(sys.argv) > 3 else 6
with_port = len(sys.argv) > 4 and sys.argv[4] == "port"
L = B.lib()
if mbp >= 1000:
    idx = Index.synthetic(int(mbp * 1e6), seed=2024, n_contigs=24); dev = Device(0); dev.build_index(idx, fill_host=with_port)
else:
    d = tempfile.mkdtemp()
    B.check(L.bsx_sim_genome((d + "/g.fa").encode(), C.c_int64(int(mbp * 1e6)), C.c_uint64(seed), fam, C.c_double(0.05)), "g")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "ib")
    idx = Index(d + "/g"); dev = Device(0); dev.upload_index(idx)
opt = default_opt(); opt.n_threads = 16; opt.flag |= 0x10 | 0x2
L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
n_pairs = 533333
p = C.c_void_p()
B.check(L.bsx_sim_pairs(idx.h, n_pairs, 150, 1001, 200, 500, 0.005, 0.0, C.byref(p)), "sp")
reads = C.cast(p, C.POINTER(B.Read))
def crc():
    c = 0
    for i in range(2 * n_pairs): c = zlib.crc32(C.string_at(reads[i].sam), c)
    return c
out = []
for host, asy, hdd in ((0, 1, 0), (0, 0, 0), (0, 0, 1), (1, 0, 0)):
    B.tune("host_chain", "1" if host else None)
    B.tune("host_dedup", "1" if hdd else None)
    B.tune("async_redo", str(asy))
    t = time.time()
    B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "ps")
    dt = time.time() - t
    ps = B.PhaseStats(); L.bsx_last_phase_stats(C.byref(ps))
    out.append(crc())
    print("host_chain", host, "host_dedup", hdd, "async", asy, "tasks", ps.n_tasks, "host tasks", ps.n_host_tasks, "second pass", ps.n_redo_tasks, "crc %08x" % out[-1], "%.2f s" % dt, flush=True)
    L.bsx_sim_reset_reads(p, 2 * n_pairs)
if with_port:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from oracle_lib import Port
    port = Port(idx, 16)
    be = port.backend()
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    t = time.time()
    B.check(L.bsx_process_seqs_backend(C.byref(be), C.byref(opt), idx.h, 0, 2 * n_pairs, p, None), "port")
    out.append(crc())
    print("CPU restatement of the kernels (oracle/port.c), host chaining: crc %08x" % out[-1], "%.1f s" % (time.time() - t), flush=True)
print("PARITY", "OK" if len(set(out)) == 1 else "MISMATCH")
