#!/usr/bin/env python3
"""End-to-end run of the command line (FASTQ parse -> align -> SAM text written), the figure SURVEY 8(d) asks for
next to the in-memory bench: python tools/cli_e2e.py [--genome-mbp 128] [--chunks 6]"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--genome-mbp", type=float, default=128)
ap.add_argument("--chunks", type=int, default=6)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--out", default=None, help="where the SAM goes (default: a file in the work directory; /dev/null isolates the aligner from the write)")
a = ap.parse_args()
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, Device
L = B.lib()
n_bases = int(a.genome_mbp * 1e6)
work = "/tmp/bsx_bench_%d" % n_bases
base = work + "/g"
t_build = None
if not os.path.exists(base + ".dau.sa"):
    os.makedirs(work, exist_ok=True)
    tb = time.time()
    if n_bases > 1_000_000_000:   # past the host builder's 32-bit suffix sorter: both indices on the device, then the seven files
        g = Index.synthetic(n_bases, seed=2024, n_contigs=24)
        dev = Device(0)
        dev.build_index(g, fill_host=True)
        g.save(base)
        dev.close(); g.close()
    else:
        B.check(L.bsx_sim_genome((work + "/g.fa").encode(), C.c_int64(n_bases), C.c_uint64(2024), 8, C.c_double(0.05)), "sim_genome")
        B.check(L.bsx_index_build((work + "/g.fa").encode(), base.encode()), "index_build")
    t_build = round(time.time() - tb, 1)
idx = Index(base)
L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
L.bsx_sim_write_fastq.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int]
L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
pairs = 10000000 * a.threads // 300
fq1, fq2 = work + "/e2e_1.fq", work + "/e2e_2.fq"
for k in range(a.chunks):
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, pairs, 150, 4000 + k, 200, 500, 0.005, 0.0, C.byref(p)), "sim_pairs")
    B.check(L.bsx_sim_write_fastq(p, 2 * pairs, fq1.encode(), fq2.encode(), 1 if k else 0), "write_fastq")
    L.bsx_sim_free_reads(p, 2 * pairs)
idx.close()
env = dict(os.environ, BSX_HOST_THREADS=str(a.threads))
t0 = time.time()
outp = a.out or (work + "/e2e.sam")
with open(outp, "wb") as out:
    p = subprocess.run([os.path.join(ROOT, "biscuit_amd", "biscuit_align"), "-@", str(a.threads), base, fq1, fq2], stdout=out, stderr=subprocess.PIPE, env=env)
dt = time.time() - t0
assert p.returncode == 0, p.stderr.decode()[-2000:]
if os.environ.get("E2E_STDERR"):
    open(os.environ["E2E_STDERR"], "wb").write(p.stderr)
n = 2 * pairs * a.chunks
print({"cli_end_to_end_reads_per_s": round(n / dt, 1), "reads": n, "seconds": round(dt, 2), "sam_bytes": os.path.getsize(outp), "out": outp,
       "genome_mbp": a.genome_mbp, "genome_and_index_files_s": t_build, "stderr_tail": p.stderr.decode()[-600:],
       "includes": "index load + upload, FASTQ parse, alignment, SAM text to a file"})
