#!/usr/bin/env python3
"""Build an hg38-sized synthetic genome and both FM indices on the device; report times and memory."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from biscuit_amd.api import Index, Device
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 3_100_000_000
fill = len(sys.argv) > 2 and sys.argv[2] == "fill"
t0 = time.time()
idx = Index.synthetic(n, seed=2024, n_contigs=24)
t1 = time.time()
print("genome %d bp generated in %.1f s, maxrss %.1f GB" % (n, t1 - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
dev = Device(0)
os.environ["BSX_INDEX_TRACE"] = "1"
dev.build_index(idx, fill_host=fill)
t2 = time.time()
print("device index build %.1f s, maxrss %.1f GB" % (t2 - t1, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
import torch
free, tot = torch.cuda.mem_get_info(0)
print("HBM in use after build: %.1f GB of %.1f" % ((tot - free) / 1e9, tot / 1e9))
