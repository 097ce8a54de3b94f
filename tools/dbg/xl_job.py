# debugging: a chain's left then right extension as plain jobs (exact h0 chaining) through k_ext4 / k_extl and the CPU restatement
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import simdata, oracle_lib
from biscuit_amd.api import Index, Device, default_opt, EXT_DT
from biscuit_amd import _lib as B
d = "/tmp/xlchk"
contigs = simdata.make_genome(1000000, seed=21, n_contigs=3)
p100 = simdata.make_pairs(contigs, 5000, 100, 1, frag=(180, 320), sub=0.005)
idx = Index(d + "/g")
dev = Device(0); dev.upload_index(idx)
port = oracle_lib.Port(idx, n_threads=2)
opt = default_opt()
cases = [(641, 121184, 4, 22), (39, 668778, 19, 19), (983, 1144765, 18, 21), (749, 1803203, 30, 19), (935, 469021, 6, 22), (755, 993892, 11, 19), (439, 625456, 1, 27)]
seqs = []
for t, rbeg, qb, ln in cases:
    n, a, b = p100[(t // 2) // 2]
    seqs.append(a if (t // 2) % 2 == 0 else b)
buf, offs = simdata.read_buffer(seqs)
for be in (port, dev):
    be.set_opt(opt); be.set_reads(buf)
def gap(q): return min(max(int((q * 1 - 6) / 1 + 1.), int((q * 1 - 6) / 1 + 1.), 1), 200)
for par in (0, 1):
    left = np.array([(rbeg - 1, offs[k] + qb - 1, qb, qb + gap(qb), ln, 100, 10, -1, -1, par, 0) for k, (t, rbeg, qb, ln) in enumerate(cases)], dtype=EXT_DT)
    pl = port.extend(left)
    right = np.array([(rbeg + ln, offs[k] + qb + ln, 100 - qb - ln, 100 - qb - ln + gap(100 - qb - ln), int(pl[k]["score"]), 100, 10, 1, 1, par, 0) for k, (t, rbeg, qb, ln) in enumerate(cases)], dtype=EXT_DT)
    pr = port.extend(right)
    for mode in ("1", "2"):
        B.tune("ext4", mode)
        dl, dr = dev.extend(left), dev.extend(right)
        print("parent", par, "mode", mode, "left bad", int((pl != dl).sum()), "right bad", int((pr != dr).sum()))
        for i in np.nonzero(pr != dr)[0][:4]:
            print(right[i], "cpu", pr[i], "dev", dr[i])
    print([int(x) for x in pl["score"]], [int(x) for x in pr["score"]])
