#!/usr/bin/env python3
"""tools/summarize_stalls.py <dir of tools/stall_round.sh> <tag> <genome profile>: per kernel, where the wave cycles go (SQ counters summed over the
dispatches of one stand-alone chunk)."""
import collections
import csv
import glob
import os
import sys

src, tag, gp = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(src, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        k = k.split("(")[0][:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, c in agg.items():
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0 or k.startswith("__amd") or k.startswith("k_ix") or "rocprim" in k or k.startswith("k_bwt") or k.startswith("k_seedtab"):
        continue
    rows.append((wc, k, c))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
L = ["# Where the wave cycles go (%s; %s genome, one stand-alone chunk of 1,066,666 reads; MI355X, rocprofv3 --pmc, tools/stall_round.sh)\n" % (tag, gp),
     "SQ counters summed over every dispatch of a kernel.  `wave cycles`: cycles summed over the waves resident (SQ_WAVE_CYCLES; share = of the kernels",
     "listed).  `waiting`: SQ_WAIT_ANY / wave cycles (the wave is waiting for anything: memory, LDS, a barrier ...); `wait for issue`: SQ_WAIT_INST_ANY",
     "/ wave cycles (an instruction is ready and waits its turn); `LDS wait`: SQ_WAIT_INST_LDS / wave cycles; `VALU`, `SALU`, `LDS`, `VMEM`:",
     "SQ_ACTIVE_INST_* / wave cycles (cycles a wave spends executing that class); `bank conflicts`: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;",
     "`waves`: SQ_LEVEL_WAVES / SQ_BUSY_CYCLES-ish occupancy is not derived here (the counters are per-SE sums).\n",
     "| kernel | wave cycles | share | waiting | wait for issue | LDS wait | VALU | SALU | LDS | VMEM | any inst | LDS bank conflicts | VMEM insts | SMEM insts |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for wc, k, c in rows:
    def fr(n, d=wc):
        return "%.2f" % (c.get(n, 0) / d) if d > 0 else "-"
    L.append("| %s | %.3g | %.1f %% | %s | %s | %s | %s | %s | %s | %s | %s | %s | %.3g | %.3g |" % (
        k, wc, 100 * wc / tot, fr("SQ_WAIT_ANY"), fr("SQ_WAIT_INST_ANY"), fr("SQ_WAIT_INST_LDS"), fr("SQ_ACTIVE_INST_VALU"), fr("SQ_ACTIVE_INST_SCA"),
        fr("SQ_ACTIVE_INST_LDS"), fr("SQ_ACTIVE_INST_VMEM"), fr("SQ_ACTIVE_INST_ANY"),
        fr("SQ_LDS_BANK_CONFLICT", c.get("SQ_LDS_IDX_ACTIVE", 0)), c.get("SQ_INSTS_VMEM", 0), c.get("SQ_INSTS_SMEM", 0)))
open(os.path.join(src, "%s_stalls.md" % tag), "w").write("\n".join(L) + "\n")
print("wrote", os.path.join(src, "%s_stalls.md" % tag))
