#!/usr/bin/env python3
"""Who runs alone?  From a rocprofv3 --kernel-trace CSV of the pipelined bench: per kernel, its summed duration, the part of it during which no
OTHER kernel was executing ("alone"), and the part during which a kernel of another chunk's front half was executing beside it.
   python tools/overlap_trace.py <kernel_trace.csv> [t0_ms t1_ms]     (a window of the trace, relative to its first kernel)"""
import collections
import csv
import sys


def short(k):
    k = k.split("(")[0].replace("void ", "")
    return k[:48]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
t00 = rows[0][0]
if len(sys.argv) > 3:
    lo, hi = t00 + float(sys.argv[2]) * 1e6, t00 + float(sys.argv[3]) * 1e6
    rows = [(max(a, lo), min(b, hi), k) for a, b, k in rows if b > lo and a < hi]
# sweep: events
ev = []
for i, (a, b, k) in enumerate(rows):
    ev.append((a, 1, i))
    ev.append((b, 0, i))
ev.sort()
active = set()
tot = collections.defaultdict(float)
alone = collections.defaultdict(float)
busy = 0.0
conc = collections.defaultdict(float)   # time with n kernels active
last = ev[0][0]
for t, kind, i in ev:
    dt = (t - last) * 1e-6
    if dt > 0 and active:
        busy += dt
        conc[min(len(active), 6)] += dt
        names = {rows[j][2] for j in active}
        for j in active:
            tot[rows[j][2]] += dt
        if len(names) == 1:
            alone[next(iter(names))] += dt * 1.0
    last = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
span = (ev[-1][0] - ev[0][0]) * 1e-6
print("window %.0f ms, a kernel executing %.0f ms (%.1f %%)" % (span, busy, 100 * busy / span))
print("time with n dispatches executing: " + "  ".join("%d%s: %.0f ms" % (n, "+" if n == 6 else "", conc[n]) for n in sorted(conc)))
print("%-50s %10s %10s %8s" % ("kernel", "summed ms", "alone ms", "alone %"))
for k in sorted(tot, key=lambda k: -tot[k]):
    if tot[k] >= 1.0:
        print("%-50s %10.1f %10.1f %7.1f%%" % (k, tot[k], alone[k], 100 * alone[k] / tot[k]))
