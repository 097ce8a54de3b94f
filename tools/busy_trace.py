#!/usr/bin/env python3
"""Device busy fraction and the idle gaps of the steady state from a rocprofv3 kernel trace of bench.py:
python tools/busy_trace.py <bench_kernel_trace.csv>   (busy = at least one kernel or copy executing)"""
import csv
import sys
import collections
rows = []
with open(sys.argv[1]) as fh:
    for x in csv.DictReader(fh):
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].replace("void ", "")[:22]))
rows.sort()
# the timed region = the longest run of activity without a pause of 0.6 s
runs, cur = [], [rows[0]]
end = rows[0][1]
for r in rows[1:]:
    if r[0] - end > 6e8:
        runs.append(cur)
        cur = []
    cur.append(r)
    end = max(end, r[1])
runs.append(cur)
run = max(runs, key=lambda c: max(r[1] for r in c) - c[0][0])
t0, t1 = run[0][0], max(r[1] for r in run)
lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)      # the middle: no pipeline fill or drain
busy, end, gaps = 0, lo, collections.Counter()
for s, e, n in run:
    if e <= lo or s >= hi:
        continue
    s, e = max(s, lo), min(e, hi)
    if s > end:
        gaps[n] += s - end
        end = s
    if e > end:
        busy += e - end
        end = e
print("stretch %.2f s; middle 60 %%: busy %.1f %%" % ((t1 - t0) / 1e9, 100.0 * busy / (hi - lo)))
print("idle time by the kernel that ended it (ms):", {k: round(v / 1e6) for k, v in gaps.most_common(8)})
