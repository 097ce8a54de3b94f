#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the files committed under profiles/.
usage: tools/summarize_profiles.py gpurun_out/prof_<tag> [gpurun_out/pmc2] <tag>"""
import collections
import csv
import glob
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[-1]
extra = sys.argv[2] if len(sys.argv) > 3 else None
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
for name in ("bench_kernel_stats.csv", "bench_domain_stats.csv"):
    p = os.path.join(src, "trace", name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, "%s_%s" % (tag, name.replace("bench_", ""))))
for name in ("bench_untraced.json", "bench_under_trace.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as f, open(os.path.join(out, "%s_%s" % (tag, name)), "w") as g:
            g.write(f.read().strip().split("\n")[-1] + "\n")


def short(k):
    k = k.split("(")[0]
    return k.replace("void ", "")[:70]


def table(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            disp[k] += 1
    return agg, disp


lines = ["# rocprofv3 PMC summary (%s; MI355X, gfx950, ROCm 7.2)\n" % tag,
         "One counter group per run, never combined with sys/hip traces (tools/profile_round.sh, tools/pmc_seed.sh):\n",
         "    rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -d <dir> -o bench -- python bench.py --genome-mbp 128 --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline\n",
         "Workload: one chunk = 1,066,666 reads (2x150 bp) vs the synthetic 128 Mbp genome, chunks one at a time (no overlap);",
         "sums over all dispatches of a kernel in that run.  FETCH_SIZE / WRITE_SIZE are KiB as reported; the MI355X guide notes that",
         "FETCH_SIZE under-reports wide coalesced reads 2x on gfx950 and is uncalibrated for other patterns (these kernels gather 64-byte",
         "blocks, 16 B per load), so the JSON's `traffic` stays null.\n"]
dirs = sorted(glob.glob(os.path.join(src, "pmc_*/")))
if extra:
    dirs += sorted(glob.glob(os.path.join(extra, "*/")))
for d in dirs:
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        continue
    agg, disp = table(f[0])
    names = sorted({c for v in agg.values() for c in v})
    lines.append("## " + " / ".join(names) + "\n")
    lines.append("| kernel | dispatches | " + " | ".join(names) + " |")
    lines.append("|---|---|" + "---|" * len(names))
    for k in sorted(agg):
        if k.startswith("__amd"):
            continue
        lines.append("| %s | %d | %s |" % (k, disp[k], " | ".join("%.4g" % agg[k].get(c, 0) for c in names)))
    lines.append("")
open(os.path.join(out, "%s_pmc_summary.md" % tag), "w").write("\n".join(lines))
print("wrote profiles/%s_*" % tag)
