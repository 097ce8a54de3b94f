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
out = os.environ.get("PROFILES_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")   # ($PROFILES_OUT: on the GPU box, into gpurun_out/)
os.makedirs(out, exist_ok=True)
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
         "    rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -d <dir> -o bench -- python bench.py --genome-mbp $GENOME_MBP --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline\n",
         "Workload: one chunk = 1,066,666 reads (2x150 bp) vs the synthetic genome of $GENOME_MBP Mbp (default 3100: hg38-sized), chunks one at a time (no overlap);",
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

# --- HBM traffic per kernel (feeds bench.py's roofline.traffic) and the device's busy time under the pipelined run
import json


def pmc_dir(name):
    f = glob.glob(os.path.join(src, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True)
    return table(f[0])[0] if f else None


def pick(agg, kernel):
    tot = collections.defaultdict(float)
    for k, v in agg.items():
        if k.startswith(kernel + "<") or k == kernel:
            for c, x in v.items():
                tot[c] += x
    return tot


fs, ws, hm = pmc_dir("FETCH_SIZE"), pmc_dir("WRITE_SIZE"), pmc_dir("TCC_HIT_sum_TCC_MISS_sum")
ins = pmc_dir("SQ_WAVES_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS")
mbp = int(round(float(os.environ.get("GENOME_MBP", "3100"))))
gprof = os.environ.get("GENOME_PROFILE", "hg38-like")   # the profile the passes ran on (tools/profile_round.sh)
if fs and ws:
    traffic = {}
    for kern in ("k_seed", "k_occ", "k_regions"):
        # k_regions: the whole family -- every tier (k_regions<>, k_regions_mid, k_regions_slab<>) and the chains -> regions launch (k_c2r)
        def pk(agg, kern=kern):
            tot = collections.defaultdict(float)
            for k, v in agg.items():
                if k == kern or k.startswith(kern + "<") or (kern == "k_seed" and (k.startswith("k_seedt<") or k.startswith("k_seedt_pack"))) or (kern == "k_regions" and k.startswith(("k_regions", "k_c2r", "k_ext_", "k_ext4", "k_extl", "k_x4prep"))):
                    for c, x in v.items():
                        tot[c] += x
            return tot
        traffic[kern] = {"FETCH_SIZE_KiB": pk(fs).get("FETCH_SIZE", 0.0), "WRITE_SIZE_KiB": pk(ws).get("WRITE_SIZE", 0.0)}
        if hm:
            traffic[kern]["TCC_HIT"] = pk(hm).get("TCC_HIT_sum", 0.0)
            traffic[kern]["TCC_MISS"] = pk(hm).get("TCC_MISS_sum", 0.0)
        if ins:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
                traffic[kern][c] = pk(ins).get(c, 0.0)
    traffic["_note"] = ("rocprofv3 --pmc passes (one counter group per run, with --kernel-trace only) of `python bench.py --genome-profile " + gprof + " --genome-mbp %d --steps 1 --warmup 0 "
                        "--no-cpu-baseline --no-pipeline --sub` (one chunk of 1,066,666 reads; tools/profile_round.sh), summed over the dispatches of each kernel "
                        "family.  FETCH_SIZE calibration on this access pattern (r01): k_occ reads exactly one 64-byte FM block per LF step; its FETCH_SIZE "
                        "was 0.88 x that byte count at a 7 %% L2 hit rate, i.e. for 64-byte gathers FETCH_SIZE is within a few percent of the bytes that "
                        "miss L2 (not the 1/2 the guide measured for wide coalesced streams)." % mbp)
    traffic["_reads_per_chunk"] = 1066666
    json.dump(traffic, open(os.path.join(out, "%s_pmc_%dmbp_%s.json" % (tag, mbp, "hg38like" if gprof == "hg38-like" else "clean")), "w"), indent=1)

kt = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
if kt:
    iv = []
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(kt[0])):
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        iv.append((a, b))
        per[short(r["Kernel_Name"])] += (b - a) * 1e-6
    iv.sort()
    # the trace covers index upload, warm-up, the timed steps and the stand-alone chunk: split it at idle gaps > 0.5 s
    # (host-only stretches: synthetic reads being generated) and report every stretch; the longest is the timed region
    merged = []
    for a, b in iv:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    phases = [[merged[0]]]
    for m in merged[1:]:
        if m[0] - phases[-1][-1][1] > 500e6:
            phases.append([m])
        else:
            phases[-1].append(m)
    with open(os.path.join(out, "%s_gpu_busy.md" % tag), "w") as g:
        g.write("# Device busy time under the pipelined bench (%s)\n\n" % tag)
        g.write("From the rocprofv3 kernel trace of the default `bench.py` run (tools/profile_round.sh): stretches of device activity\n"
                "separated by idle gaps > 0.5 s.  The longest stretch is warm-up + timed region (chunks through the depth-3 stream);\n"
                "`busy` = at least one kernel executing, which includes single-wave launches (the second seeding pass), so it says\n"
                "when the device had nothing at all to do, not how full it was.\n\n")
        g.write("| stretch | start, ms | length, ms | busy, ms | busy % | kernels |\n|---|---|---|---|---|---|\n")
        t0 = iv[0][0]
        for i, ph in enumerate(phases):
            a, b = ph[0][0], ph[-1][1]
            busy = sum(y - x for x, y in ph)
            nk = sum(1 for x, y in iv if a <= x <= b)
            g.write("| %d | %.0f | %.1f | %.1f | %.1f | %d |\n" % (i, (a - t0) * 1e-6, (b - a) * 1e-6, busy * 1e-6, 100.0 * busy / max(1, b - a), nk))
        g.write("\n| kernel | summed duration over the whole trace, ms |\n|---|---|\n")
        for k in sorted(per, key=lambda k: -per[k]):
            if per[k] >= 1.0:
                g.write("| %s | %.1f |\n" % (k, per[k]))

# --- the dominant kernel's launches by kind: rocprofv3's per-kernel average mixes the chunk-wide launch (what bench.py's
# `roofline` times with HIP events) with the single-workgroup launches of the second seeding pass and of the host path
if kt:
    cls = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        nm = short(r["Kernel_Name"])
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        if nm.startswith("k_seedt<"):   # the table form: the chunk-wide launch is persistent waves (thousands of one-wave workgroups), the second pass a few dozen
            wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
            cls["k_seedt, chunk-wide (>= 1000 workgroups): what bench.py's roofline times" if wgs >= 1000 else "k_seedt, second pass / host path (small grids)"].append(ms)
        elif nm.startswith("k_seedt_pack"):
            cls["k_seedt_pack (the reads as base-3 digits, ahead of every k_seedt launch)"].append(ms)
        elif nm.startswith("k_seed<"):
            cls["k_seed without the table (one chunk after the timed region: counts the reference's FM-block touches)"].append(ms)
    with open(os.path.join(out, "%s_kseed_launches.md" % tag), "w") as g:
        g.write("# k_seed launches in the kernel trace (%s), by kind\n\n" % tag)
        g.write("`%s_kernel_stats.csv` averages every launch of a kernel; bench.py's `roofline.avg_launch_ms` is the chunk-wide launch only\n"
                "(HIP events on its stream), so compare it with the first row.  The trace covers warm-up chunk + timed chunks (pipelined:\n"
                "kernels of two or three chunks share the device) + one stand-alone chunk (`avg_launch_ms_standalone`).\n\n" % tag)
        g.write("| kind | launches | average ms | min ms | max ms |\n|---|---|---|---|---|\n")
        for k, v in sorted(cls.items()):
            g.write("| %s | %d | %.1f | %.1f | %.1f |\n" % (k, len(v), sum(v) / len(v), min(v), max(v)))

# --- each kernel family's durations in the pipelined trace, per chunk: what bench.py quotes beside its HIP-event spans
# (`kernel_ms_sum_from_trace`).  Chunks in the trace = launches of k_dedup (one per chunk: warm-up, timed, stand-alone and the classic-seeding chunk).
if kt:
    fam = {"seeding": ("k_seedt<", "k_seedt_pack"), "sa_lookup": ("k_occ",),
           "region_family": ("k_regions", "k_c2r", "k_ext4", "k_extl", "k_x4prep", "k_seedsw", "k_swl16"),
           "mate_rescue": ("k_swl<", "k_sw<"), "cigar": ("k_global",), "dedup": ("k_dedup",)}
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    chunks = 0
    for r in csv.DictReader(open(kt[0])):
        nm = short(r["Kernel_Name"])
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        if nm == "k_dedup":   # one launch per chunk (k_occ also runs for the few strand searches seeded again on the side stream)
            chunks += 1
        for f_, pre in fam.items():
            if any(nm == p_ or nm.startswith(p_) for p_ in pre) and not (f_ == "sa_lookup" and not nm.startswith("k_occ")):
                tot[f_][nm[:60]] += ms
                calls[nm[:60]] += 1
                break
    if chunks:
        fj = {}
        for f_, d_ in tot.items():
            fj[f_] = {"chunks": chunks, "pipelined_ms_per_chunk": round(sum(d_.values()) / chunks, 2),
                      "by_kernel": {k: {"ms_per_chunk": round(v / chunks, 2), "launches": calls[k]} for k, v in sorted(d_.items(), key=lambda kv: -kv[1])}}
        fj["_note"] = ("rocprofv3 --kernel-trace of the default pipelined bench command (tools/profile_round.sh): kernel durations summed per family and divided by the "
                       "chunks in the trace (k_dedup launches: warm-up + timed + the stand-alone chunk + the classic-seeding chunk, which adds no k_seedt time).")
        json.dump(fj, open(os.path.join(out, "%s_family_ms_%dmbp_%s.json" % (tag, mbp, "hg38like" if gprof == "hg38-like" else "clean")), "w"), indent=1)
