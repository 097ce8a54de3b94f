#!/bin/bash
# kernel trace of the default bench (16 pipelined steps) -> gpurun_out/busy/: how much of the steady state has a kernel executing
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/busy; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o bench -- python $ROOT/bench.py --steps ${STEPS:-12} --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/trace.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f)))
t0 = iv[0][0]
# 100 ms bins: busy fraction (union) and summed kernel time
end = iv[-1][1]
nb = int((end - t0) / 100e6) + 1
busy = [0.0] * nb; tot = [0.0] * nb
merged = []
for a, b, _ in iv:
    if merged and a <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], b)
    else: merged.append([a, b])
def spread(arr, a, b):
    i = int((a - t0) / 100e6)
    while a < b:
        e = min(b, t0 + (i + 1) * 100e6)
        arr[i] += e - a; a = e; i += 1
for a, b in merged: spread(busy, a, b)
for a, b, _ in iv: spread(tot, a, b)
print("bin(100ms) busy% concurrency")
for i in range(nb):
    if tot[i] > 0: print(i, round(busy[i] / 1e6), round(tot[i] / max(1.0, busy[i]), 2))
PY
