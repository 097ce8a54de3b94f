#!/bin/bash
# Collect the rocprofv3 evidence committed under profiles/: one kernel-trace run of the default bench command and
# separate --pmc passes (never combined with sys/hip traces).  Run on the GPU box from the repo root:
#   tools/profile_round.sh r05            (GENOME_PROFILE=hg38-like, the headline genome; GENOME_PROFILE=clean for the other)
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
GP=${GENOME_PROFILE:-hg38-like}
BENCH="python $ROOT/bench.py --genome-profile $GP --genome-mbp ${GENOME_MBP:-3100} --steps ${STEPS:-12} --warmup 1 --no-cpu-baseline --sub"
cd /tmp
$BENCH > "$OUT/bench_untraced.json" 2> "$OUT/bench_untraced.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
	name=$(echo $grp | tr ' ' '_')
	rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o bench -- python $ROOT/bench.py --genome-profile $GP --genome-mbp ${GENOME_MBP:-3100} --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --sub > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err"
	# (the profiler may crash while the process exits, after the counters have been written: judge by the output)
	[ -n "$(find "$OUT/pmc_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)" ] || echo "pmc group $grp: no counter file" >> "$OUT/errors.txt"
done
find "$OUT" -name "*.csv" | head -50 > "$OUT/files.txt"
# gpurun brings back at most 64 MiB: the summaries are made here, the raw traces (tens of MB each) stay behind
cd "$ROOT"
PROFILES_OUT="$OUT/summary" GENOME_PROFILE=$GP python tools/summarize_profiles.py "$OUT" "$TAG" > "$OUT/summarize.log" 2>&1
find "$OUT" -name "*.csv" -size +1M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT" >> "$OUT/summarize.log"
