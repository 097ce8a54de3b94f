#!/bin/bash
# Where the waves of the region kernels spend their cycles: separate rocprofv3 --pmc passes (one counter group per run, kernel trace only)
# of one stand-alone chunk, condensed on the box into gpurun_out/stalls_<tag>/<tag>_stalls.md (copy it to profiles/).
#   tools/stall_round.sh r05            (GENOME_PROFILE=hg38-like | clean)
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/stalls_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
GP=${GENOME_PROFILE:-hg38-like}
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LEVEL_WAVES"; do
	i=$((i + 1))
	rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o bench -- python $ROOT/bench.py --genome-profile $GP --genome-mbp ${GENOME_MBP:-3100} --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --sub ${BENCH_EXTRA:-} > "$OUT/g$i.json" 2> "$OUT/g$i.err"
done
cd "$ROOT"
python3 tools/summarize_stalls.py "$OUT" "$TAG" "$GP" > "$OUT/summarize.log" 2>&1
find "$OUT" -name "*.csv" -size +1M -delete
find "$OUT" -name "*.db" -delete
