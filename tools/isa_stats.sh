#!/bin/bash
# Static look at a kernel's gfx950 code without a GPU: instruction counts by kind, scalar-register spills (v_writelane /
# v_readlane: "lane-moves", which also counts the kernel's own cross-lane reads), s_nop padding, and the same per loop nest, from the compiler's assembly.  What it is for: the region kernels are
# bound by scalar issue, so a change that removes spills or scalar control flow shows here before it is timed.
#   tools/isa_stats.sh biscuit_amd/csrc/hip/k_regions.hip _Z5k_c2r
#   tools/isa_stats.sh biscuit_amd/csrc/hip/k_seed.hip _Z6k_seedILi3
set -eu
SRC=$1; KERNEL=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=${TMPDIR:-/tmp}/isa_stats.$$
mkdir -p "$TMP"
/opt/rocm/bin/hipcc -Wno-unused-command-line-argument --offload-arch=gfx950 -O3 -std=c++17 -I"$ROOT/include" -I"$ROOT/biscuit_amd/csrc/host" -I"$ROOT/biscuit_amd/csrc/hip" \
	-S --cuda-device-only "$SRC" -o "$TMP/all.s"
awk -v k="$KERNEL" 'index($0,k)==1{f=1} f{print} f&&/s_endpgm/{exit}' "$TMP/all.s" > "$TMP/k.s"
[ -s "$TMP/k.s" ] || { echo "no kernel whose mangled name starts with $KERNEL" >&2; exit 1; }
mix() { grep -v '^\s*;' | grep -v '^\.' | awk '{ if ($1 ~ /^s_nop/) n++; else if ($1 ~ /^s_/) s++; else if ($1 ~ /^v_(write|read)lane/) sp++; else if ($1 ~ /^v_/) v++; else if ($1 ~ /^ds_/) d++; else if ($1 ~ /^(global|flat|buffer|scratch)_/) g++; else o++ } END {printf "salu %d valu %d lane-moves %d nop %d lds %d mem %d other %d\n", s, v, sp, n, d, g, o}'; }
echo "whole kernel: $(mix < "$TMP/k.s")"
grep -A80 "\.name: *$KERNEL" "$TMP/all.s" | grep -E "\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" | tr -s ' ' | tr '\n' ';'; echo
# loop headers in order, with the code from each header to the next one
grep -n "Loop Header: Depth=" "$TMP/k.s" | while IFS=: read ln rest; do
	depth=$(echo "$rest" | sed 's/.*Depth=\([0-9]*\).*/\1/')
	nxt=$(awk -v a="$ln" 'NR>a && /Loop Header: Depth=/{print NR; exit}' "$TMP/k.s"); [ -n "$nxt" ] || nxt=$(wc -l < "$TMP/k.s")
	printf "  line %5d depth %s, to the next header: %s\n" "$ln" "$depth" "$(awk -v a="$ln" -v b="$nxt" 'NR>=a && NR<b' "$TMP/k.s" | mix)"
done
rm -rf "$TMP"
