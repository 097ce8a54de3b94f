#!/bin/bash
# Debug builds of libbiscuit_amd.so (never the product's): k_regions.hip compiled under -DRG_DBG=<n> (1: guard words around the LDS objects of
# the chaining tiers, 2: their tables filled with 0xff before every strand search -- see k_regions.hip) and shim.hip with the check of the
# exported chain records (-DBSX_DEBUG_XCHECK), one per directory tests/_build/lds_dbg_<n>/.  A command line runs against one of them with
# LD_LIBRARY_PATH=tests/_build/lds_dbg_<n> (biscuit_align's RUNPATH comes after it).  Needs the product's objects under build/ (make all).
# The compiler is the Makefile's ($HIPCC, else $ROCM/bin/hipcc); every compile is waited for by pid, so a failed one ends the script there.
cd "$(dirname "$0")/../.."
HIPCC=${HIPCC:-${ROCM:-/opt/rocm}/bin/hipcc}
F="--offload-arch=gfx950 -O3 -gline-tables-only -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Ibiscuit_amd/csrc/host -Ibiscuit_amd/csrc/hip"
pids=()
for v in "$@"; do
	d=tests/_build/lds_dbg_$v; mkdir -p $d
	$HIPCC $F -DRG_DBG=$v -c biscuit_amd/csrc/hip/k_regions.hip -o $d/hip_k_regions.o & pids+=($!)
	$HIPCC $F -DRG_DBG=$v -DBSX_DEBUG_XCHECK -c biscuit_amd/csrc/hip/shim.hip -o $d/hip_shim.o & pids+=($!)
done
for p in "${pids[@]}"; do
	wait $p || { echo "lds_variants: a compile failed" >&2; exit 1; }
done
for v in "$@"; do
	d=tests/_build/lds_dbg_$v
	objs=$(ls build/host_*.o build/hip_*.o | grep -v "hip_k_regions.o\|hip_shim.o")
	$HIPCC --offload-arch=gfx950 -shared -o $d/libbiscuit_amd.so $objs $d/hip_k_regions.o $d/hip_shim.o -lz -lm -lpthread || exit 1
	echo built $d
done
