#!/bin/bash
# Every device source compiled to gfx950 assembly (no GPU needed) and looked over by tools/dbg/exec_join_check.py: register copies placed
# ahead of the EXEC restore at the end of a divergent region (the round-5 miscompile).  Exit code 1 when a place is flagged.
# Usage: tools/isa_check.sh [extra hipcc flags]     (assembly kept under build/asm/)
set -u
cd "$(dirname "$0")/.."
mkdir -p build/asm
pids=()
for f in biscuit_amd/csrc/hip/*.hip; do
	b=$(basename $f .hip)
	if [ ! -s build/asm/$b.s ] || [ $f -nt build/asm/$b.s ] || [ -n "$(find biscuit_amd/csrc/hip -name '*.hpp' -newer build/asm/$b.s 2>/dev/null)" ] || [ -n "$(find biscuit_amd/csrc/hip -name '*.h' -newer build/asm/$b.s 2>/dev/null)" ]; then
		/opt/rocm/bin/hipcc -Wno-unused-command-line-argument --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ibiscuit_amd/csrc/host -Ibiscuit_amd/csrc/hip "$@" \
			-S --cuda-device-only $f -o build/asm/$b.s &
		pids+=($!)
	fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
python3 tools/dbg/exec_join_check.py build/asm/*.s
