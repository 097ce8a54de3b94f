#!/bin/bash
cd /root/repo
BSX_PHASES=1 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --sub > gpurun_out/bench_phase.json 2> gpurun_out/bench_phase.err
grep "M::stream" gpurun_out/bench_phase.err | cut -c1-220 > gpurun_out/bench_phase.txt
rm -f gpurun_out/bench_phase.err
CHUNKS=12 tools/dbg/cli_diag.sh "BSX_HOST_THREADS=24"
