#!/bin/bash
# a sampling profile of the host side of the pipelined bench run (hg38-like genome), symbolized on the box
cd /root/repo
BSX_PROF_SAMPLE=/root/repo/gpurun_out/hp_samples.txt timeout 1500 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --sub > gpurun_out/hp_bench.json 2> gpurun_out/hp_bench.err
python3 tools/prof_symbols.py gpurun_out/hp_samples.txt 70 libc > gpurun_out/hp_funcs.txt 2>&1
rm -f gpurun_out/hp_samples.txt
