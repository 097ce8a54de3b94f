#!/bin/bash
# sampling profile of the command line's host side (hg38-like genome) and the cgroup's CPU throttling over the run: tools/dbg/cli_prof.sh
cd /root/repo
rm -f gpurun_out/cli_samples.*.txt
cat /sys/fs/cgroup/cpu.stat > gpurun_out/cli_cpustat0.txt 2>/dev/null
BSX_PROF_SAMPLE=/root/repo/gpurun_out/cli_samples.%d.txt timeout 1500 python tools/cli_e2e.py --genome-mbp 3100 --profile 1 --chunks ${CHUNKS:-12} --out /dev/null --json > gpurun_out/cli_prof.json 2> gpurun_out/cli_prof.log
cat /sys/fs/cgroup/cpu.stat > gpurun_out/cli_cpustat1.txt 2>/dev/null
for f in gpurun_out/cli_samples.*.txt; do echo "== $f $(head -1 $f)"; python3 tools/prof_symbols.py $f 50 libc; done > gpurun_out/cli_funcs.txt 2>&1
rm -f gpurun_out/cli_samples.*.txt
# the same for a short in-library bench run
cat /sys/fs/cgroup/cpu.stat > gpurun_out/bench_cpustat0.txt 2>/dev/null
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sub > gpurun_out/bench_cpustat.json 2>/dev/null
cat /sys/fs/cgroup/cpu.stat > gpurun_out/bench_cpustat1.txt 2>/dev/null
