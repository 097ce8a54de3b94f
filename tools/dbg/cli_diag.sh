#!/bin/bash
# the command line on the hg38-like genome with the stream's per-chunk phase lines: where a chunk's time goes between FASTQ text and SAM text
cd /root/repo
BSX_PHASES=1 E2E_STDERR=/root/repo/gpurun_out/cli_diag.err timeout 1500 python tools/cli_e2e.py --genome-mbp 3100 --profile 1 --chunks ${CHUNKS:-10} --out /dev/null --json > gpurun_out/cli_diag.json 2> gpurun_out/cli_diag.log
grep "M::stream\|Processed\|device de-dup\|M::main\|M::cli" gpurun_out/cli_diag.err | cut -c1-260 > gpurun_out/cli_diag.txt
rm -f gpurun_out/cli_diag.err
