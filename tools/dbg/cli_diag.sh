#!/bin/bash
# the command line on the hg38-like genome with the stream's per-chunk phase lines: where a chunk's time goes between FASTQ text and SAM text
# usage: tools/dbg/cli_diag.sh "" "BSX_TUNE=stream_whole_chunk=2" ...   (one run per configuration, same files)
cd /root/repo
i=0
for cfg in "$@"; do
	i=$((i + 1))
	env $cfg BSX_PHASES=1 E2E_STDERR=/root/repo/gpurun_out/cli_diag.err timeout 1500 python tools/cli_e2e.py --genome-mbp 3100 --profile 1 --chunks ${CHUNKS:-10} --out /dev/null --json > gpurun_out/cli_diag_$i.json 2> gpurun_out/cli_diag_$i.log
	echo "== $cfg" > gpurun_out/cli_diag_$i.txt
	grep "M::stream\|M::main\|lane [0-9] from\|M::regions\] regions_batch" gpurun_out/cli_diag.err | cut -c1-260 >> gpurun_out/cli_diag_$i.txt
	rm -f gpurun_out/cli_diag.err
done
