#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_align.py -x -q -m gpu -k "long or seed_sw or seed_filter" 2>&1 | tail -5
timeout 900 python tools/chunk_ab.py --profile 0 --read-len 1000 --single-end --crc --reps 1 "BSX_PHASES=1" 2>&1 | grep "region launches\|seed filter\|^BSX\|^ " | tail -5
export TMPDIR=/tmp
O=/root/repo/gpurun_out/ssw_prof; mkdir -p $O
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python /root/repo/tools/chunk_ab.py --profile 0 --read-len 1000 --single-end --reps 1 --crc "" > $O/out.txt 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
head -1 $f > $O/trace_sel.csv
grep "k_seedsw\|k_swl16\|k_c2r\|k_regions" $f >> $O/trace_sel.csv
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
