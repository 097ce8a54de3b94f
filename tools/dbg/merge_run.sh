#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_hg38_like_scale.py -x -q -m gpu -k "repeat_rich or hg38 or stream or long_1kb" 2>&1 | tail -5
STEPS=10 tools/hg_ab.sh "BSX_REDO_MERGE_MIN=1000000000" "" "BSX_REDO_MERGE_MIN=1000000000" "" 2>&1
CHUNKS=12 tools/dbg/cli_diag.sh ""
