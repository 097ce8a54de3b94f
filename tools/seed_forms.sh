#!/bin/bash
# k_seed forms side by side on one box (unpipelined chunks, stand-alone kernel time), checked for identical work counters
for f in 1 0 2 1; do
  BSX_SEED_FORM=$f timeout 400 python bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-hard-genome 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['kernel_ms_per_step'];r=d['roofline'];print('form $f', d['value'], 'seed ms', k['seed'], 'blocks/read', r['fm_block_touches_per_read'], 'achieved', r['achieved'])"
done
