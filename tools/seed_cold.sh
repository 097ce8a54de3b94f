#!/bin/bash
# k_seed: how often the full state machine runs (every N-th trip / when more than M lanes wait for it)
for cfg in "4 24" "8 24" "8 32" "16 32" "2 16" "4 40" "4 24"; do
  set -- $cfg
  BSX_SEED_COLD_EVERY=$1 BSX_SEED_COLD_LANES=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-hard-genome 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('every $1 lanes $2: seed ms', d['kernel_ms_per_step']['seed'], 'blocks/read', d['roofline']['fm_block_touches_per_read'])"
done
