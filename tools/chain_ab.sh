#!/bin/bash
# how the front halves of consecutive chunks share the device: $BSX_CHAIN_STAGES 2 (stage by stage, default) against 4 (one front half after the other) and 0
for c in 2 4 0 2 4; do
  BSX_CHAIN_STAGES=$c timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-hard-genome 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['kernel_ms_per_step'];print('chain $c', d['value'], d['ms_per_step'], k)"
done
