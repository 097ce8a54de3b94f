#!/bin/bash
# k_seed duration on this box under a few launch shapes (one chunk each, unpipelined)
for cfg in "BSX_SEED_QUOTA=2" "BSX_SEED_QUOTA=0 BSX_SEED_WAVES_PER_CU=12" "BSX_SEED_QUOTA=0 BSX_SEED_WAVES_PER_CU=8" "BSX_SEED_QUOTA=0 BSX_SEED_WAVES_PER_CU=4" "BSX_SEED_QUOTA=0 BSX_SEED_WAVES_PER_CU=2"; do
  echo -n "$cfg: "
  env $cfg python bench.py --genome-mbp 128 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step']['seed'])"
done
