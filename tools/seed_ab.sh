#!/bin/bash
# A/B of k_seed variants on one box: unpipelined chunks, stand-alone kernel times
run() { env "$@" timeout 300 python bench.py --steps 3 --no-pipeline --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['kernel_ms_per_step'];print('$*',d['value'],k['seed'],k['regions_tier1'],k['regions_tiers23'])"; }
for v in "$@"; do run $v; done
