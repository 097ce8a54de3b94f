#!/bin/bash
# pipelined default bench under a list of environment settings, one line each (A/B on one box): tools/sweep_knobs.sh A=1 BSX_X=2 ...
run() { env "$@" timeout 300 python bench.py --steps 12 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['kernel_ms_per_step_standalone'];print('$*',d['value'],d['ms_per_step'],k['regions_tiers23'],d['kernel_ms_per_step']['regions_tiers23'])"; }
for v in "$@"; do run $v; done
