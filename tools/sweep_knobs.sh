run() { env "$@" timeout 300 python bench.py --steps 12 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$*',d['value'],d['ms_per_step'])"; }
run A=1
run BSX_SEED_OCC=4
run BSX_REGIONS_OCC=3
run BSX_REGIONS_OCC=5
run BSX_SEED_QUOTA=4
run BSX_SEED_QUOTA=1
run BSX_REGIONS_QUOTA=32
run BSX_REGIONS_QUOTA=8
run BSX_RESERVE_CU_EVERY=16
run A=2
