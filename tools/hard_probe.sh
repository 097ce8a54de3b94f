# seeding-overflow sensitivity on the hg38-like genome: BSX_SEED_TRIP_BUDGET / BSX_SEED_MEM_CAP
for cfg in "4096 0" "0 0" "32768 0" "32768 400"; do
  set -- $cfg
  if [ "$2" = "0" ]; then unset BSX_SEED_MEM_CAP; else export BSX_SEED_MEM_CAP=$2; fi
  BSX_SEED_TRIP_BUDGET=$1 BSX_PHASES=1 timeout 600 python bench.py --genome-profile hg38-like --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline > gpurun_out/hp_$1_$2.json 2> gpurun_out/hp_$1_$2.err
  echo "budget=$1 memcap=$2: $(grep -E 'M::regions\] on device' gpurun_out/hp_$1_$2.err | tail -1)"
  echo "   $(grep -E 'seed kernel done' gpurun_out/hp_$1_$2.err | tail -1)"
  python -c "
import json; d=json.load(open('gpurun_out/hp_$1_$2.json')); print('   ', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
