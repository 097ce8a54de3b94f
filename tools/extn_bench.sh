set -u
for cfg in "1 19" "3 19" "0 19" "1 16"; do
  set -- $cfg
  BSX_EXTN_COLD=$1 BSX_EXTN_WPC=$2 BSX_C2R_LANES=1 BSX_PHASES=1 python bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline > gpurun_out/n2_$1_$2.json 2> gpurun_out/n2_$1_$2.err
  echo "cold=$1 wpc=$2: $(grep -E 'c2r_lanes\] k_ext_n' gpurun_out/n2_$1_$2.err | tail -1 | cut -c1-200)"
  python -c "
import json; d=json.load(open('gpurun_out/n2_$1_$2.json')); print('   tiers23', d['kernel_ms_per_step']['regions_tiers23'], 'step', d['ms_per_step'])"
done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -x -q -m gpu -k "extend" 2>&1 | tail -3
