#!/bin/bash
# A/B of environment settings on the headline workload (hg38-like genome, pipelined stream): tools/hg_ab.sh "" "BSX_STREAM_WHOLE_CHUNK=1" ...
# One bench process per configuration (index built each time: ~30 s); prints reads/s, ms per step, the push thread's time and the host's CPU seconds.
cd "$(dirname "$0")/.."
for cfg in "$@"; do
	env $cfg timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --sub 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-44s %8.0f reads/s  %7.1f ms/step  push %.2f s  host cpu %.1f s/step  busy cores %.1f | kernels %s' % (sys.argv[1] or '(defaults)', d['value'], d['ms_per_step'], d['push_loop_s_per_step']['in_stream_push'], d['host_cpu_s_per_step']['user']+d['host_cpu_s_per_step']['system'], d['host_cores_busy_per_gpu'], ' '.join('%s %.0f' % (k, v) for k, v in d['kernel_ms_per_step'].items())))
" "$cfg"
done
