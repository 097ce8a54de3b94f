#!/bin/bash
# A/B of settings on the headline workload (hg38-like genome, pipelined stream): tools/hg_ab.sh "" "BSX_TUNE=stream_whole_chunk=1" "BSX_TUNE=tier1c=0,x4=0 BSX_STREAM_DEPTH=5" ...
# (the library's settings travel in $BSX_TUNE, csrc/host/tune.c; BSX_STREAM_DEPTH / BSX_HOST_THREADS are environment variables of their own)
# One bench process per configuration (index built each time: ~30 s); prints reads/s, ms per step, the push thread's time and the host's CPU seconds.
cd "$(dirname "$0")/.."
for cfg in "$@"; do
	env $cfg timeout 900 python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline --sub 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-44s %8.0f reads/s  %7.1f ms/step  push %.2f s  host cpu %.1f s/step  busy cores %.1f | kernels %s' % (sys.argv[1] or '(defaults)', d['value'], d['ms_per_step'], d['push_loop_s_per_step']['in_stream_push'], d['host_cpu_s_per_step']['user']+d['host_cpu_s_per_step']['system'], d['host_cores_busy_per_gpu'], ' '.join('%s %.0f' % (k, v) for k, v in d['kernel_ms_per_step'].items())))
" "$cfg"
done
