cd "$(dirname "$0")/.."
for ht in 16 32 48; do
  timeout 900 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sub --host-threads $ht 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('host-threads %s: %8.0f reads/s  %7.1f ms/step  push %.2f s  host cpu %.1f s/step  busy cores %.1f  phases %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['push_loop_s_per_step']['in_stream_push'], d['host_cpu_s_per_step']['user']+d['host_cpu_s_per_step']['system'], d['host_cores_busy_per_gpu'], {k:v for k,v in d['host_phase_s_per_chunk'].items() if k in ('t_merge','t_matesw','t_primary','t_cigar','t_sam','t_extend')}))
" $ht
done
BSX_TUNE=stream_whole_chunk=1 timeout 900 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sub --host-threads 32 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('whole-chunk, 32 threads: %8.0f reads/s  %7.1f ms/step  busy cores %.1f' % (d['value'], d['ms_per_step'], d['host_cores_busy_per_gpu']))
"
