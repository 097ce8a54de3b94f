cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; grep -c processor /proc/cpuinfo
for ht in 24 12 16; do
  a=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sub --host-threads $ht 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('host-threads $ht: %8.0f reads/s %7.1f ms/step push %.2f cpu %.1f busy %.1f' % (d['value'], d['ms_per_step'], d['push_loop_s_per_step']['in_stream_push'], d['host_cpu_s_per_step']['user']+d['host_cpu_s_per_step']['system'], d['host_cores_busy_per_gpu']))"
  b=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  echo "before: $a"; echo "after:  $b"
done
