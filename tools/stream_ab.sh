#!/bin/bash
# where the stream's time goes: is the pushing thread waiting for front halves or running back halves?  And back halves on the chunks' own threads.
BSX_PHASES=1 timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-hard-genome 2> /tmp/st.err | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('default (traced)', d['value'], d['ms_per_step'], d['push_loop_s_per_step'])"
grep "M::stream" /tmp/st.err | tail -6 | cut -c1-260
for cfg in "BSX_STREAM_WHOLE_CHUNK=0" "BSX_STREAM_WHOLE_CHUNK=1" "BSX_STREAM_WHOLE_CHUNK=1 BSX_STREAM_DEPTH=5" "BSX_STREAM_WHOLE_CHUNK=1 BSX_STREAM_DEPTH=6" "BSX_STREAM_WHOLE_CHUNK=0 BSX_STREAM_DEPTH=5"; do
  env $cfg timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-hard-genome 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', d['value'], d['ms_per_step'], d['push_loop_s_per_step'], d['host_cpu_s_per_step'])"
done
