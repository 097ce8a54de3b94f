#!/bin/bash
# pipelined bench with different host pool sizes (the stream is bound by the host back half)
for t in "$@"; do timeout 300 python bench.py --steps 16 --no-cpu-baseline --host-threads $t 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print($t,d['value'],d['ms_per_step'],d['host_cpu_s_per_step'])"; done
