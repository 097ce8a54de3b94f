#!/bin/bash
# tools/pmc_quick.sh "<COUNTERS>" [bench args]: one rocprofv3 --pmc pass (with --kernel-trace only) of a one-chunk unpipelined bench,
# counter sums per kernel printed.  Run on the GPU box from the repo root.
set -u
ROOT=$(pwd); GRP="$1"; shift
OUT=$ROOT/gpurun_out/pmcq_$(echo $GRP | tr ' ' '_' | cut -c1-40); mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc $GRP --output-format csv -d $OUT -o b -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in sorted(agg.items()):
        if k.startswith(('k_seed', 'k_regions', 'k_c2r', 'k_occ', 'k_sw', 'k_global')):
            print('%-46s' % k, ' '.join('%s=%.3e' % (a, b) for a, b in sorted(v.items())))
PY
