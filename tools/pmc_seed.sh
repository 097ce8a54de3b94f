set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc2; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python $ROOT/bench.py --genome-mbp ${GENOME_MBP:-128} --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline > /dev/null 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_WAIT_ANY SQ_WAVES SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TD_BUSY_avr"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-60)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -o b -- python $ROOT/bench.py --genome-mbp ${GENOME_MBP:-128} --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline > $OUT/$name.log 2>&1 || echo "FAILED $grp" >> $OUT/errors.txt
done
cd $OUT; python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('*/b_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if k.startswith('k_seed') or k.startswith('k_regions') or k.startswith('k_occ'):
            print(k, {a:('%.3e'%b) for a,b in v.items()})
PY
cat $OUT/errors.txt 2>/dev/null
