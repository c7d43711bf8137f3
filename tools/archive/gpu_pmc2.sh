export TMPDIR=/tmp
OUT=gpurun_out/pmc_r1b
mkdir -p $OUT
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/$tag.json 2> $OUT/$tag.err || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_r1b/*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'step_sens' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k,'n=',len(v),'mean=',sum(v)/len(v))
PY
