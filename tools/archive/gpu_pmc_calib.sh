export TMPDIR=/tmp
OUT=gpurun_out/pmc_calib
mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 tools/pmc_calib.py > $OUT/$tag.out 2> $OUT/$tag.err || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_calib/*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'step_sens' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, 'n=',len(v), 'mean=', sum(v)/len(v))
PY
