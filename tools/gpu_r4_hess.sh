# Round 4: kernel breakdown + counters of the second-order path (width 128 and the default model) at 204 800 units.
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4hess}; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_hess.py > $OUT/bench_hess.txt 2>&1
grep -v amdgpu $OUT/bench_hess.txt | tail -1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/trace/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e3:10.1f} us")
PY
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc/$g -- python3 tools/hess_prof.py > $OUT/pmc_$g.log 2>&1 || echo "pass $g failed"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/pmc/*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'stage_tensors' in k or 'step_hess' in k:
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
