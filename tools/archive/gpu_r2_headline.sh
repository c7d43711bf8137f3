set -e
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/r2_pytest_gpu_6.log 2>&1 || true
tail -3 gpurun_out/r2_pytest_gpu_6.log
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
python -c "import json; d=json.load(open('gpurun_out/r2_bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['one_core']['value'])"
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r2_pmc_final/$grp -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2_pmc_final_$grp.json 2> gpurun_out/r2_pmc_final_$grp.err || echo "pass $grp failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r2_pmc_final/*_SIZE/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'step_sens' in r['Kernel_Name']:
            kern = 'pair' if 'step_sens_pair' in r['Kernel_Name'] else 'main'
            agg[r['Counter_Name']][kern].append(float(r['Counter_Value']))
    for k, byk in agg.items():
        means = {kk: sum(v[-5:]) / len(v[-5:]) for kk, v in byk.items()}
        print(f, k, 'per step =', sum(means.values()), means)
PY
