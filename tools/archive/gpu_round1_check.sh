# Round-1 evidence run: GPU tests, bench with CPU baseline, rocprof kernel stats, PMC passes.
set -e
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu_final.log 2>&1 || true
tail -3 gpurun_out/pytest_gpu_final.log
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err || true
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_final/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_final_$tag.json 2> gpurun_out/pmc_final_$tag.err || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof_final/*/*_kernel_stats.csv')): print(open(f).read())
for f in sorted(glob.glob('gpurun_out/pmc_final/*/*/*_counter_collection.csv')):
    # one bench step = k_nn_step_sens (whole rounds) + k_nn_step_sens_pair (the remainder): per-step figure = sum of the means
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'step_sens' in r['Kernel_Name']:
            kern = 'pair' if 'step_sens_pair' in r['Kernel_Name'] else 'main'
            agg[r['Counter_Name']][kern].append(float(r['Counter_Value']))
    for k,byk in agg.items():
        means={kk: sum(v)/len(v) for kk,v in byk.items()}
        print(k, 'per step =', sum(means.values()), means, 'n =', {kk: len(v) for kk,v in byk.items()})
PY
