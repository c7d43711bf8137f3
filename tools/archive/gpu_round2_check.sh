# Round-2 evidence run: GPU tests, bench (N=1, 2-rank rehearsal), rocprofv3 kernel stats of the bench command, PMC passes
# for the headline kernel and for the cfg2 "MFMA off" kernel, batch sweep, every-mode figures.
set -e
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
rm -rf gpurun_out/r2_prof_final gpurun_out/r2_pmc_final gpurun_out/r2_pmc_cfg2 gpurun_out/r2_prof_cfg2_256 gpurun_out/r2_prof_cfg2_4096
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r2_pytest_gpu_final.log 2>&1 || true
tail -3 gpurun_out/r2_pytest_gpu_final.log
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
cat gpurun_out/r2_bench_final.json
AIRCRAFT_BENCH_ONE_GPU=1 AIRCRAFT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r2_bench_n2_rehearsal.json 2> gpurun_out/r2_bench_n2_rehearsal.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_final -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r2_prof_bench.json 2> gpurun_out/r2_prof_bench.err || true
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r2_pmc_final/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2_pmc_final_$tag.json 2> gpurun_out/r2_pmc_final_$tag.err || echo "pass $tag failed"
done
python tools/pmc_traffic.py || true
# the headline line again, now carrying this build's measured traffic (bench.py reads profiles/r02_pmc_traffic.json)
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
cat gpurun_out/r2_bench_final.json
# the cfg2 "MFMA off" kernel (k_nn_step_sens_tiled): kernel stats + counters at B=256 (cfg2) and B=4096
for B in 256 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_cfg2_$B -- python3 bench.py --no-mfma --hidden 64,64,64 --batch $B --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r2_prof_cfg2_$B.json 2> gpurun_out/r2_prof_cfg2_$B.err || true
done
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r2_pmc_cfg2/$tag -- python3 bench.py --no-mfma --hidden 64,64,64 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2_pmc_cfg2_$tag.json 2> gpurun_out/r2_pmc_cfg2_$tag.err || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/r2_prof_final/*/*_kernel_stats.csv')) + sorted(glob.glob('gpurun_out/r2_prof_cfg2_*/*/*_kernel_stats.csv')):
    print('##', f); print(open(f).read())
for base in ('gpurun_out/r2_pmc_final', 'gpurun_out/r2_pmc_cfg2'):
    for f in sorted(glob.glob(base + '/*/*/*_counter_collection.csv')):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'step_sens' in r['Kernel_Name']:
                kern = 'pair' if 'step_sens_pair' in r['Kernel_Name'] else ('tiled' if 'tiled' in r['Kernel_Name'] else 'main')
                agg[r['Counter_Name']][kern].append(float(r['Counter_Value']))
        for k, byk in agg.items():
            means = {kk: sum(v) / len(v) for kk, v in byk.items()}
            print(base.split('/')[-1], k, 'per step =', sum(means.values()), means, 'n =', {kk: len(v) for kk, v in byk.items()})
            out[base.split('/')[-1] + ':' + k] = {'per_step': sum(means.values()), **means}
json.dump(out, open('gpurun_out/r2_pmc_counters.json', 'w'), indent=1)
PY
python tools/batch_sweep.py > gpurun_out/r2_batch_sweep.jsonl 2> gpurun_out/r2_batch_sweep.err || true
cat gpurun_out/r2_batch_sweep.jsonl
python tools/bench_modes.py > gpurun_out/r2_bench_modes.jsonl 2> gpurun_out/r2_bench_modes.err || true
cat gpurun_out/r2_bench_modes.jsonl
python tools/bench_hess.py > gpurun_out/r2_bench_hess.txt 2>&1 || true
cat gpurun_out/r2_bench_hess.txt
# phase stamps of the headline kernel (diagnostic flavour: python aircraft_amd/build.py --diag before the call)
if [ -f aircraft_amd/libaircraft_hip_diag.so ]; then python tools/diag_stamps.py > gpurun_out/r2_stamps_final.txt 2>&1 || true; cat gpurun_out/r2_stamps_final.txt; fi
