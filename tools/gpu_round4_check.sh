# Round-4 evidence run: GPU tests, bench (N=1, 2-rank rehearsal), rocprofv3 kernel stats of the bench command (warm-up excluded),
# PMC passes for the headline kernel and for the cfg2 "MFMA off" kernel, batch sweep, every-mode figures, clocks, the
set -e
export TMPDIR=/tmp
L=$PWD/aircraft_amd
rm -f gpurun_out/parity_report.jsonl
rm -rf gpurun_out/r4_prof_final gpurun_out/r4_pmc_final gpurun_out/r4_pmc_cfg2 gpurun_out/r4_prof_cfg2_256 gpurun_out/r4_prof_cfg2_4096
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r4_pytest_gpu_final.log 2>&1 || true
tail -3 gpurun_out/r4_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_smoke.log 2>&1 || echo "smoke FAILED"
tail -2 gpurun_out/r4_smoke.log
python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err || echo "bench rc=$?"
cat gpurun_out/r4_bench_final.json
AIRCRAFT_BENCH_ONE_GPU=1 AIRCRAFT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r4_bench_n2_rehearsal.json 2> gpurun_out/r4_bench_n2_rehearsal.err || true
cat gpurun_out/r4_bench_n2_rehearsal.json
# kernel trace of the bench command: 1 naming call + 2 warm-up steps + 20 timed steps of the sensitivity kernels
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof_final -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4_prof_bench.json 2> gpurun_out/r4_prof_bench.err || true
python tools/kernel_trace_stats.py $(ls gpurun_out/r4_prof_final/*/*_kernel_trace.csv | head -1) --skip 3 > gpurun_out/r4_kernel_stats_final.csv || true
cat gpurun_out/r4_kernel_stats_final.csv
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r4_pmc_final/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_pmc_final_$tag.json 2> gpurun_out/r4_pmc_final_$tag.err || echo "pass $tag failed"
done
python tools/pmc_traffic.py 4 r4_pmc_final || true
# the headline line again, now carrying this build's measured traffic
python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err || echo "bench rc=$?"
cat gpurun_out/r4_bench_final.json
# the cfg2 "MFMA off" kernel (k_nn_step_sens_tiled8): kernel stats (warm-up excluded) + counters at B=256 (cfg2) and B=4096
for B in 256 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof_cfg2_$B -- python3 bench.py --no-mfma --hidden 64,64,64 --batch $B --steps 20 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4_prof_cfg2_$B.json 2> gpurun_out/r4_prof_cfg2_$B.err || true
  python tools/kernel_trace_stats.py $(ls gpurun_out/r4_prof_cfg2_$B/*/*_kernel_trace.csv | head -1) --skip 3 > gpurun_out/r4_kernel_stats_cfg2_$B.csv || true
  cat gpurun_out/r4_kernel_stats_cfg2_$B.csv
done
bash tools/gpu_pmc_valu.sh r4final "" 2>&1 | tail -2
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/r4_pmc_final/*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'step_sens' in r['Kernel_Name']:
            kern = 'pair' if 'step_sens_pair' in r['Kernel_Name'] else 'main'
            agg[r['Counter_Name']][kern].append(float(r['Counter_Value']))
    for k, byk in agg.items():
        means = {kk: sum(v[-3:]) / len(v[-3:]) for kk, v in byk.items()}
        print('headline', k, 'per step =', sum(means.values()), means)
        out['r4_pmc_final:' + k] = {'per_step': sum(means.values()), **means}
try:
    out['r4_pmc_cfg2'] = json.load(open('gpurun_out/r4final_pmc_valu.json'))
except Exception as e:
    print('no cfg2 counters', e)
json.dump(out, open('gpurun_out/r4_pmc_counters.json', 'w'), indent=1)
PY
python tools/batch_sweep.py > gpurun_out/r4_batch_sweep.jsonl 2> gpurun_out/r4_batch_sweep.err || true
cat gpurun_out/r4_batch_sweep.jsonl
python tools/bench_modes.py > gpurun_out/r4_bench_modes.jsonl 2> gpurun_out/r4_bench_modes.err || true
cat gpurun_out/r4_bench_modes.jsonl
python tools/bench_hess.py > gpurun_out/r4_bench_hess.txt 2>&1 || true
cat gpurun_out/r4_bench_hess.txt
if [ -f $L/libaircraft_hip_clk.so ]; then
  (AIRCRAFT_HIP_LIB=$L/libaircraft_hip_clk.so python tools/diag_clock_ratio.py --no-mfma; AIRCRAFT_HIP_LIB=$L/libaircraft_hip_clk.so python tools/diag_clock_ratio.py --no-mfma --batch 256; AIRCRAFT_HIP_LIB=$L/libaircraft_hip_clk.so python tools/diag_clock_ratio.py --hidden 128,128,128,128) > gpurun_out/r4_clock_ratio_final.txt 2>&1 || true
  grep -v amdgpu gpurun_out/r4_clock_ratio_final.txt
fi
if [ -f $L/libaircraft_hip_diagenv.so ]; then
  AIRCRAFT_HIP_LIB=$L/libaircraft_hip_diagenv.so python tools/pair_stub.py > gpurun_out/r4_pair_stub.jsonl 2>&1 || true
  cat gpurun_out/r4_pair_stub.jsonl
fi
python tools/diag_clocks.py --no-mfma 2>&1 | grep -v amdgpu | head -4 > gpurun_out/r4_smi_clocks.txt || true
python tools/diag_clocks.py --hidden 128,128,128,128 2>&1 | grep -v amdgpu | head -4 >> gpurun_out/r4_smi_clocks.txt || true
cat gpurun_out/r4_smi_clocks.txt
