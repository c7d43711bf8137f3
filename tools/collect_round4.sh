# Copies the summaries of tools/gpu_round4_check.sh (and of the per-topic round-4 scripts) from gpurun_out/ into profiles/,
# unchanged, under their round-4 names.  Run here after the GPU call has merged its output.
set -e
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/r4_bench_final.json $P/r04_bench_final.json
cp $G/r4_bench_n2_rehearsal.json $P/r04_bench_n2_rehearsal_one_gpu.json
cp $G/r4_kernel_stats_final.csv $P/r04_kernel_stats_final.csv
cp $(ls $G/r4_prof_final/*/*_kernel_stats.csv | head -1) $P/r04_kernel_stats_final_rocprof_all_dispatches.csv
cp $G/r4_pmc_counters.json $P/r04_pmc_counters.json
cp $G/r4_pmc_traffic.json $P/r04_pmc_traffic.json
cp $G/r4_batch_sweep.jsonl $P/r04_batch_sweep.jsonl
cp $G/r4_bench_modes.jsonl $P/r04_bench_modes.jsonl
cp $G/r4_bench_hess.txt $P/r04_bench_hess.txt
cp $G/r4_kernel_stats_cfg2_256.csv $P/r04_kernel_stats_cfg2_256.csv
cp $G/r4_kernel_stats_cfg2_4096.csv $P/r04_kernel_stats_cfg2_4096.csv
cp $G/parity_report.jsonl $P/r04_parity_report.jsonl 2>/dev/null || true
tail -5 $G/r4_pytest_gpu_final.log > $P/r04_pytest_gpu_final.txt
cp $G/r4_smi_clocks.txt $P/r04_smi_clocks_power.txt
[ -f $G/r4_kernel_stats_hess.csv ] && cp $G/r4_kernel_stats_hess.csv $P/r04_kernel_stats_hess.csv
[ -f $G/r4_analytic_pmc_after.json ] && cp $G/r4_analytic_pmc_after.json $P/r04_analytic_pmc_after.json
ls $P | grep r04
[ -f $G/r4_analytic_after.txt ] && cp $G/r4_analytic_after.txt $P/r04_analytic_after.txt
[ -f $G/r4_hess_pmc.txt ] && cp $G/r4_hess_pmc.txt $P/r04_hess_pmc.txt
ls $P | grep -c r04
