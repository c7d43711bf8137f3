#!/bin/bash
# Copy the summaries of tools/gpu_round3_check.sh from gpurun_out/ (scratch) to profiles/ (tracked).
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/r3_bench_final.json $P/r03_bench_final.json
cp $G/r3_bench_n2_rehearsal.json $P/r03_bench_n2_rehearsal_one_gpu.json
cp $G/r3_kernel_stats_final.csv $P/r03_kernel_stats_final.csv
cp $(ls $G/r3_prof_final/*/*_kernel_stats.csv | head -1) $P/r03_kernel_stats_final_rocprof_all_dispatches.csv
cp $G/r3_kernel_stats_cfg2_256.csv $P/r03_kernel_stats_cfg2_256.csv
cp $G/r3_kernel_stats_cfg2_4096.csv $P/r03_kernel_stats_cfg2_4096.csv
cp $G/r3_pmc_counters.json $P/r03_pmc_counters.json
cp $G/r3_batch_sweep.jsonl $P/r03_batch_sweep.jsonl
cp $G/r3_bench_modes.jsonl $P/r03_bench_modes.jsonl
grep -v amdgpu.ids $G/r3_bench_hess.txt > $P/r03_bench_hess.txt
cp $G/r3_kernel_stats_hess.csv $P/r03_kernel_stats_hess.csv
cp $G/r3_hess_rev_ab.txt $P/r03_hess_rev_ab.txt
cp $G/r3_hess_rev_clocks.txt $P/r03_hess_rev_clocks.txt
grep "stage_tensors\|step_hess" $G/r3_hess_rev_pmc.txt > $P/r03_hess_rev_pmc.txt
grep -v amdgpu.ids $G/r3_clock_ratio_final.txt > $P/r03_wave_clocks.txt
cp $G/r3_pair_stub.jsonl $P/r03_pair_stub.jsonl
cp $G/r3_smi_clocks.txt $P/r03_smi_clocks_power.txt
cp $G/parity_report.jsonl $P/r03_parity_report.jsonl
tail -3 $G/r3_pytest_gpu_final.log > $P/r03_pytest_gpu_final.txt
for f in $G/r3_ab_valu_*.txt $G/r3_ab_persist.txt; do [ -f $f ] && cp $f $P/$(basename $f | sed 's/^r3_/r03_/'); done
ls $P | grep r03 | wc -l
