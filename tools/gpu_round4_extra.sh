# Round-4 evidence, second call: kernel traces + PMC passes of the reference's own models (poly / default / linear / real net,
# step + sensitivities and the derivative kernel of the implicit rows) and of the second-order path, with the closing build.
export TMPDIR=/tmp
bash tools/gpu_r4_analytic.sh r4_analytic_after poly default linear real deriv-poly deriv-default > gpurun_out/r4_analytic_after.log 2>&1
tail -12 gpurun_out/r4_analytic_after.log | cut -c1-400
cp gpurun_out/r4_analytic_after/pmc_summary.json gpurun_out/r4_analytic_pmc_after.json
python3 - <<'PY' > gpurun_out/r4_analytic_after.txt
import csv, glob
for m in ("poly", "default", "linear", "real", "deriv-poly", "deriv-default"):
    try:
        print(open(f"gpurun_out/r4_analytic_after/plain_{m}.txt").read().strip().splitlines()[-1])
    except Exception as e:
        print(m, "no timing line", e)
    for f in glob.glob(f"gpurun_out/r4_analytic_after/trace_{m}/*/*_kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:3]:
            print("   ", r["Name"][:90], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "min_us", round(float(r["MinNs"]) / 1e3, 1))
PY
cat gpurun_out/r4_analytic_after.txt
bash tools/gpu_r4_hess.sh r4_hess_final > gpurun_out/r4_hess_final.log 2>&1
grep -v amdgpu gpurun_out/r4_hess_final.log | tail -14 | cut -c1-400
cp $(ls gpurun_out/r4_hess_final/trace/*/*_kernel_stats.csv | head -1) gpurun_out/r4_kernel_stats_hess.csv
grep -v amdgpu gpurun_out/r4_hess_final.log | grep "stage_tensors\|step_hess" > gpurun_out/r4_hess_pmc.txt
