export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_cfg5_trace -- python3 tools/bench_modes.py cfg5 > gpurun_out/r4_cfg5_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r4_cfg5_trace/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
tail -2 gpurun_out/r4_cfg5_trace.log | cut -c1-300
