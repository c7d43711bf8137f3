# PMC counters of k_nn_stage_tensors_rev (tools/hess_prof.py: four full-size calls): instruction fetch and issue.
set -e
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/hessrev_pmc/$g -- python3 tools/hess_prof.py > gpurun_out/hessrev_pmc_$g.log 2>&1 || echo "pass $g failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/hessrev_pmc/*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'stage_tensors' in k or 'step_hess' in k:
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
