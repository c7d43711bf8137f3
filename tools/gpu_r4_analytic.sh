# Round 4: kernel trace + PMC passes of the analytic-model and real-net sensitivity kernels at B x H = 204 800 units.
#   tools/gpu_r4_analytic.sh <tag> [models...]
export TMPDIR=/tmp
tag=${1:-r4a}; shift
models=${@:-poly default linear real}
OUT=gpurun_out/$tag
mkdir -p $OUT
for m in $models; do
  python3 tools/prof_sens.py $m > $OUT/plain_$m.txt 2>&1 || echo "plain $m failed"
  cat $OUT/plain_$m.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$m -- python3 tools/prof_sens.py $m > $OUT/trace_$m.log 2>&1 || echo "trace $m failed"
  for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    g=$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$m/$g -- python3 tools/prof_sens.py $m 4096 50 4 > $OUT/pmc_${m}_$g.log 2>&1 || echo "pass $m $g failed"
  done
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + '/pmc_*/*/*/*_counter_collection.csv')):
    m = f.split('/')[2][4:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'step_sens' in k or 'deriv_sens' in k:
            agg[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        for c, v in d.items():
            res.setdefault(m, {}).setdefault(k, {})[c] = sum(v[-4:]) / len(v[-4:])
json.dump(res, open(out + '/pmc_summary.json', 'w'), indent=1)
for m, d in res.items():
    for k, c in d.items():
        print(m, k, {a: round(b) for a, b in c.items()})
PY
