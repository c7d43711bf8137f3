# Round 4: the cfg2 "MFMA off" kernel at and around its own size (B = 256, H = 50): kernel time over B, counters at B = 256.
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4cfg2}; mkdir -p $OUT
for B in 64 128 160 200 256 320 400 512 1024; do
  python3 bench.py --no-mfma --hidden 64,64,64 --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(f\"B=$B units {d['config']['units_per_step']:6d} groups {d['config']['units_per_step']//8:5d} kernel {r['kernel_ms']*1e3:7.1f} us  {d['value']:.3e} steps/s  frac {r['frac']:.3f}\")"
done | tee $OUT/batch_times.txt
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQC_ICACHE_MISSES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-30)
  for B in 256 4096; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$B/$g -- python3 bench.py --no-mfma --hidden 64,64,64 --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_${B}_$g.log 2>&1 || echo "pass $B $g failed"
  done
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + '/pmc_*/*/*/*_counter_collection.csv')):
    B = f.split('/')[2][4:]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'step_sens_tiled8' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in agg.items():
        res.setdefault(B, {})[c] = sum(v[-3:]) / len(v[-3:])
json.dump(res, open(out + '/pmc_cfg2.json', 'w'), indent=1)
for B, d in res.items():
    w = d.get('SQ_WAVES', 1)
    print('B =', B, {k: round(v / w) for k, v in d.items()}, 'per wave; waves', w)
PY
