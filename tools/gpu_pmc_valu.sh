# PMC counters of the cfg2 "MFMA off" sensitivity kernel at B=4096 (and 256) for one or more library flavours.
#   tools/gpu_pmc_valu.sh <outdir-tag> <libtag|""> ...
set -e
export TMPDIR=/tmp
tag=$1; shift
for t in "$@"; do
  s=${t:+_$t}
  export AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip$s.so
  for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"; do
    g=$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${tag}_pmc/${t:-product}/$g -- python3 bench.py --no-mfma --hidden 64,64,64 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_pmc_${t:-product}_$g.json 2> gpurun_out/${tag}_pmc_${t:-product}_$g.err || echo "pass $g failed"
  done
done
python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob(f'gpurun_out/{tag}_pmc/*/*/*/*_counter_collection.csv')):
    flavour = f.split('/')[2]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'step_sens_tiled' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(flavour, {})[k] = sum(v[-3:]) / len(v[-3:])
for fl, d in out.items():
    w = d.get('SQ_WAVES', 0) or 1
    print(fl, {k: round(v / w, 1) for k, v in d.items()}, 'per wave; waves', w)
json.dump(out, open(f'gpurun_out/{tag}_pmc_valu.json', 'w'), indent=1)
PY
