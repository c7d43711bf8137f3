# Round 4 working loop: parity tests of the step / sensitivity kernels, every-mode timings, headline bench.
#   tools/gpu_r4_check.sh <tag> [pytest -k expression]
export TMPDIR=/tmp
tag=${1:-r4}; kexpr=${2:-}
OUT=gpurun_out/$tag; mkdir -p $OUT
export AIRCRAFT_PARITY_REPORT=$PWD/$OUT/parity_report.jsonl
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_constraints.py -m gpu -x -q ${kexpr:+-k "$kexpr"} > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 tools/bench_modes.py cfg3 cfg2 cfg2_valu real poly default linear > $OUT/bench_modes.jsonl 2> $OUT/bench_modes.err
python3 - $OUT/bench_modes.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"{d['case']:40s} sens {d['sens_ms']*1e3:8.1f} us {d['sens_steps_per_s']:.3e}/s  fwd {d['fwd_ms']*1e3:7.1f} us  rollout {d['rollout_ms']*1e3:7.1f} us  [{d['sens_kernel']}]")
PY
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
