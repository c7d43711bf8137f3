set -e
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_constraints.py -m gpu -q --timeout 900 -p no:cacheprovider --tb=short -k "valu or constraints" > gpurun_out/r2_pytest_valu.log 2>&1 || true
tail -25 gpurun_out/r2_pytest_valu.log | cut -c1-300
for B in 256 4096; do
python bench.py --no-mfma --hidden 64,64,64 --batch $B --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 valu B=$B', d['value'], 'steps/s', d['roofline']['kernel_ms'], 'ms', d['roofline']['kernel'], 'frac', d['roofline']['frac'])"
python bench.py --hidden 64,64,64 --batch $B --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 mfma B=$B', d['value'], 'steps/s', d['roofline']['kernel_ms'], 'ms', d['roofline']['kernel'], 'frac', d['roofline']['frac'])"
done
