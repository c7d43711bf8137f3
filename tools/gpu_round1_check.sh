set -e
python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu_3.log 2>&1 || true
tail -4 gpurun_out/pytest_gpu_3.log
python bench.py > gpurun_out/bench_3.json 2> gpurun_out/bench_3.err
cat gpurun_out/bench_3.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err || true
find gpurun_out/prof_r1 -name "*stats*" | head
