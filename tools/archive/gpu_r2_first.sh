# Round-2 first GPU call: GPU tests, microbenchmark of v_mfma_f32_32x32x2_f32 + fillers, bench N=1, 2-rank rehearsal.
set -e
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/r2_pytest_gpu_1.log 2>&1 || true
tail -15 gpurun_out/r2_pytest_gpu_1.log
hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma32 tools/micro/mfma32_valu_overlap.hip 2>/dev/null
timeout -k 10 120 /tmp/mfma32 > gpurun_out/r2_micro_mfma32.txt
cat gpurun_out/r2_micro_mfma32.txt
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_1.json 2> gpurun_out/r2_bench_1.err
cat gpurun_out/r2_bench_1.json
AIRCRAFT_BENCH_ONE_GPU=1 AIRCRAFT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2_bench_n2_rehearsal.json 2> gpurun_out/r2_bench_n2_rehearsal.err
cat gpurun_out/r2_bench_n2_rehearsal.json
