# Round-3 first GPU call: the new tests (selection kernels, full-size cfg2 / cfg5), then the whole GPU suite, smoke, bench.
set -e
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
python -m pytest tests/test_gpu_multigpu.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "best_records or accept or cfg2_full or cfg5_full or gather_best or cfg4_shard_solve or refuses" > gpurun_out/r3_new_tests.log 2>&1 || true
tail -15 gpurun_out/r3_new_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1 || true
tail -3 gpurun_out/r3_smoke.log
python bench.py > gpurun_out/r3_bench_1.json 2> gpurun_out/r3_bench_1.err || echo "bench rc=$?"
cat gpurun_out/r3_bench_1.json
tail -3 gpurun_out/r3_bench_1.err
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r3_pytest_gpu_1.log 2>&1 || true
tail -8 gpurun_out/r3_pytest_gpu_1.log
