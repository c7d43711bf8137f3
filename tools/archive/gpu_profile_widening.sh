# rocprofv3 kernel statistics of the kernels outside the headline bench: modes table, second-order blocks, MHTT loop, cfg4.
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_wide
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide/modes -- python3 tools/bench_modes.py cfg5 poly default > gpurun_out/prof_wide_modes.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide/hess -- python3 tools/bench_hess.py > gpurun_out/prof_wide_hess.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide/mhtt -- python3 examples/mhtt_track.py --batch 1024 --eager > gpurun_out/prof_wide_mhtt.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide/cfg4 -- python3 examples/random_restart_mpc.py --batch 16384 --horizon 100 > gpurun_out/prof_wide_cfg4.log 2>&1 || true
find gpurun_out/prof_wide -name "*kernel_stats.csv"
