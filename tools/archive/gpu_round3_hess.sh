# Round-3 evidence, second call (tools/gpu_round3_check.sh fills the first): the width-128 second-order path.
set -e
export TMPDIR=/tmp
L=$PWD/aircraft_amd
# the width-128 second-order path: kernel stats of four full-size calls, same-box A/B of the reverse sweep against the
# slab-per-derivative kernel (-DAC_NO_HESS_REV flavour), per-phase clocks of one wave (-DAC_REV_CLOCKS flavour)
rm -rf gpurun_out/r3_prof_hess
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_hess -- python3 tools/hess_prof.py > gpurun_out/r3_prof_hess.log 2>&1 || true
python tools/kernel_trace_stats.py $(ls gpurun_out/r3_prof_hess/*/*_kernel_trace.csv | head -1) --skip 1 > gpurun_out/r3_kernel_stats_hess.csv || true
cat gpurun_out/r3_kernel_stats_hess.csv
rm -f gpurun_out/hess_rev_*.npz
if [ -f $L/libaircraft_hip_norev.so ] && [ -f $L/libaircraft_hip_rev6.so ]; then
  # norev: one slab per derivative (round 2); rev6: reverse sweep with six slabs; rev: the product (two halves of three slabs)
  (AIRCRAFT_HIP_LIB=$L/libaircraft_hip_norev.so python tools/hess_rev_ab.py norev; AIRCRAFT_HIP_LIB=$L/libaircraft_hip_rev6.so python tools/hess_rev_ab.py rev6; python tools/hess_rev_ab.py rev) 2>&1 | grep -v amdgpu > gpurun_out/r3_hess_rev_ab.txt || true
  cat gpurun_out/r3_hess_rev_ab.txt
fi
if [ -f $L/libaircraft_hip_rev_clk.so ]; then
  AIRCRAFT_HIP_LIB=$L/libaircraft_hip_rev_clk.so python tools/hess_rev_ab.py clk 2>&1 | grep "block 0 wave\|step_hess" | tail -3 > gpurun_out/r3_hess_rev_clocks.txt || true
  cat gpurun_out/r3_hess_rev_clocks.txt
fi
bash tools/gpu_pmc_hessrev.sh > gpurun_out/r3_hess_rev_pmc.txt 2>&1 || true
tail -4 gpurun_out/r3_hess_rev_pmc.txt
