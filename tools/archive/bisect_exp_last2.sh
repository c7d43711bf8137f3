#!/bin/bash
# is the NaN of k_nn_stage_tensors<2,true,0> on single-layer nets in the -DAC_EXP_LAST2 flavour (with the
# runtime activation flag inside the second-order epilogues, -DAC_EXP_RUNTIME_ACT) a compiler defect or undefined behaviour in
# the source?  Runs ON THE GPU BOX: rebuilds the one translation unit under `-mllvm -opt-bisect-limit=N`, links it with the
# product objects (copied to aircraft_amd/csrc/_objx so that they travel), runs the exposing test, and bisects N.
#   usage: tools/bisect_exp_last2.sh            (about 15 compile + test cycles)
set -u
cd "$(dirname "$0")/.."
C=aircraft_amd/csrc; O=$C/_objx
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-pass-failed -ffp-contract=on -DAC_EXP_LAST2 -DAC_EXP_RUNTIME_ACT"
T=tests/test_gpu_hessian.py
run() {  # $1 = extra flags; echoes PASS / FAIL
  hipcc $FLAGS $1 -c $C/nn_inst_wt2_mfma_hess.hip -o /tmp/hess_exp.o 2>/tmp/hess_exp.log || { echo COMPILE_ERROR; return; }
  objs=""; for f in $O/*.o; do case $(basename $f .o) in nn_inst_wt2_mfma_hess) objs="$objs /tmp/hess_exp.o";; *) objs="$objs $f";; esac; done
  hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libaircraft_hip_bisect.so $objs || { echo LINK_ERROR; return; }
  if AIRCRAFT_HIP_LIB=/tmp/libaircraft_hip_bisect.so timeout -k 10 300 python -m pytest $T -m gpu -q -x -p no:cacheprovider -k "single_layer" > /tmp/bisect_test.log 2>&1; then echo PASS; else echo FAIL; fi
}
echo "unlimited: $(run "")"
echo "unlimited, every accumulation / vector / scalar register and the LDS zeroed at kernel entry (-DAC_EXP_ZERO_REGS): $(run "-DAC_EXP_ZERO_REGS")"
FLAGS_SAVE=$FLAGS; FLAGS=${FLAGS/-DAC_EXP_RUNTIME_ACT/}
echo "the product's form (activation of the last / only layer dispatched at compile time) under -DAC_EXP_LAST2: $(run "")"
FLAGS=$FLAGS_SAVE
echo "-O1: $(run "-O1")"
[ "${1:-}" = "quick" ] && exit 0
# total number of bisectable steps of the device compilation
hipcc $FLAGS -mllvm -opt-bisect-limit=-1 -c $C/nn_inst_wt2_mfma_hess.hip -o /tmp/x.o 2> /tmp/bisect_all.log
total=$(grep -c "BISECT: running pass" /tmp/bisect_all.log)
echo "bisectable pass executions (host + device): $total"
lo=0; hi=$total   # invariant: limit lo passes, limit hi fails (if unlimited fails)
echo "limit 0: $(run "-mllvm -opt-bisect-limit=0")"
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(( (lo + hi) / 2 ))
  r=$(run "-mllvm -opt-bisect-limit=$mid")
  echo "limit $mid: $r"
  if [ "$r" = "PASS" ]; then lo=$mid; else hi=$mid; fi
done
echo "first failing limit: $hi"
grep "BISECT: running pass ($hi)" /tmp/bisect_all.log | head -3
grep "BISECT: running pass ($((hi - 1)))" /tmp/bisect_all.log | head -3
