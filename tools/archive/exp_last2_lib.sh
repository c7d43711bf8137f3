#!/bin/bash
# Experiment helper: libaircraft_hip_exp.so = every unit of widths 32 / 64 recompiled with -DAC_EXP_LAST2
# (the last layer interleaves two slabs' accumulator chains), the rest taken from the standard build.
set -e
cd "$(dirname "$0")/.."
C=aircraft_amd/csrc; O=$C/_obj_exp; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-pass-failed -ffp-contract=on -DAC_EXP_LAST2"
objs=""
for f in $C/_obj/*.o; do
  b=$(basename $f .o)
  case $b in
    nn_inst_wt2_mfma_*|nn_inst_wt4_mfma_*) hipcc $FLAGS -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o";;
    *) objs="$objs $f";;
  esac
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o aircraft_amd/libaircraft_hip_exp.so $objs
echo built aircraft_amd/libaircraft_hip_exp.so
