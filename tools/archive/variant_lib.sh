#!/bin/bash
# Experiment helper: libaircraft_hip_<tag>.so = the standard objects with the headline kernels (wt8 sens + pair units)
# recompiled under extra flags.   usage: tools/variant_lib.sh <tag> [-DMACRO=... ...]   (run aircraft_amd/build.py first)
# UNITS="nn_inst_x nn_inst_y" recompiles those units instead (the headline units keep the product flags).
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=aircraft_amd/csrc; O=$C/_obj_$tag; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-pass-failed -ffp-contract=on"
objs=""
for f in $C/_obj/*.o; do
  b=$(basename $f .o)
  case $b in
    nn_inst_wt8_mfma_sens|nn_inst_wt8_mfma_pair)
      if [ -z "$UNITS" ]; then hipcc $FLAGS "$@" -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o"
      else hipcc $FLAGS -DAC_CH=2 -mllvm -slp-threshold=6 -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o"; fi;;
    *) case " $UNITS " in *" $b "*) hipcc $FLAGS "$@" -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o";; *) objs="$objs $f";; esac;;
  esac
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o aircraft_amd/libaircraft_hip_$tag.so $objs
echo built aircraft_amd/libaircraft_hip_$tag.so
