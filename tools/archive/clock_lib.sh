#!/bin/bash
# libaircraft_hip_clk.so: the product objects with the three sensitivity kernels rebuilt under -DAC_CLOCKS
# (every wave reports its lifetime in shader cycles and in 100 MHz ticks: tools/diag_clock_ratio.py).  Run build.py first.
set -e
cd "$(dirname "$0")/.."
C=aircraft_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-pass-failed -ffp-contract=on -DAC_CLOCKS"
for tag in clk; do
  O=$C/_obj_$tag; mkdir -p $O
  extra=""
  objs=""
  for f in $C/_obj/*.o; do
    b=$(basename $f .o)
    case $b in
      nn_inst_tiled8_64|aircraft_hip) hipcc $FLAGS $extra -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o";;
      nn_inst_wt8_mfma_sens) hipcc $FLAGS -DAC_CH=2 -mllvm -slp-threshold=6 -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o";;
      *) objs="$objs $f";;
    esac
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o aircraft_amd/libaircraft_hip_$tag.so $objs
  echo built aircraft_amd/libaircraft_hip_$tag.so
done
