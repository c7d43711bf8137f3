#!/bin/bash
# Same-box A/B helper: libaircraft_hip_<tag>.so = the product's objects with the units named in UNITS recompiled under extra flags
# (their per-unit product flags, aircraft_amd/build.py UNIT_FLAGS, are NOT applied: pass what you need).  Run aircraft_amd/build.py first.
#   UNITS="an_inst_sens" tools/variant_lib.sh wps3 -DAC_SENS_WPS=3     then on the GPU box:  AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip_wps3.so python tools/bench_modes.py default
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=aircraft_amd/csrc; O=$C/_obj_$tag; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-pass-failed -ffp-contract=on"
objs=""
for f in $C/_obj/*.o; do
  b=$(basename $f .o)
  case " $UNITS " in *" $b "*) hipcc $FLAGS "$@" -c $C/$b.hip -o $O/$b.o & objs="$objs $O/$b.o";; *) objs="$objs $f";; esac
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o aircraft_amd/libaircraft_hip_$tag.so $objs
echo built aircraft_amd/libaircraft_hip_$tag.so
