#!/bin/bash
# Timing experiments on the SIX-SLAB reverse-sweep kernel (k_nn_stage_tensors_rev, the -DAC_HESS_REV6 flavour — not the product,
# which runs the sweep in two halves: k_nn_stage_tensors_rev3): variants of the library with one phase of the kernel left out.
# Results are WRONG in these flavours (and the compiler may drop work whose result is no longer used: read them with the
# per-phase clocks of -DAC_REV_CLOCKS beside them); only the time of the full-size call is read.
# Build here, run the printed command on the GPU box.
set -e
cd "$(dirname "$0")/.."
for v in CONTRACT RAW STORE FWD; do
  UNITS="aircraft_hip nn_inst_wt8_mfma_hessrev" tools/variant_lib.sh rev_no_$v -DAC_HESS_REV6 -DAC_REV_SKIP_$v > /dev/null
done
echo 'for v in CONTRACT RAW STORE FWD; do AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip_rev_no_$v.so python tools/hess_rev_ab.py no_$v | grep step_hess; done'
