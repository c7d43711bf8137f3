#!/bin/bash
# Timing experiments on k_nn_stage_tensors_rev: variants of the library with one phase of the kernel left out (results are
# WRONG in these flavours; only the time of the full-size call is read).  Build here, run the printed command on the GPU box.
set -e
cd "$(dirname "$0")/.."
for v in CONTRACT RAW STORE FWD; do
  UNITS="nn_inst_wt8_mfma_hessrev" tools/variant_lib.sh rev_no_$v -DAC_REV_SKIP_$v > /dev/null
done
UNITS="nn_inst_wt8_mfma_hessrev" tools/variant_lib.sh rev_no_all -DAC_REV_SKIP_CONTRACT -DAC_REV_SKIP_RAW -DAC_REV_SKIP_STORE -DAC_REV_SKIP_FWD > /dev/null
echo 'for v in CONTRACT RAW STORE FWD all; do AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip_rev_no_$v.so python tools/hess_rev_ab.py no_$v | grep step_hess; done'
