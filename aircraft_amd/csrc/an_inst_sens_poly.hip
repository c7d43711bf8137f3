// Step + sensitivity and derivative + sensitivity kernels of the cubic-fit force model: explicit instantiations, built
// with -fno-slp-vectorize (aircraft_amd/build.py UNIT_FLAGS; the reason is next to the declarations in
// ac_kernels_analytic.hpp).
#define AC_AN_SENS_INSTANTIATE 2
#include "ac_kernels_analytic.hpp"
