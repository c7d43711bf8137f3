// Step + sensitivity and derivative + sensitivity kernels of the default / linear / quadrotor force models: explicit
// instantiations (see the declarations at the end of ac_kernels_analytic.hpp).
#define AC_AN_SENS_INSTANTIATE 1
#include "ac_kernels_analytic.hpp"
