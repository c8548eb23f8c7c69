// kernels_fast.hip -- throughput flavour: compiled with FMA contraction, restructured models
// (models_fast.hpp).  Results agree with the reference order to rounding level, not bitwise;
// the tolerance is stated and tested in tests/test_gpu_parity.py.
#include "models_fast.hpp"
#define SOCP_FLAVOUR fast
#define SOCP_HAVE_DOPRI5 1      // adaptive Dormand-Prince on the restructured right-hand sides too
#define SOCP_GODDARD GoddardFast
#define SOCP_GODDARD_SMOOTH GoddardFastSmooth
#define SOCP_COVID CovidFast
#define SOCP_DINT DIntFast
#include "launch_impl.hpp"
