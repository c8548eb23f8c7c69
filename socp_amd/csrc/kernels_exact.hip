// kernels_exact.hip -- reference-operation-order kernels.  MUST be compiled with
// -ffp-contract=off (see socp_amd/csrc/Makefile): the parity tests compare these against the
// CPU oracle at the few-ulp level.
#include "models_exact.hpp"
#define SOCP_FLAVOUR exact
#define SOCP_DEFINE_COMMON 1   // fd_diff lives in the no-contraction TU
#define SOCP_GODDARD GoddardExact
#define SOCP_GODDARD_SMOOTH GoddardExactSmooth
#define SOCP_DINT DIntExact
#include "launch_impl.hpp"
