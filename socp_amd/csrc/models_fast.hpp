// models_fast.hpp -- restructured device dynamics for the throughput flavour.
// (first cut: the reference-order models under FMA contraction; the reciprocal-restructured
// Goddard RHS replaces GoddardFast below.)
#pragma once
#include "models_exact.hpp"
namespace socp {
using GoddardFast = GoddardExact;
using DIntFast = DIntExact;
}
