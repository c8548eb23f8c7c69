"""socp_amd -- MI355X (gfx950) implementation of SOCP's data-parallel hot path.

Batched state+costate trajectory integration, the fused shooting residual and the
finite-difference Jacobian that feeds the MINPACK hybrd Newton solve, behind a C-ABI
(include/socp_hip.h) and a C++ mirror of the reference's model / shooting classes
(socp_amd/host).  `socp_amd.capi` holds the ctypes plumbing used by tests and bench.py.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
