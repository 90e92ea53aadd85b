"""lcqpow_amd -- MI355X-native penalty-homotopy inner loop of LCQPow.

The product is the C-ABI shared library ``liblcqpow_hip.so`` (HIP kernels for gfx950, built from
``lcqpow_amd/csrc``; interface in ``include/lcqp_hip.h``) and the C++ host layer that mirrors the
reference's ``LCQProblem`` / ``SubsolverBase`` surface (``lcqpow_amd/csrc/host``).  This Python
package is only the ctypes plumbing used by tests and ``bench.py``; it never falls back to a CPU
path: importing :mod:`lcqpow_amd.capi` without the built library raises.
"""
from .capi import (Options, Stats, BatchLCQP, SubsolverHIP, default_options, lib, library_path,  # noqa: F401
                   util_symv, util_gemv, util_gemv_t, util_symm_product, chol_solve, device_count, CSCMatrix)
