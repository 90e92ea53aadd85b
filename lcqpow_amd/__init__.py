"""lcqpow_amd -- MI355X-native penalty-homotopy inner loop of LCQPow.

The product is the C-ABI shared library ``liblcqpow_hip.so`` (HIP kernels for gfx950, built from
``lcqpow_amd/csrc``; interface in ``include/lcqp_hip.h``) and the C++ host layer that mirrors the
reference's ``LCQProblem`` / ``SubsolverBase`` surface (``lcqpow_amd/csrc/host``).  This Python
package is only ctypes plumbing: :mod:`lcqpow_amd.capi` binds ``include/lcqp_hip.h`` (used by tests and
``bench.py``) and :mod:`lcqpow_amd.lcqpow` gives the reference's Python surface (``LCQProblem``, ``Options``,
``OutputStatistics``, ``cscWrapper``, enums) over ``include/lcqp_host.h``.  Neither falls back to a CPU path:
using them without the built libraries raises.
"""
from .capi import (Options, Stats, BatchLCQP, BatchPipeline, SubsolverHIP, default_options, lib, library_path, request_hw_queues, solve_mixed,  # noqa: F401
                   util_symv, util_gemv, util_gemv_t, util_symm_product, util_rows_list, chol_solve, device_count, CSCMatrix, SparseBatchLCQP)
