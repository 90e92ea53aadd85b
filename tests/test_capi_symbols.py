"""The C-ABI library loads on a CPU-only box and exports every symbol include/lcqp_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "lcqp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lcqp_hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_declared_symbols():
    import lcqpow_amd
    L = ctypes.CDLL(lcqpow_amd.library_path())
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n


def test_options_default_matches_reference_defaults():
    import lcqpow_amd as la
    o = la.default_options()
    assert o.complementarityTolerance == 1e3 * 2.221e-16 and o.stationarityTolerance == 1e6 * 2.221e-16
    assert (o.initialPenaltyParameter, o.penaltyUpdateFactor, o.maxPenaltyParameter) == (0.01, 2.0, 1e8)
    assert (o.solveZeroPenaltyFirst, o.perturbStep, o.maxIterations, o.nDynamicPenalty) == (1, 1, 1000, 3)


def test_option_structs_have_identical_layout():
    import lcqpow_amd as la
    import oracle_py
    assert [f[0] for f in la.Options._fields_] == [f[0] for f in oracle_py.Options._fields_]
    assert ctypes.sizeof(la.Options) == ctypes.sizeof(oracle_py.Options)
    assert ctypes.sizeof(la.Stats) == ctypes.sizeof(oracle_py.Stats)


def test_no_product_import_of_oracle():
    """the shipped path must never route through the oracle"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lcqpow_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"#\s*include[^\n]*oracle", txt), f
                assert not re.search(r"\bimport\s+oracle_py|from\s+oracle_py", txt), f
                assert "liblcqp_oracle" not in txt and not re.search(r"\borc_[a-z_]+\s*\(", txt), f


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """argument checks of the C ABI that need no device, and the loud failure of every compute entry when none is visible"""
    import numpy as np
    import lcqpow_amd as la
    from lcqpow_amd import capi
    L = la.lib()
    assert L.lcqp_hip_batch_create(0, 4, 1, 1, 0, 0) is None and "invalid" in capi.last_error()
    assert L.lcqp_hip_batch_create(4, 4, -1, 1, 0, 0) is None
    assert L.lcqp_hip_batch_create(1, 4097, 0, 1, 0, 0) is None and "4096" in capi.last_error()
    Q = np.eye(2); dp = ctypes.POINTER(ctypes.c_double)
    assert L.lcqp_hip_qp_create(0, 0, Q.ctypes.data_as(dp), None, None, 0) is None
    assert L.lcqp_hip_qp_create(2, 1, Q.ctypes.data_as(dp), None, None, 0) is None          # nC > 0 without A
    assert L.lcqp_hip_batch_run(None) != 0 and L.lcqp_hip_batch_setup(None) != 0
    n = ctypes.c_int(0)
    if la.device_count() == 0:
        # no GPU here: creating a batch must fail with the HIP error, never fall back to anything
        assert L.lcqp_hip_batch_create(2, 4, 1, 1, 0, 0) is None and capi.last_error() != ""
        q = L.lcqp_hip_qp_create(2, 0, Q.ctypes.data_as(dp), None, None, 0)           # host-side object only
        assert q is not None
        it, ef = ctypes.c_int(0), ctypes.c_int(0)
        g = np.zeros(2)
        rc = L.lcqp_hip_qp_solve(ctypes.c_void_p(q), 1, ctypes.byref(it), ctypes.byref(ef), g.ctypes.data_as(dp), None, None, None, None, None, None)
        assert rc == capi.SUBPROBLEM_SOLVER_ERROR and ef.value != 0
        L.lcqp_hip_qp_destroy(ctypes.c_void_p(q))


def test_missing_extension_fails_loudly(monkeypatch):
    """no CPU fallback: without the built HIP library the binding raises instead of computing anything"""
    import pytest
    from lcqpow_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "_SO", "/nonexistent/liblcqpow_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.lib()
    with pytest.raises(RuntimeError):
        capi.BatchLCQP(1, 2, 0, 1)


def test_sparse_pattern_validation_needs_no_gpu():
    """lcqp_hip_sparse_create checks the CSC pattern before it touches a device: decreasing column pointers, unsorted or duplicated row
    indices, an upper-triangular Q (OSQP's own convention for P -- this surface, like the reference's, takes the full matrix) and a short
    index array are refused with a message (src/SubsolverOSQP.cpp:136-152 hands such arrays to OSQP unchecked)"""
    import numpy as np
    import lcqpow_amd as la

    class Pat:
        def __init__(self, indptr, indices):
            self.indptr, self.indices = np.array(indptr, dtype=np.int32), np.array(indices, dtype=np.int32)
    n, nC, nK = 4, 0, 1
    Aok = Pat([0, 1, 2, 2, 2], [0, 1])                                  # L = e_0', R = e_1'
    Qok = Pat([0, 2, 4, 5, 6], [0, 1, 0, 1, 2, 3])
    cases = [(Pat([0, 2, 1, 5, 6], [0, 1, 0, 1, 2, 3]), Aok, "decrease"),
             (Pat([0, 2, 4, 5, 6], [1, 0, 0, 1, 2, 3]), Aok, "sorted"),
             (Pat([0, 2, 4, 5, 6], [0, 0, 0, 1, 2, 3]), Aok, "duplicates"),
             (Pat([0, 1, 3, 4, 5], [0, 0, 1, 2, 3]), Aok, "symmetric"),          # upper triangle only
             (Qok, Pat([0, 1, 2, 2, 2], [0, 7]), "out of bounds")]
    for Qp, Ap, word in cases:
        with pytest.raises(RuntimeError, match=word):
            la.SparseBatchLCQP(1, n, nC, nK, Qp, Ap)
    with pytest.raises(ValueError, match="indptr"):
        la.SparseBatchLCQP(1, n, nC, nK, Pat([0, 2, 4, 5, 6], [0, 1, 0, 1, 2]), Aok)


def test_dense_size_limit_is_reported_not_hidden():
    """Limits the reference does not have: the dense kernels take nV <= 4096 (padded sizes 128 ... 4096; round 3: 1024) and at most
    min(max(2 nV, 64), rows, 896 / 1216 / 2432 / 3264) simultaneously active rows (tests/test_gpu_parity.py::test_subsolver_active_row_capacity).  A larger
    problem is refused at creation with a message -- through the C ABI (NULL handle) and through LCQProblem (SUBPROBLEM_SOLVER_ERROR from
    runSolver, the reference's return code for a subsolver that cannot take the problem, src/SubsolverQPOASES.cpp:165-168) -- never run wrongly.
    The size check precedes any device call, so this holds on a CPU-only box too."""
    import numpy as np
    import lcqpow_amd as la
    from lcqpow_amd import capi
    L = la.lib()
    assert L.lcqp_hip_batch_create(1, 4097, 0, 1, 0, 0) is None
    assert "nV > 4096" in capi.last_error()
    h = L.lcqp_hip_batch_create(1, 4096, 0, 1, 0, 0)          # the largest size: accepted as far as the size goes (no device here: NULL for another reason)
    if h is None:
        assert "nV > 4096" not in capi.last_error()
    else:
        L.lcqp_hip_batch_destroy.argtypes = [ctypes.c_void_p]; L.lcqp_hip_batch_destroy(ctypes.c_void_p(h))
    import lcqpow_amd.lcqpow as lcqpow
    n = 4097
    p = lcqpow.LCQProblem(nV=n, nC=0, nComp=1)
    o = lcqpow.Options(); o.setPrintLevel(lcqpow.PrintLevel.NONE); p.setOptions(o)
    Lm = np.zeros((1, n)); Lm[0, 0] = 1.0; Rm = np.zeros((1, n)); Rm[0, 1] = 1.0
    assert p.loadLCQP(Q=np.eye(n), g=-np.ones(n), L=Lm, R=Rm, order="C") == 0
    assert int(p.runSolver()) == capi.SUBPROBLEM_SOLVER_ERROR
    assert "nV > 4096" in capi.last_error()
