"""The reference's own known-answer tests for Utilities (test/RunUnitTests.cpp:33-246) applied to the
oracle's restatement: this is what pins oracle/lcqp_oracle.c against the reference."""
import numpy as np


def test_matrix_multiplication(oracle):   # RunUnitTests.cpp:33-57
    C = oracle.util_matmul([1, 0, 2, 3, 1, 1], [2, 0, 0, 2, 1, 0, 0, 1, 0, -1, -1, 0], 2, 3, 4)
    assert list(C) == [2, -2, -2, 2, 7, -1, -1, 7]


def test_transposed_matrix_multiplication(oracle):   # :60-78
    C = oracle.util_matmul_t([1, 0, 2, 3, 1, 1], [98, -10], 2, 3, 1)
    assert list(C) == [68, -10, 186]


def test_matrix_symmetrization(oracle):   # :81-104 (asserted values, not the comment)
    C = oracle.util_symm_product([1, 0, 2, 3, 1, 1], [2, 0, 1, 0, 0, -1], 2, 3)
    assert list(C) == [4, 0, 2, 0, 0, -1, 2, -1, 2]


def test_affine_transformation(oracle):   # :107-129
    d = oracle.util_affine(2.0, [1, 0, 2, 3, 1, 1], [2, 0, 1], [-3, -3], 2, 3)
    assert list(d) == [5, 11]


def test_matrix_add(oracle):   # :132-159
    C = oracle.util_weighted_matadd(-1.0, [0, 1, 3, 1, 10, 1], 0.5, [2, 0, 0, 4, 2, 2], 3, 2)
    assert list(C) == [1, -1, -3, 1, -9, 0]


def test_vector_add(oracle):   # :162-187
    d = oracle.util_weighted_vecadd(2.0, [0, 1, 2, 3], -1.0, [10, 2, 0, 3], 4)
    assert list(d) == [-10, 0, 4, 3]


def test_quadratic_form(oracle):   # :190-204
    assert oracle.util_quadform([0, 1, 0, 1, 2, 1, 0, 1, 0], [1, 2, 3], 3) == 24


def test_dot_product(oracle):   # :207-221
    assert oracle.util_dot([0, 1, 2, 3], [10, 2, 0, 3], 4) == 11


def test_max_abs(oracle):   # :224-246
    assert oracle.util_maxabs([0, 1, 2, 3], 4) == 3
    assert oracle.util_maxabs([0, -1, 2, 0], 4) == 2
    assert oracle.util_maxabs([0, -4, 2, 0], 4) == 4


def test_options_defaults(oracle):   # src/Options.cpp:296-318
    o = oracle.default_options()
    assert o.complementarityTolerance == 1e3 * 2.221e-16
    assert o.stationarityTolerance == 1e6 * 2.221e-16
    assert (o.initialPenaltyParameter, o.penaltyUpdateFactor) == (0.01, 2.0)
    assert (o.solveZeroPenaltyFirst, o.perturbStep, o.maxIterations) == (1, 1, 1000)
    assert (o.maxPenaltyParameter, o.nDynamicPenalty, o.etaDynamicPenalty) == (1e8, 3, 0.9)
    assert (o.printLevel, o.storeSteps) == (2, 0)


def test_utilities_against_numpy(oracle):
    rng = np.random.default_rng(5)
    m, n, p = 7, 5, 3
    A = rng.standard_normal((m, n)); B = rng.standard_normal((n, p)); Bt = rng.standard_normal((m, p))
    assert np.allclose(oracle.util_matmul(A, B, m, n, p).reshape(m, p), A @ B, atol=1e-14)
    assert np.allclose(oracle.util_matmul_t(A, Bt, m, n, p).reshape(n, p), A.T @ Bt, atol=1e-14)
    L = rng.standard_normal((m, n)); R = rng.standard_normal((m, n))
    assert np.allclose(oracle.util_symm_product(L, R, m, n).reshape(n, n), L.T @ R + R.T @ L, atol=1e-14)
