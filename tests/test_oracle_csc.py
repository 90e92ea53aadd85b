"""CSC half of the Utilities restatement, pinned by the reference's known-answer tests
test/RunUnitTests.cpp:265-410 and by dense equivalents."""
import ctypes as C

import numpy as np


def test_csc_to_dns_kat(oracle):      # RunUnitTests.cpp:265-332
    M = oracle.csc_create(2, 3, [2.0, 1.0, 2.0], [0, 0, 1], [0, 1, 3, 4])
    assert oracle.csc_to_dns(M).ravel().tolist() == [2, 1, 0, 0, 2, 0]
    M = oracle.csc_create(2, 3, [2.0, 1.0, 10.0], [0, 1, 1], [0, 2, 2, 4])
    assert oracle.csc_to_dns(M).ravel().tolist() == [2, 0, 0, 1, 0, 10]
    T = oracle.csc_create(3, 2, [2.0, 10.0, 1.0], [0, 2, 0], [0, 2, 3])
    assert oracle.csc_to_dns(T).ravel().tolist() == [2, 1, 0, 0, 10, 0]


def test_sparse_dense_back_and_forth(oracle):      # :335-375 (seeded instead of time(NULL))
    rng = np.random.default_rng(3)
    for _ in range(100):
        Q = np.where(rng.integers(0, 4, (2, 5)) == 0, rng.integers(0, 9, (2, 5)), 0).astype(float)
        S = oracle.dns_to_csc(Q)
        assert np.array_equal(oracle.csc_to_dns(S), Q)


def test_csc_to_triangular_kat(oracle):      # :378-410
    M = oracle.csc_create(2, 2, [2.0, 3.0, 3.0, 2.0], [0, 1, 0, 1], [0, 2, 4])
    U = oracle._csc_setup().orc_csc_upper(M)
    m, n, p, i, x = oracle.csc_arrays(U)
    assert (m, n) == (2, 2) and p.tolist() == [0, 1, 3] and i.tolist() == [0, 0, 1] and x.tolist() == [2, 3, 2]
    assert U.contents.nzmax == 3 and U.contents.nz == -1


def test_csc_products_match_dense(oracle):
    L = oracle._csc_setup()
    rng = np.random.default_rng(9)
    m, n = 7, 5
    A = np.where(rng.random((m, n)) < 0.4, rng.standard_normal((m, n)), 0.0)
    S = oracle.dns_to_csc(A)
    b = rng.standard_normal(n); bt = rng.standard_normal(m)
    c = np.zeros(m); L.orc_csc_matmul(S, oracle._p(b), oracle._p(c))
    assert np.allclose(c, A @ b, atol=1e-14)
    ct = np.zeros(n); L.orc_csc_matmul_t(S, oracle._p(bt), oracle._p(ct))
    assert np.allclose(ct, A.T @ bt, atol=1e-14)
    L.orc_csc_add_matmul_t(S, oracle._p(bt), oracle._p(ct))
    assert np.allclose(ct, 2 * (A.T @ bt), atol=1e-14)
    # symmetric matrix: affine map and quadratic form
    Qd = np.where(rng.random((n, n)) < 0.5, rng.standard_normal((n, n)), 0.0); Qd = Qd + Qd.T
    Qs = oracle.dns_to_csc(Qd)
    cc = rng.standard_normal(n); d = np.zeros(n)
    L.orc_csc_affine(2.0, Qs, oracle._p(b), oracle._p(cc), oracle._p(d), n)
    assert np.allclose(d, 2.0 * Qd @ b + cc, atol=1e-13)
    assert abs(L.orc_csc_quadform(Qs, oracle._p(b), n) - b @ Qd @ b) < 1e-12
    # C = L'R + R'L against the dense restatement (which carries the reference's known answer)
    Ld = np.where(rng.random((3, n)) < 0.5, rng.standard_normal((3, n)), 0.0)
    Rd = np.where(rng.random((3, n)) < 0.5, rng.standard_normal((3, n)), 0.0)
    Cs = L.orc_csc_symm_product(oracle.dns_to_csc(Ld), oracle.dns_to_csc(Rd))
    assert np.allclose(oracle.csc_to_dns(Cs), oracle.util_symm_product(Ld, Rd, 3, n).reshape(n, n), atol=1e-14)
    # the dense known answer of RunUnitTests.cpp:81-104 through the sparse routine
    Cs = L.orc_csc_symm_product(oracle.dns_to_csc(np.array([[1., 0, 2], [3, 1, 1]])), oracle.dns_to_csc(np.array([[2., 0, 1], [0, 0, -1]])))
    assert oracle.csc_to_dns(Cs).ravel().tolist() == [4, 0, 2, 0, 0, -1, 2, -1, 2]
