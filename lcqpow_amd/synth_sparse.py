"""The synthetic sparse workload (BASELINE configs[4]): B sparse LCQPs of ONE banded pattern, values from numpy's PCG64 seeded with
seed0 ^ instance id.  Used by bench.py (`--workload sparse` and the `sparse_config5` object of the default line) and, through
tests/problems.py, by the parity tests -- one definition of the workload.

Pattern: Q tridiagonal; row r of A touches the `span` (6) variables around 2 r (n = 2 nC); complementarity pairs L_i = e_{8 i},
R_i = e_{8 i + 4} (n = 8 nComp).  Values: Q = B'B + I with B upper bidiagonal (SPD), g in U(-1, 1), A values U(-1, 1) / sqrt(span),
bounds strictly feasible around a point x* that satisfies the complementarities."""
import numpy as np

SPARSE_SEED0 = 0x4C43515000000005


class CscPattern:
    """what lcqpow_amd.SparseBatchLCQP needs of a pattern: .indptr, .indices (CSC, sorted row indices), .shape"""

    def __init__(self, shape, indptr, indices):
        self.shape, self.indptr, self.indices = shape, np.ascontiguousarray(indptr, dtype=np.int32), np.ascontiguousarray(indices, dtype=np.int32)
        self.nnz = int(self.indptr[-1])


def _coo_to_csc(m, n, rows, cols):
    """CSC arrays of a pattern given as COO without duplicates, and the permutation COO entry -> CSC position"""
    order = np.lexsort((rows, cols))                   # by column, then row
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, cols + 1, 1)
    return CscPattern((m, n), np.cumsum(indptr), rows[order]), order


def _a_coords(n, nC, span):
    c0 = np.clip(2 * np.arange(nC) - 2, 0, n - span)
    return np.repeat(np.arange(nC), span), (c0[:, None] + np.arange(span)[None, :]).ravel()


def sparse_pattern_arrays(n=4096, nC=2048, nComp=512, span=6):
    """(Q pattern, stacked [A; L; R] pattern, order of Q's values, order of E's values): the orders map the natural value lists
    (Q: diagonal, sub-diagonal, super-diagonal; E: A row by row, L, R) to the CSC positions of the patterns"""
    assert n >= 2 * nC and n >= 8 * nComp and n >= max(8, span)
    qi = np.concatenate([np.arange(n), np.arange(1, n), np.arange(n - 1)])      # diagonal, sub-diagonal (i = j + 1), super-diagonal
    qj = np.concatenate([np.arange(n), np.arange(n - 1), np.arange(1, n)])
    Qp, qorder = _coo_to_csc(n, n, qi, qj)
    ai, aj = _a_coords(n, nC, span)
    ei = np.concatenate([ai, nC + np.arange(nComp), nC + nComp + np.arange(nComp)])
    ej = np.concatenate([aj, 8 * np.arange(nComp), 8 * np.arange(nComp) + 4])
    Ep, eorder = _coo_to_csc(nC + 2 * nComp, n, ei, ej)
    return Qp, Ep, qorder, eorder


def sparse_values(inst, n=4096, nC=2048, nComp=512, seed0=SPARSE_SEED0, span=6, orders=None):
    """Values of instance `inst` in the CSC order of sparse_pattern_arrays: dict(Qx, g, Ex, lbA, ubA).  `orders` = (qorder, eorder)
    of the pattern (computed when absent)."""
    if orders is None:
        orders = sparse_pattern_arrays(n, nC, nComp, span)[2:]
    qorder, eorder = orders
    rng = np.random.Generator(np.random.PCG64(seed0 ^ inst))
    a = rng.uniform(0.5, 1.5, n); bq = rng.uniform(-0.5, 0.5, n - 1)
    dq = a * a + 1.0; dq[1:] += bq * bq
    off = a[:-1] * bq
    g = rng.uniform(-1, 1, n)
    xs = rng.uniform(-1, 1, n)
    coin = rng.integers(0, 2, nComp)
    xs[8 * np.arange(nComp)] = np.where(coin == 0, 0.0, rng.uniform(0, 1, nComp))
    xs[8 * np.arange(nComp) + 4] = np.where(coin == 0, rng.uniform(0, 1, nComp), 0.0)
    av = rng.uniform(-1, 1, (nC, span)) / np.sqrt(float(span))
    ai, aj = _a_coords(n, nC, span)
    ax = _rowsum(av, xs[aj].reshape(nC, span))         # A x*, every row summed left to right
    lbA = ax - rng.uniform(0.1, 1.0, nC); ubA = ax + rng.uniform(0.1, 1.0, nC)
    Qx = np.concatenate([dq, off, off])[qorder]
    Ex = np.concatenate([av.ravel(), np.ones(2 * nComp)])[eorder]
    return dict(Qx=Qx, g=g, Ex=Ex, lbA=lbA, ubA=ubA)


def _rowsum(av, xv):
    """sum_k av[r, k] * xv[r, k], k ascending, one rounding per product and per addition (what a CSR matrix-vector product does)"""
    s = np.zeros(av.shape[0])
    for k in range(av.shape[1]):
        s = s + av[:, k] * xv[:, k]
    return s
