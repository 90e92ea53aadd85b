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


# ---- a second sparse workload: a pattern that is neither banded nor bordered (round 6: the general sparse LDL') --------------------------------
GRID_SEED0 = 0x4C43515000000006


def grid_pattern_arrays(g=128, nC=800, nComp=1200):
    """B sparse LCQPs whose KKT graph is a g x g grid: Q a 5-point stencil, row r of A couples two vertically adjacent cells, complementarity
    between horizontally adjacent cells (cells (r, 2c), (r, 2c + 1)); which cells carry rows is fixed by the pattern seed, the values differ per
    instance.  Returns (Q pattern, stacked [A; L; R] pattern, qorder, eorder, info) with info = the cell indices the value generator needs."""
    n = g * g
    rng = np.random.Generator(np.random.PCG64(GRID_SEED0))
    idx = lambda r, c: r * g + c
    rr, cc = np.divmod(np.arange(n), g)
    right = np.flatnonzero(cc + 1 < g); down = np.flatnonzero(rr + 1 < g)
    qi = np.concatenate([np.arange(n), right + 1, right, down + g, down])      # diagonal; (i+1, i), (i, i+1); (i+g, i), (i, i+g)
    qj = np.concatenate([np.arange(n), right, right + 1, down, down + g])
    Qp, qorder = _coo_to_csc(n, n, qi, qj)
    cells = np.array([idx(r, c) for r in range(g) for c in range(0, g - 1, 2)])
    assert nComp <= len(cells)
    li = np.sort(rng.choice(cells, nComp, replace=False))                       # L_k = e_{li[k]}, R_k = e_{li[k] + 1}
    ar = rng.integers(0, g - 1, nC); ac = rng.integers(0, g, nC)
    a0 = ar * g + ac                                                            # A_k touches a0[k] and a0[k] + g
    ei = np.concatenate([np.arange(nC), np.arange(nC), nC + np.arange(nComp), nC + nComp + np.arange(nComp)])
    ej = np.concatenate([a0, a0 + g, li, li + 1])
    Ep, eorder = _coo_to_csc(nC + 2 * nComp, n, ei, ej)
    return Qp, Ep, qorder, eorder, dict(g=g, n=n, nC=nC, nComp=nComp, right=right, down=down, li=li, a0=a0)


def grid_values(inst, info, orders, seed0=GRID_SEED0):
    """Values of instance `inst` of the grid workload in the CSC order of grid_pattern_arrays: dict(Qx, g, Ex, lbA, ubA)"""
    qorder, eorder = orders
    g, n, nC, nK = info["g"], info["n"], info["nC"], info["nComp"]
    rng = np.random.Generator(np.random.PCG64(seed0 ^ (inst + 1)))
    dq = 4.5 + rng.uniform(0, 1, n)
    offr = -rng.uniform(0.8, 1.2, len(info["right"])); offd = -rng.uniform(0.8, 1.2, len(info["down"]))      # diagonally dominant: positive definite
    xs = rng.uniform(-1, 1, n)
    coin = rng.integers(0, 2, nK)
    xs[info["li"]] = np.where(coin == 0, 0.0, rng.uniform(0, 1, nK))
    xs[info["li"] + 1] = np.where(coin == 0, rng.uniform(0, 1, nK), 0.0)
    a1 = rng.uniform(0.5, 1.5, nC); a2 = rng.uniform(-1.5, -0.5, nC)
    ax = a1 * xs[info["a0"]] + a2 * xs[info["a0"] + g]
    Qx = np.concatenate([dq, offr, offr, offd, offd])[qorder]
    Ex = np.concatenate([a1, a2, np.ones(2 * nK)])[eorder]
    return dict(Qx=Qx, g=rng.uniform(-2, 2, n), Ex=Ex, lbA=ax - rng.uniform(0.1, 1.0, nC), ubA=ax + rng.uniform(0.1, 1.0, nC))
