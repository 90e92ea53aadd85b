"""Sparse arm of the oracle (oracle/lcqp_oracle_sparse.c: OSQP_SPARSE conventions of src/LCQProblem.cpp:929-960 over an ADMM-KKT +
polish subsolver) pinned against the dense oracle, which is itself pinned by the reference's tests (test_oracle_kat.py,
test_oracle_solver.py): on the same problem both arms must return the same solution; the sparse arm returns nC + 2 nComp duals
(no box part) with the qpOASES sign (src/SubsolverOSQP.cpp:196-199)."""
import numpy as np
import pytest
import scipy.sparse as sp

import problems as P


def _sparse_from_dense(oracle, d, opt):
    nV, nC, nK = d["nV"], d.get("nC", 0), d["nComp"]
    E = np.vstack([M for M in (d.get("A"), d["L"], d["R"]) if M is not None])
    kw = {k: d[k] for k in ("lbA", "ubA", "lbL", "ubL", "lbR", "ubR", "x0") if k in d}
    if "y0" in d:
        kw["y0"] = d["y0"][nV:]
    return oracle.sparse_lcqp_solve(nV, nC, nK, sp.csr_matrix(d["Q"]), d["g"], sp.csr_matrix(E), opt=opt, **kw)


@pytest.mark.parametrize("name", ["warm_up", "warm_up_binary", "circle"])
def test_sparse_arm_matches_dense_arm(oracle, name):
    d = P.circle(10) if name == "circle" else getattr(P, name)()
    opt = oracle.default_options(perturbStep=0)
    rd = P.oracle_solve(oracle, d, opt)
    rs = _sparse_from_dense(oracle, d, opt)
    assert rs["ret"] == rd["ret"] == 0
    assert np.abs(rs["x"] - rd["x"]).max() < 1e-9
    assert np.abs(rs["y"] - rd["y"][d["nV"]:]).max() < 1e-7                     # no box duals on this arm
    assert rs["stats"]["status"] == rd["stats"]["status"]
    if name == "warm_up":                                                       # test/RunUnitTests.cpp:537-546 without the box term
        x, y = rs["x"], rs["y"]
        assert np.abs(d["Q"] @ x + d["g"] - d["L"].T @ y[0:1] - d["R"].T @ y[1:2]).max() < 1e-9


def test_sparse_arm_run_warm_up_seeds(oracle):
    """SolverTest.RunWarmUp (test/RunUnitTests.cpp:505-551) on the sparse arm: both strongly stationary points are reached"""
    d = P.warm_up_x0()
    seen = set()
    for seed in range(40):
        r = _sparse_from_dense(oracle, d, oracle.default_options(perturbStep=1, perturbSeed=seed))
        assert r["ret"] == 0
        x = r["x"]
        assert min(np.abs(x - [1, 0]).max(), np.abs(x - [0, 1]).max()) < 2.2e-10
        seen.add(int(round(x[0])))
    assert seen == {0, 1}


@pytest.mark.parametrize("shape", [(64, 32, 8), (512, 256, 64)])
def test_sparse_synthetic_matches_dense_oracle(oracle, shape):
    """the banded synthetic workload of BASELINE configs[4] at sizes the dense oracle can take"""
    n, nC, nK = shape
    opt = oracle.default_options(perturbStep=0)
    for inst in range(3):
        d = P.sparse_instance(inst, n, nC, nK)
        rs = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        E = d["E"].toarray()
        rd = oracle.lcqp_solve(d["Q"].toarray(), d["g"], E[nC:nC + nK], E[nC + nK:], A=E[:nC], lbA=d["lbA"], ubA=d["ubA"], opt=opt, nV=n, nC=nC, nComp=nK)
        assert rs["ret"] == rd["ret"] == 0
        assert np.abs(rs["x"] - rd["x"]).max() < 1e-9 and np.abs(rs["y"] - rd["y"][n:]).max() < 1e-7
        assert rs["w"] <= 63


def test_sparse_full_size_properties(oracle):
    """n = 4096: properties that need no second solver -- stationarity of the returned (x, y) with the transformed duals,
    exact complementarity, feasibility; and the result does not depend on the KKT ordering beyond rounding"""
    n, nC, nK = 4096, 2048, 512
    d = P.sparse_instance(0, n, nC, nK)
    opt = oracle.default_options(perturbStep=0)
    Qc, Ec = d["Q"].tocsr(), d["E"].tocsr()
    r = oracle.sparse_lcqp_solve(n, nC, nK, Qc, d["g"], Ec, lbA=d["lbA"], ubA=d["ubA"], opt=opt)
    assert r["ret"] == 0 and r["stats"]["status"] in (1, 2, 3, 4)
    x, y = r["x"], r["y"]
    Lx, Rx = x[8 * np.arange(nK)], x[8 * np.arange(nK) + 4]
    assert (Lx * Rx).sum() < 2.2e-13 and Lx.min() > -1e-9 and Rx.min() > -1e-9
    ax = (Ec @ x)[:nC]
    assert (ax >= d["lbA"] - 1e-8).all() and (ax <= d["ubA"] + 1e-8).all()
    assert np.abs(Qc @ x + d["g"] - Ec.T @ y).max() < 1e-8                       # after transformDuals: Q x + g - [A;L;R]' y = 0
    # a different (worse) ordering: natural order of the KKT nodes sorted by their mean column
    pos = np.concatenate([np.arange(n, dtype=float), np.array([Ec.indices[Ec.indptr[i]:Ec.indptr[i + 1]].mean() + 0.5 for i in range(nC + 2 * nK)])])
    perm2 = np.argsort(pos, kind="stable").astype(np.int32)
    ip = np.empty_like(perm2); ip[perm2] = np.arange(perm2.size)
    coo = sp.bmat([[Qc, Ec.T], [Ec, None]]).tocoo()
    w2 = int(np.abs(ip[coo.row] - ip[coo.col]).max())
    r2 = oracle.sparse_lcqp_solve(n, nC, nK, Qc, d["g"], Ec, lbA=d["lbA"], ubA=d["ubA"], opt=opt, perm=perm2, w=w2)
    assert r2["ret"] == 0 and np.abs(r2["x"] - x).max() < 1e-9
