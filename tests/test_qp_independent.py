"""Independent-solver evidence for the subsolver boundary: tests/golden/qp_independent.npz holds QPs solved by
scipy (tools/make_qp_independent.py: SLSQP + trust-constr), i.e. by nothing in this repository.  The oracle's QP solver (CPU) and
the HIP subsolver (GPU) must reach the same minimiser / optimal value.  This is NOT qpOASES parity (unpinned: qpOASES is an absent
submodule, src/SubsolverQPOASES.cpp:152) -- it pins the contract: a KKT point of the convex QP."""
import os

import numpy as np
import pytest

import problems as P

FIX = np.load(os.path.join(P.GOLDEN, "qp_independent.npz"))


def _case(name):
    return {k: FIX[f"{name}_{k}"] for k in ("Q", "g", "A", "lbA", "ubA", "lb", "ub", "x", "obj")}


def _check(c, x, strictly_convex):
    obj = 0.5 * x @ c["Q"] @ x + c["g"] @ x
    ax = c["A"] @ x
    assert (ax >= c["lbA"] - 1e-7).all() and (ax <= c["ubA"] + 1e-7).all() and (x >= c["lb"] - 1e-7).all() and (x <= c["ub"] + 1e-7).all()
    assert obj <= float(c["obj"]) + 1e-6 * (1.0 + abs(float(c["obj"])))       # at least as good as the independent solver's point
    assert abs(obj - float(c["obj"])) <= 1e-5 * (1.0 + abs(float(c["obj"])))
    if strictly_convex:
        assert np.abs(x - c["x"]).max() < 1e-5                                # unique minimiser (scipy's accuracy is ~1e-7)


CASES = [f"convex{k}" for k in range(int(FIX["n_convex"]))] + ["circle", "example_data"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_qp_vs_independent_solver(oracle, name):
    c = _case(name)
    n = c["g"].size
    q = oracle.QP(c["Q"], c["A"])
    box = np.isfinite(c["lb"]).any() or np.isfinite(c["ub"]).any()
    ret, it, flag = q.solve(True, c["g"], c["lbA"], c["ubA"], np.zeros(n), None, c["lb"] if box else None, c["ub"] if box else None)
    assert ret == 0 and flag == 0
    x, y = q.solution()
    _check(c, x, name.startswith("convex"))
    # dual layout / sign of the boundary: Q x + g - A'y_A - y_box = 0
    assert np.abs(c["Q"] @ x + c["g"] - c["A"].T @ y[n:] - y[:n]).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_qp_vs_independent_solver(hip, name):
    c = _case(name)
    n, m = c["g"].size, c["A"].shape[0]
    q = hip.SubsolverHIP(n, m, c["Q"], c["A"])
    box = np.isfinite(c["lb"]).any() or np.isfinite(c["ub"]).any()
    ret, it, flag = q.solve(True, c["g"], c["lbA"], c["ubA"], np.zeros(n), None, c["lb"] if box else None, c["ub"] if box else None)
    assert ret == 0 and flag == 0
    x, y = q.getSolution()
    _check(c, x, name.startswith("convex"))
    assert np.abs(c["Q"] @ x + c["g"] - c["A"].T @ y[n:] - y[:n]).max() < 1e-8
    q.close()
