"""Independent answers for the QP subsolver boundary: the QP fixtures below are solved with scipy (SLSQP,
then trust-constr from its result), not with anything in this repository, and stored as tests/golden/qp_independent.npz.
This is independent-solver evidence for the contract of SubsolverBase::solve / getSolution (a KKT point of the convex QP) --
it is NOT qpOASES parity, which stays unpinned (src/SubsolverQPOASES.cpp:152: qpOASES is an absent submodule).
Run in the build container:  python tools/make_qp_independent.py"""
import os, sys
import numpy as np
from scipy.optimize import minimize, LinearConstraint, Bounds
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P


def solve(Q, g, A, lbA, ubA, lb, ub):
    n = g.size
    cons = []
    for i in range(A.shape[0]):
        a = A[i]
        if lbA[i] == ubA[i]:
            cons.append(dict(type="eq", fun=lambda x, a=a, b=lbA[i]: a @ x - b, jac=lambda x, a=a: a))
        else:
            if np.isfinite(lbA[i]): cons.append(dict(type="ineq", fun=lambda x, a=a, b=lbA[i]: a @ x - b, jac=lambda x, a=a: a))
            if np.isfinite(ubA[i]): cons.append(dict(type="ineq", fun=lambda x, a=a, b=ubA[i]: b - a @ x, jac=lambda x, a=a: -a))
    bnds = [(None if not np.isfinite(lo) else lo, None if not np.isfinite(hi) else hi) for lo, hi in zip(lb, ub)]
    f = lambda x: 0.5 * x @ Q @ x + g @ x
    df = lambda x: Q @ x + g
    r = minimize(f, np.zeros(n), jac=df, method="SLSQP", bounds=bnds, constraints=cons, options=dict(ftol=1e-15, maxiter=2000))
    lc = LinearConstraint(A, lbA, ubA) if A.shape[0] else ()
    r2 = minimize(f, r.x, jac=df, hess=lambda x: Q, method="trust-constr", bounds=Bounds(lb, ub), constraints=lc,
                  options=dict(xtol=1e-14, gtol=1e-12, barrier_tol=1e-14, maxiter=5000))
    x = r2.x if f(r2.x) <= f(r.x) + 1e-12 and (A.shape[0] == 0 or ((A @ r2.x >= lbA - 1e-9).all() and (A @ r2.x <= ubA + 1e-9).all())) else r.x
    return x, f(x)


out = {}
rng = np.random.default_rng(20261002)
k = 0
for n, m, neq in ((8, 6, 2), (16, 20, 4), (32, 40, 6), (64, 80, 10), (24, 30, 0), (48, 20, 8)):
    M = rng.standard_normal((n, n)); Q = M.T @ M / n + np.eye(n)             # strictly convex
    g = rng.standard_normal(n)
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    xs = rng.standard_normal(n)
    ax = A @ xs
    lbA = ax - rng.uniform(0.05, 1.0, m); ubA = ax + rng.uniform(0.05, 1.0, m)
    lbA[:neq] = ubA[:neq] = ax[:neq]                                         # equalities
    one = rng.random(m) < 0.3; one[:neq] = False
    ubA[one] = np.inf                                                        # one-sided rows
    lb = np.full(n, -np.inf); ub = np.full(n, np.inf)
    bx = rng.random(n) < 0.4
    lb[bx] = xs[bx] - rng.uniform(0.0, 0.5, bx.sum()); ub[bx] = xs[bx] + rng.uniform(0.0, 0.5, bx.sum())
    x, obj = solve(Q, g, A, lbA, ubA, lb, ub)
    for nm, v in (("Q", Q), ("g", g), ("A", A), ("lbA", lbA), ("ubA", ubA), ("lb", lb), ("ub", ub), ("x", x), ("obj", np.array(obj))):
        out[f"convex{k}_{nm}"] = v
    k += 1
out["n_convex"] = np.array(k)
# the first (zero-penalty) QPs of two reference examples with PSD-only Hessians: minimisers are not unique in the flat directions,
# so only the optimal value and feasibility are pinned
for name in ("circle", "example_data"):
    d = P.circle(20) if name == "circle" else P.example_data()
    E = np.vstack([d["A"], d["L"], d["R"]])
    nC, nK = d["nC"], d["nComp"]
    lbE = np.concatenate([d["lbA"], np.zeros(2 * nK)]); ubE = np.concatenate([d["ubA"], np.full(2 * nK, np.inf)])
    lb = d.get("lb", np.full(d["nV"], -np.inf)); ub = d.get("ub", np.full(d["nV"], np.inf))
    if lb is None: lb = np.full(d["nV"], -np.inf)
    if ub is None: ub = np.full(d["nV"], np.inf)
    x, obj = solve(d["Q"], d["g"], E, lbE, ubE, lb, ub)
    for nm, v in (("Q", d["Q"]), ("g", d["g"]), ("A", E), ("lbA", lbE), ("ubA", ubE), ("lb", lb), ("ub", ub), ("x", x), ("obj", np.array(obj))):
        out[f"{name}_{nm}"] = v
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qp_independent.npz"), **out)
print("wrote", len(out), "arrays")
