"""Independent answers at the BASELINE shape (n = 256, m = nC + 2 nComp = 640) for the QP subsolver boundary:
for instances 0..3 of the synthetic workload (SURVEY.md §8(d) generator; tests/oracle_py.py::synth_generate reproduces the instance, the
fixture stores no matrices) two QPs each are solved with scipy's trust-constr -- by nothing in this repository:
  qp0: the first, zero-penalty QP of the homotopy      min 1/2 x'Qx + g'x        s.t. lbA <= A x <= ubA, L x >= 0, R x >= 0
  qp1: a penalised QP                                   min 1/2 x'Qx + gk'x,      gk = g + rho C x_qp0, C = L'R + R'L, rho = 1
       (the linear term updateLinearization forms, src/LCQProblem.cpp:1105-1112)
Stored: tests/golden/qp_independent_baseline.npz = {b}_{qp}_{g, x, obj}.  Not qpOASES parity (unpinned, src/SubsolverQPOASES.cpp:152).
Takes about 7 minutes per QP; the four instances run in parallel.  Run in the build container:
    python tools/make_qp_independent_baseline.py"""
import os, sys
import numpy as np
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
N, NC, NK, RHO = 256, 512, 64, 1.0


def tc(Q, g, E, lbE, ubE, x0):
    from scipy.optimize import minimize, LinearConstraint
    f = lambda x: 0.5 * x @ Q @ x + g @ x
    r = minimize(f, x0, jac=lambda x: Q @ x + g, hess=lambda x: Q, method="trust-constr", constraints=LinearConstraint(E, lbE, ubE),
                 options=dict(xtol=1e-14, gtol=1e-12, barrier_tol=1e-14, maxiter=3000))
    return r.x, f(r.x)


def one(b):
    import oracle_py as O
    d = O.synth_generate(b, N, NC, NK)
    E = np.vstack([d["A"], d["L"], d["R"]])
    lbE = np.concatenate([d["lbA"], np.zeros(2 * NK)]); ubE = np.concatenate([d["ubA"], np.full(2 * NK, np.inf)])
    C = d["L"].T @ d["R"] + d["R"].T @ d["L"]
    x0, o0 = tc(d["Q"], d["g"], E, lbE, ubE, np.zeros(N))
    gk = d["g"] + RHO * (C @ x0)
    x1, o1 = tc(d["Q"], gk, E, lbE, ubE, x0)
    return b, d["g"], x0, o0, gk, x1, o1


if __name__ == "__main__":
    out = {"instances": np.arange(4), "rho": np.array(RHO)}
    with ProcessPoolExecutor(4) as ex:
        for b, g0, x0, o0, g1, x1, o1 in ex.map(one, range(4)):
            out[f"{b}_qp0_g"], out[f"{b}_qp0_x"], out[f"{b}_qp0_obj"] = g0, x0, np.array(o0)
            out[f"{b}_qp1_g"], out[f"{b}_qp1_x"], out[f"{b}_qp1_obj"] = g1, x1, np.array(o1)
            print("instance", b, "objectives", o0, o1, flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qp_independent_baseline.npz"), **out)
    print("wrote", len(out), "arrays")
