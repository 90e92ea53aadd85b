"""Problem instances used by the parity tests.

The small ones restate the data of the reference's own tests/examples (file:line cited); the
example_data fixture is the reference's data directory stored as tests/golden/example_data.npz by
tools/make_golden.py (data only, no reference source)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INF = np.inf


def warm_up():
    """examples/warm_up.cpp:32-42, test/RunUnitTests.cpp:505-512"""
    return dict(Q=2 * np.eye(2), g=np.array([-2., -2.]), L=np.array([[1., 0.]]), R=np.array([[0., 1.]]),
                nV=2, nC=0, nComp=1)


def warm_up_x0():
    d = warm_up()
    d.update(x0=np.array([1., 1.]), y0=np.zeros(4))
    return d


def warm_up_w_A():
    """test/examples/warm_up_w_A.cpp:32-41"""
    d = warm_up()
    d.update(A=np.array([[1., -1.]]), lbA=np.array([-0.5]), ubA=np.array([INF]), nC=1)
    return d


def warm_up_binary():
    """test/examples/warm_up_binary.cpp:32-46"""
    return dict(Q=2 * np.eye(2), g=np.array([-2., -2.]), L=np.array([[1., 0.], [1., 0.]]),
                R=np.array([[0., 1.], [-1., 0.]]), lbL=np.zeros(2), lbR=np.array([0., -0.5]), x0=np.zeros(2),
                nV=2, nC=0, nComp=2)


def infeasible():
    """test/RunUnitTests.cpp:463-480: 0 <= x1 <= -1"""
    d = warm_up()
    d.update(A=np.array([[1., 0.]]), lbA=np.array([0.]), ubA=np.array([-1.]), nC=1)
    return d


def circle(N=100):
    """examples/OptimizeOnCircle.cpp:32-99"""
    nV = 2 + 2 * N; nC = N + 1; nComp = N
    Q = np.zeros((nV, nV)); Q[0, 0] = Q[1, 1] = 17; Q[0, 1] = Q[1, 0] = -15
    for i in range(2, nV):
        Q[i, i] = 5e-12
    xr = np.array([0.5, -0.6])
    g = np.zeros(nV); g[:2] = -(np.array([[17., -15.], [-15., 17.]]) @ xr)
    A = np.zeros((nC, nV)); L = np.zeros((nComp, nV)); R = np.zeros((nComp, nV)); x0 = np.zeros(nV); x0[:2] = xr
    for i in range(N):
        A[i, 0] = np.cos(2 * np.pi * i / N); A[i, 1] = np.sin(2 * np.pi * i / N); A[i, 2 + 2 * i] = 1
        A[N, 3 + 2 * i] = 1; L[i, 2 + 2 * i] = 1; R[i, 3 + 2 * i] = 1; x0[2 * i + 2] = 1; x0[2 * i + 3] = 1
    return dict(Q=Q, g=g, L=L, R=R, A=A, lbA=np.ones(nC), ubA=np.ones(nC), x0=x0, nV=nV, nC=nC, nComp=nComp)


def example_data():
    """examples/example_data/*.txt (nV=151, nC=50, nComp=100), loaded as solve_lcqp_from_file.cpp:63-97 does"""
    z = np.load(os.path.join(GOLDEN, "example_data.npz"))
    g = z["g"]; n = g.size; nComp = z["lbL"].size; nC = z["lbA"].size
    return dict(Q=z["Q"].reshape(n, n), g=g, L=z["L"].reshape(nComp, n), R=z["R"].reshape(nComp, n),
                A=z["A"].reshape(nC, n), lbA=z["lbA"], ubA=z["ubA"], lbL=z["lbL"], ubL=z["ubL"], lbR=z["lbR"],
                ubR=z["ubR"], lb=z["lb"], ub=z["ub"], x0=z["x0"], nV=n, nC=nC, nComp=nComp)


KEYS = ("lbL", "ubL", "lbR", "ubR", "A", "lbA", "ubA", "lb", "ub", "x0", "y0")


def oracle_solve(O, d, opt, trace=0):
    kw = {k: d.get(k) for k in KEYS}
    return O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], opt=opt, trace=trace, nV=d["nV"], nC=d["nC"], nComp=d["nComp"], **kw)


def hip_solve(la, d, opt, device=0, trace=False):
    """one-instance batch through the C ABI"""
    with_box = d.get("lb") is not None or d.get("ub") is not None
    bt = la.BatchLCQP(1, d["nV"], d["nC"], d["nComp"], with_box=with_box, device=device, opt=opt)
    kw = {k: d.get(k) for k in KEYS}
    rc = bt.load(0, 1, d["Q"], d["g"], d["L"], d["R"], **kw)
    if rc != 0:
        bt.close()
        return dict(ret=rc, x=None, y=None, stats=None)
    bt.run()
    x, y, st = bt.solution()
    out = dict(ret=st[0]["returnValue"], x=x[0], y=y[0], stats=st[0])
    if trace:
        out["trace_scalars"], out["trace_x"] = bt.trace(0)
    bt.close()
    return out


def kkt_residuals(Q, g, A, lbA, ubA, lb, ub, x, y):
    """KKT residuals of min 1/2x'Qx+g'x s.t. lbA<=Ax<=ubA, lb<=x<=ub with the qpOASES dual layout/sign
    (y[0:n] box, y[n:] rows; Qx+g-A'yA-yB=0; y>0 at lower bounds, y<0 at upper bounds)."""
    n = Q.shape[0]
    yB, yA = y[:n], y[n:]
    stat = np.abs(Q @ x + g - A.T @ yA - yB).max()
    Ax = A @ x
    pf = max(np.maximum(lbA - Ax, Ax - ubA).max(initial=0.0), np.maximum(lb - x, x - ub).max(initial=0.0), 0.0)
    cs = 0.0
    for yy, v, lo, hi in ((yA, Ax, lbA, ubA), (yB, x, lb, ub)):
        with np.errstate(invalid="ignore"):
            dl = np.where(np.isfinite(lo), v - lo, np.inf); du = np.where(np.isfinite(hi), hi - v, np.inf)
            cs = max(cs, np.abs(np.where(yy > 0, yy * dl, 0.0)).max(initial=0.0))
            cs = max(cs, np.abs(np.where(yy < 0, yy * du, 0.0)).max(initial=0.0))
    return stat, pf, cs


def certificate_qps(seed=3, n=24, m=30):
    """two QPs for the certificates of the subsolver: (infeasible) a feasible polytope plus two rows that contradict each
    other, x_0 + x_1 >= 1 and x_0 + x_1 <= -1, on different rows so that no bound pair is inconsistent by itself;
    (unbounded) a singular Hessian whose null space holds a descent direction that no constraint blocks."""
    r = np.random.default_rng(seed)
    M = r.standard_normal((n, n)); Q = M.T @ M / n + np.eye(n)
    A = r.standard_normal((m, n)) / np.sqrt(n); xs = r.standard_normal(n)
    lbA = A @ xs - r.uniform(0.1, 1, m); ubA = A @ xs + r.uniform(0.1, 1, m)
    A[0] = 0; A[0, 0] = A[0, 1] = 1; lbA[0] = 1.0; ubA[0] = np.inf
    A[1] = 0; A[1, 0] = A[1, 1] = 1; lbA[1] = -np.inf; ubA[1] = -1.0
    infeasible = dict(Q=Q, A=A, g=r.standard_normal(n), lbA=lbA, ubA=ubA)
    Q2 = Q.copy(); Q2[:, 0] = 0; Q2[0, :] = 0                       # x_0 does not enter the objective's quadratic part
    A2 = r.standard_normal((m, n)) / np.sqrt(n); A2[:, 0] = 0        # ... nor any constraint
    g2 = r.standard_normal(n); g2[0] = 1.0                          # ... and decreasing it lowers the objective without bound
    unbounded = dict(Q=Q2, A=A2, g=g2, lbA=A2 @ xs - 1.0, ubA=A2 @ xs + 1.0)
    return infeasible, unbounded


def branch_qp_solution(O, d, x, tol=1e-7):
    """Independent check of an LCQP solution: fix the complementarity branch x sits on (per pair the side that is at its lower
    bound becomes an equality, the other keeps its bounds) and solve the resulting convex QP with the oracle's QP solver.  A
    strongly stationary point of the LCQP is the minimiser of that branch QP when biactive pairs (both sides at their bounds)
    keep both sides as inequalities -- strong stationarity says exactly that both multipliers are non-negative there.
    Returns None when x is not complementary or the branch QP cannot be solved."""
    n, nC, nComp = d["nV"], d["nC"], d["nComp"]
    lbL = d.get("lbL", np.zeros(nComp)); lbR = d.get("lbR", np.zeros(nComp))
    ubL = d.get("ubL", np.full(nComp, INF)); ubR = d.get("ubR", np.full(nComp, INF))
    Lx, Rx = d["L"] @ x - lbL, d["R"] @ x - lbR
    loL, hiL, loR, hiR = lbL.copy(), ubL.copy(), lbR.copy(), ubR.copy()
    for i in range(nComp):
        if Lx[i] <= tol and Rx[i] <= tol:
            continue
        if Lx[i] <= tol:
            hiL[i] = lbL[i]
        elif Rx[i] <= tol:
            hiR[i] = lbR[i]
        else:
            return None
    A = d["A"] if nC else np.zeros((0, n))
    E = np.vstack([A, d["L"], d["R"]])
    lo = np.concatenate([d["lbA"] if nC else [], loL, loR]); hi = np.concatenate([d["ubA"] if nC else [], hiL, hiR])
    q = O.QP(d["Q"], E)
    ret, it, ef = q.solve(True, d["g"], lo, hi, np.array(x, dtype=float), None, d.get("lb"), d.get("ub"))
    if ret != 0:
        return None
    return q.solution()[0]


def stationarity_type(d, x, y, rho, ctol=1e3 * 2.221e-16, merge_box=False):
    """determineStationarityType + getWeakComplementarities (src/LCQProblem.cpp:1412-1482) restated in numpy on a RETURNED solution:
    y holds the transformed duals (transformDuals :1381-1409), so the multipliers of L and R are shifted back by rho R x / rho L x first.
    Returns 1 (W), 2 (C), 3 (M), 4 (S).
    merge_box: the reference's rule reads the signs of the multipliers a solver happened to return, but a complementarity row c e_v on a variable
    whose box bound is active has the box bound's normal (up to sign): the two multipliers are not unique, only their combined contribution
    c y_row + y_box to the stationarity in v is.  With merge_box the rule is applied to the most favourable admissible split -- which is what
    "there EXIST multipliers with these signs" (the definition of the stationarity types) means:
      same direction  (c > 0 at an active lower bound, c < 0 at an active upper bound): the row can carry the whole contribution, y_row + y_box / c;
      opposite direction (the two constraints pin x_v from both sides): any amount can be added to both, the row's multiplier is unbounded above."""
    n, nC, nK = d["nV"], d["nC"], d["nComp"]
    L, R = d["L"], d["R"]
    Lx, Rx = L @ x, R @ x
    yL = y[n + nC:n + nC + nK] + rho * Rx
    yR = y[n + nC + nK:] + rho * Lx
    if merge_box:
        lb = d.get("lb") if d.get("lb") is not None else np.full(n, -INF)
        ub = d.get("ub") if d.get("ub") is not None else np.full(n, INF)
        yL, yR = yL.copy(), yR.copy()
        for M_, ym in ((L, yL), (R, yR)):
            for i in range(nK):
                nz = np.nonzero(M_[i])[0]
                if nz.size != 1:
                    continue
                v, c = nz[0], M_[i, nz[0]]
                at_lo, at_hi = abs(x[v] - lb[v]) <= 1e-9, abs(x[v] - ub[v]) <= 1e-9
                if (c > 0 and at_hi) or (c < 0 and at_lo):
                    ym[i] = INF
                elif (c > 0 and at_lo) or (c < 0 and at_hi):
                    ym[i] += y[v] / c
    sflag, mflag = True, True
    for i in range(nK):
        if not (Lx[i] <= ctol and Rx[i] <= ctol):
            continue
        a, b = yL[i], yR[i]
        prod, mn = (0.0 if (a == 0 or b == 0) else a * b), min(a, b)
        if mn < 0:
            sflag = False
        if abs(prod) >= ctol and mn <= 0:
            if prod <= ctol:
                return 1
            mflag = False
    return 4 if sflag else (3 if mflag else 2)


def grid_lcqp(g=44, nK=300, nC=200, seed=0):
    """An LCQP whose KKT graph is a 2-D grid: g x g variables coupled by a 5-point stencil Hessian, complementarity between horizontally
    adjacent cells, constraint rows that couple vertically adjacent cells.  Reverse Cuthill-McKee leaves a band of half width ~ 2 g, far beyond
    the 63 the sparse engine's lane groups cover, and no handful of border nodes fixes that: the pattern that is "neither banded nor
    bordered".  Returns scipy CSC matrices Q (n x n), E = [A; L; R] and the dense vectors."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = g * g
    idx = lambda r, c: r * g + c
    rows, cols, vals = [], [], []
    for r in range(g):
        for c in range(g):
            i = idx(r, c)
            rows.append(i); cols.append(i); vals.append(4.5 + rng.uniform(0, 1))
            for dr, dc in ((0, 1), (1, 0)):
                if r + dr < g and c + dc < g:
                    j = idx(r + dr, c + dc)
                    rows += [i, j]; cols += [j, i]; vals += [-1.0, -1.0]
    Q = sp.csc_matrix((vals, (rows, cols)), shape=(n, n))
    cells = [(r, c) for r in range(g) for c in range(0, g - 1, 2)]
    pick = rng.choice(len(cells), nK, replace=False)
    Lr, Lc, Rr, Rc = [], [], [], []
    xs = rng.uniform(-1, 1, n)
    for k, p in enumerate(pick):
        r, c = cells[p]
        i, j = idx(r, c), idx(r, c + 1)
        Lr.append(k); Lc.append(i); Rr.append(k); Rc.append(j)
        if rng.random() < 0.5: xs[i], xs[j] = 0.0, rng.uniform(0, 1)
        else: xs[i], xs[j] = rng.uniform(0, 1), 0.0
    Lm = sp.csc_matrix((np.ones(nK), (Lr, Lc)), shape=(nK, n)); Rm = sp.csc_matrix((np.ones(nK), (Rr, Rc)), shape=(nK, n))
    ar, ac, av = [], [], []
    for k in range(nC):
        r, c = int(rng.integers(0, g - 1)), int(rng.integers(0, g))
        ar += [k, k]; ac += [idx(r, c), idx(r + 1, c)]; av += [rng.uniform(0.5, 1.5), rng.uniform(-1.5, -0.5)]
    A = sp.csc_matrix((av, (ar, ac)), shape=(nC, n))
    ax = A @ xs
    return dict(nV=n, nC=nC, nComp=nK, Q=Q, g=rng.uniform(-2, 2, n), A=A, L=Lm, R=Rm, E=sp.vstack([A, Lm, Rm]).tocsc(),
                lbA=ax - rng.uniform(0.1, 1, nC), ubA=ax + rng.uniform(0.1, 1, nC))


# ---- sparse synthetic workload (BASELINE config 5): banded, OCP-like, one pattern for the whole batch ------------------------------
from lcqpow_amd.synth_sparse import SPARSE_SEED0  # noqa: E402


def sparse_pattern(n=4096, nC=2048, nComp=512, span=6):
    """Pattern of the sparse synthetic LCQPs (lcqpow_amd/synth_sparse.py, the one definition of the workload) as scipy CSC matrices
    of ones: (Q, stacked [A; L; R])."""
    import scipy.sparse as sp
    from lcqpow_amd import synth_sparse as S
    Qp, Ep, _, _ = S.sparse_pattern_arrays(n, nC, nComp, span)
    mk = lambda pt: sp.csc_matrix((np.ones(pt.nnz), pt.indices, pt.indptr), shape=pt.shape)
    return mk(Qp), mk(Ep)


def sparse_instance(inst, n=4096, nC=2048, nComp=512, seed0=SPARSE_SEED0, span=6):
    """Instance `inst` of the sparse synthetic workload (lcqpow_amd/synth_sparse.py) with scipy matrices:
    dict(Q, E (CSC with values), g, lbA, ubA, nV, nC, nComp)."""
    import scipy.sparse as sp
    from lcqpow_amd import synth_sparse as S
    Qp, Ep, qo, eo = S.sparse_pattern_arrays(n, nC, nComp, span)
    v = S.sparse_values(inst, n, nC, nComp, seed0=seed0, span=span, orders=(qo, eo))
    Q = sp.csc_matrix((v["Qx"], Qp.indices, Qp.indptr), shape=Qp.shape)
    E = sp.csc_matrix((v["Ex"], Ep.indices, Ep.indptr), shape=Ep.shape)
    return dict(Q=Q, E=E, g=v["g"], lbA=v["lbA"], ubA=v["ubA"], nV=n, nC=nC, nComp=nComp)


def lcqp_kkt_residuals(d, x, y, rho):
    """Independent of every solver in this repository: the first-order conditions of the LCQP itself, evaluated with numpy from the
    problem data, at a returned point (x, y) with y in the reference's layout [box (nV), A (nC), L (nComp), R (nComp)] AFTER transformDuals
    (src/LCQProblem.cpp:1381-1409: y_L -= rho R x, y_R -= rho L x -- without the shift by lbR / lbL, which is added back here).
    Returns (stationarity, primal infeasibility, complementarity |sum (Lx - lbL)(Rx - lbR)|, largest wrong-signed multiplier of an
    inequality row of A or a box bound)."""
    n, nC, nComp = d["nV"], d["nC"], d["nComp"]
    Q, g, L, R = d["Q"], d["g"], d["L"], d["R"]
    A = d.get("A") if d.get("A") is not None else np.zeros((0, n))
    lbL = d.get("lbL") if d.get("lbL") is not None else np.zeros(nComp)
    lbR = d.get("lbR") if d.get("lbR") is not None else np.zeros(nComp)
    ubL = d.get("ubL") if d.get("ubL") is not None else np.full(nComp, INF)
    ubR = d.get("ubR") if d.get("ubR") is not None else np.full(nComp, INF)
    lbA = d.get("lbA") if d.get("lbA") is not None else np.full(nC, -INF)
    ubA = d.get("ubA") if d.get("ubA") is not None else np.full(nC, INF)
    lb = d.get("lb") if d.get("lb") is not None else np.full(n, -INF)
    ub = d.get("ub") if d.get("ub") is not None else np.full(n, INF)
    yB, yA, yL, yR = y[:n], y[n:n + nC], y[n + nC:n + nC + nComp], y[n + nC + nComp:]
    stat = np.abs(Q @ x + g - A.T @ yA - L.T @ (yL + rho * lbR) - R.T @ (yR + rho * lbL) - yB).max()
    Ax, Lx, Rx = A @ x, L @ x, R @ x
    feas = max(np.maximum(lbA - Ax, Ax - ubA).max(initial=0.0), np.maximum(lb - x, x - ub).max(initial=0.0),
               np.maximum(lbL - Lx, Lx - ubL).max(initial=0.0), np.maximum(lbR - Rx, Rx - ubR).max(initial=0.0), 0.0)
    compl = abs(float((Lx - lbL) @ (Rx - lbR)))
    sign = 0.0
    for yy, v, lo, hi in ((yA, Ax, lbA, ubA), (yB, x, lb, ub)):
        tol = 1e-7 * (1.0 + np.abs(v))
        with np.errstate(invalid="ignore"):
            at_lo = np.isfinite(lo) & (v - lo <= tol); at_hi = np.isfinite(hi) & (hi - v <= tol)
        sign = max(sign, np.where(~at_lo, np.maximum(yy, 0.0), 0.0).max(initial=0.0), np.where(~at_hi, np.maximum(-yy, 0.0), 0.0).max(initial=0.0))
    return float(stat), float(feas), compl, float(sign)
