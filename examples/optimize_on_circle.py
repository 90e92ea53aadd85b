"""Python twin of examples/optimize_on_circle.cpp (the reference's interfaces/python/examples/OptimizeOnCircle.py and
OptimizeOnCircleStoreSteps.py): the point of the (polygonal) unit circle closest to x_ref in the norm of Q = [17 -15; -15 17], with the reference's
Python call sequence on the MI355X backend.      python examples/optimize_on_circle.py [N] [--sparse] [--store-steps]      Needs a GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcqpow_amd.lcqpow as lcqpow  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 100
sparse, store = "--sparse" in sys.argv, "--store-steps" in sys.argv
nV, nC, nComp = 2 + 2 * N, N + 1, N
Qx = np.array([[17.0, -15.0], [-15.0, 17.0]])
x_ref = np.array([0.5, -0.6])
Q = np.zeros((nV, nV)); Q[:2, :2] = Qx
Q[np.arange(2, nV), np.arange(2, nV)] = 5e-12
g = np.zeros(nV); g[:2] = -Qx @ x_ref
A = np.zeros((nC, nV)); L = np.zeros((nComp, nV)); R = np.zeros((nComp, nV))
t = 2 * np.pi * np.arange(N) / N
A[np.arange(N), 0], A[np.arange(N), 1] = np.cos(t), np.sin(t)
A[np.arange(N), 2 + 2 * np.arange(N)] = 1.0          # tangent i and its slack s_i
A[N, 3 + 2 * np.arange(N)] = 1.0                     # the selectors z_i sum to one
L[np.arange(N), 2 + 2 * np.arange(N)] = 1.0          # 0 <= s_i _|_ z_i >= 0
R[np.arange(N), 3 + 2 * np.arange(N)] = 1.0
lbA = ubA = np.ones(nC)
x0 = np.ones(nV); x0[:2] = x_ref


def csc(M):
    """dense array -> lcqpow.cscWrapper (compressed sparse columns, the fields of OSQP's csc)"""
    cols = [np.nonzero(M[:, c])[0] for c in range(M.shape[1])]
    p = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(int)
    i = np.concatenate(cols).astype(int) if p[-1] else np.zeros(0, dtype=int)
    x = np.concatenate([M[c_, k] for k, c_ in enumerate(cols)]) if p[-1] else np.zeros(0)
    return lcqpow.cscWrapper(M.shape[0], M.shape[1], int(p[-1]), x.astype(float), list(i), list(p))


lcqp = lcqpow.LCQProblem(nV=nV, nC=nC, nComp=nComp)
options = lcqpow.Options()
options.setPrintLevel(lcqpow.PrintLevel.OUTER_LOOP_ITERATES)
options.setStoreSteps(store)
if sparse:
    options.setQPSolver(lcqpow.QPSolver.OSQP_SPARSE)      # runs on this backend's sparse engine
lcqp.setOptions(options)
if sparse:
    rc = lcqp.loadLCQP(Q=csc(Q), g=g, L=csc(L), R=csc(R), A=csc(A), lbA=lbA, ubA=ubA, x0=x0)
else:
    rc = lcqp.loadLCQP(Q=Q, g=g, L=L.T, R=R.T, A=A.T, lbA=lbA, ubA=ubA, x0=x0)      # dense matrices as the reference's binding takes them
if rc != lcqpow.ReturnValue.SUCCESSFUL_RETURN:
    sys.exit("Failed to load LCQP.")
if lcqp.runSolver() != lcqpow.ReturnValue.SUCCESSFUL_RETURN:
    sys.exit("Failed to solve LCQP.")

stats = lcqpow.OutputStatistics()
lcqp.getOutputStatistics(stats)
x = lcqp.getPrimalSolution()
print("xOpt = [%.6f, %.6f], |xOpt| = %.6f" % (x[0], x[1], np.hypot(x[0], x[1])))
print("i = ", stats.getIterTotal(), " k = ", stats.getIterOuter(), " rho = ", stats.getRhoOpt(), " WSR = ", stats.getSubproblemIter())
if store:
    print("per iterate: complementarity", ["%.2e" % v for v in stats.getPhiVals()])
    print("             step length    ", ["%.3f" % v for v in stats.getStepLength()])
