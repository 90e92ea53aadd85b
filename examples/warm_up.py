"""Python twin of examples/warm_up.cpp: the two-variable LCQP of the reference's warm-up example, through the
reference's own Python call sequence (import ... as lcqpow) on the MI355X backend.  Needs a GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcqpow_amd.lcqpow as lcqpow  # noqa: E402

Q = np.array([[2.0, 0.0], [0.0, 2.0]])
g = np.array([-2.0, -2.0])
L = np.array([[1.0, 0.0]])
R = np.array([[0.0, 1.0]])
x0 = np.array([1.0, 1.0])
y0 = np.zeros(4)

lcqp = lcqpow.LCQProblem(nV=2, nC=0, nComp=1)
options = lcqpow.Options()
options.setPrintLevel(lcqpow.PrintLevel.INNER_LOOP_ITERATES)
options.setQPSolver(lcqpow.QPSolver.HIP_DENSE)
lcqp.setOptions(options)

if lcqp.loadLCQP(Q=Q, g=g, L=L.T, R=R.T, x0=x0, y0=y0) != lcqpow.ReturnValue.SUCCESSFUL_RETURN:
    sys.exit("Failed to load LCQP.")
if lcqp.runSolver() != lcqpow.ReturnValue.SUCCESSFUL_RETURN:
    sys.exit("Failed to solve LCQP.")

stats = lcqpow.OutputStatistics()
lcqp.getOutputStatistics(stats)
print("xOpt = ", lcqp.getPrimalSolution())
print("yOpt = ", lcqp.getDualSolution())
print("i = ", stats.getIterTotal(), " k = ", stats.getIterOuter(), " rho = ", stats.getRhoOpt(), " WSR = ", stats.getSubproblemIter())
