"""Python twin of examples/solve_lcqp_from_file.cpp (the reference's interfaces/python/examples/solve_lcqp_from_file.py): an LCQP from a
directory of text files (Q.txt, g.txt, L.txt, R.txt and whichever of lbL ubL lbR ubR A lbA ubA lb ub x0 y0 exist), through the file
overload of loadLCQP with the reference's keyword names.      python examples/solve_lcqp_from_file.py <directory>      Needs a GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcqpow_amd.lcqpow as lcqpow  # noqa: E402

if len(sys.argv) < 2:
    sys.exit(__doc__)
d = sys.argv[1]
count = lambda name: sum(1 for l in open(os.path.join(d, name + ".txt")) if l.strip()) if os.path.exists(os.path.join(d, name + ".txt")) else 0
nV = count("g")
if nV == 0 or count("L") % nV or count("L") == 0:
    sys.exit(d + ": need Q.txt, g.txt, L.txt, R.txt with matching sizes")
nComp, nC = count("L") // nV, count("A") // nV
print("LCQP from %s: nV = %d, nC = %d, nComp = %d" % (d, nV, nC, nComp))
files = {k + "_file": os.path.join(d, k + ".txt") for k in ("Q", "g", "L", "R", "lbL", "ubL", "lbR", "ubR", "A", "lbA", "ubA", "lb", "ub", "x0", "y0")
         if count(k) > 0}

lcqp = lcqpow.LCQProblem(nV=nV, nC=nC, nComp=nComp)
options = lcqpow.Options()
options.setPrintLevel(lcqpow.PrintLevel.OUTER_LOOP_ITERATES)
lcqp.setOptions(options)
if lcqp.loadLCQP(**files) != lcqpow.ReturnValue.SUCCESSFUL_RETURN:
    sys.exit("Failed to load LCQP.")
if lcqp.runSolver() != lcqpow.ReturnValue.SUCCESSFUL_RETURN:
    sys.exit("Failed to solve LCQP.")
stats = lcqpow.OutputStatistics()
lcqp.getOutputStatistics(stats)
print("i = ", stats.getIterTotal(), " k = ", stats.getIterOuter(), " rho = ", stats.getRhoOpt(), " WSR = ", stats.getSubproblemIter())
print("xOpt = ", lcqp.getPrimalSolution())
