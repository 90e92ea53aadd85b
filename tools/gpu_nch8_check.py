import sys, numpy as np
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests"); sys.path.insert(0,"/root/repo/tools")
from gpu_ab import load_variant
import oracle_py as O
m = load_variant("v", sys.argv[1])
for (n, nC, nComp) in ((513, 0, 50), (600, 300, 100), (1024, 600, 256)):
    B = 2
    bt = m.BatchLCQP(B, n, nC, nComp, opt=m.default_options(perturbStep=0))
    bt.generate_synthetic(0); bt.run()
    x, y, st = bt.solution()
    for b in range(B):
        d = bt.read_problem(b)
        ro = O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], A=d["A"] if nC else None, lbA=d["lbA"] if nC else None, ubA=d["ubA"] if nC else None, opt=O.default_options(perturbStep=0), nV=n, nC=nC, nComp=nComp)
        print(sys.argv[1], (n, nC, nComp), b, "ret", st[b]["returnValue"], ro["ret"], "iters", st[b]["iterTotal"], ro["stats"]["iterTotal"], "trials", st[b]["trials"], ro["stats"]["trials"], "dx %.2e" % np.abs(ro["x"] - x[b]).max(), "timing", bt.last_timing())
    bt.close()
