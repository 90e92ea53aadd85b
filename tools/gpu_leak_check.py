"""Create / load / run / destroy many batch, QP and CSC objects and watch the free device memory (hipMemGetInfo via torch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import lcqpow_amd as la, lcqpow_amd.lcqpow as lcqpow, problems as P
d = P.circle(20)
free0 = None
for k in range(151):
    if k == 1:
        free0 = torch.cuda.mem_get_info()[0]      # after the first cycle: the runtime's one-time scratch / code-object allocations are in
    bt = la.BatchLCQP(16, 64, 96, 16, opt=la.default_options(perturbStep=0)); bt.generate_synthetic(k); bt.run(); bt.solution(); bt.rerun_failed(); bt.close()
    q = la.SubsolverHIP(d["nV"], d["nC"] + 2 * d["nComp"], d["Q"], np.vstack([d["A"], d["L"], d["R"]]))
    q.solve(True, d["g"], np.r_[d["lbA"], np.zeros(2 * d["nComp"])], np.r_[d["ubA"], np.full(2 * d["nComp"], np.inf)], d["x0"]); q.close()
    lc = lcqpow.LCQProblem(nV=2, nC=0, nComp=1); o = lcqpow.Options(); o.setPrintLevel(0); lc.setOptions(o)
    lc.loadLCQP(Q=2 * np.eye(2), g=np.array([-2., -2.]), L=np.array([[1., 0.]]), R=np.array([[0., 1.]]), order="C"); lc.runSolver(); del lc
    if k % 50 == 0 and k:
        print(k, "cycles: free memory change %.1f MiB" % ((torch.cuda.mem_get_info()[0] - free0) / 2**20), flush=True)
free1 = torch.cuda.mem_get_info()[0]
print("leak check:", "OK" if abs(free1 - free0) < 64 * 2**20 else "LEAK", "(%.1f MiB)" % ((free1 - free0) / 2**20))
