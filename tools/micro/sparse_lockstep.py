"""How much of k_sparse_run is lock-step waiting?  The 8 instances of a wavefront run their QP solves together: a QP costs the wavefront
the largest trial count among its instances.  From the per-iterate trace (trials per QP) of the first 64 instances (8 wavefronts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la
import problems as P
n, nC, nK, B = 4096, 2048, 512, 64
Qp, Ap = P.sparse_pattern(n, nC, nK)
inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
sb = la.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=la.default_options(perturbStep=0, printLevel=0, storeSteps=1))
assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
               lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst])) == 0
sb.run(); sb.synchronize()
per = []
for b in range(B):
    sc, _ = sb.trace(b)
    per.append(np.asarray(sc)[:, 7])                       # subproblem iterations (trials) of the QP of each iterate
G = 8
tot_own, tot_wave = 0.0, 0.0
for w in range(B // G):
    grp = per[w * G:(w + 1) * G]
    K = max(len(g) for g in grp)
    M = np.zeros((G, K))
    for i, g in enumerate(grp): M[i, :len(g)] = g
    tot_own += M.sum() / G
    tot_wave += M.max(axis=0).sum()
print("trials an instance needs (mean over 64): %.1f; trials its wavefront executes: %.1f -> an instance is active %.0f %% of its wavefront's trials"
      % (tot_own / (B // G), tot_wave / (B // G), 100 * tot_own / tot_wave))
print("QPs per instance min/mean/max:", min(len(g) for g in per), np.mean([len(g) for g in per]), max(len(g) for g in per))
