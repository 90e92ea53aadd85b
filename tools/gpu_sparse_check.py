"""Sparse arm on the GPU against the sparse CPU oracle: python tools/gpu_sparse_check.py [B] [n nC nComp]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la
import oracle_py as O, problems as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, nC, nK = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 2048, 512)
Qp, Ap = P.sparse_pattern(n, nC, nK)
inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
sb = la.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=la.default_options(perturbStep=0, printLevel=0))
print("half bandwidth", sb.bandwidth())
rc = sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
             lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]))
assert rc == 0, rc
sb.run(); sb.synchronize()
t0 = time.perf_counter(); sb.run(); sb.synchronize(); dt = time.perf_counter() - t0
x, y, st = sb.solution()
print("timing (setup ms, solve ms)", sb.last_timing(), "wall %.1f ms" % (1e3 * dt), "solved", sum(s["returnValue"] == 0 for s in st), "/", B,
      "alg GB %.3f" % (sb.algorithmic_bytes() / 1e9))
opt = O.default_options(perturbStep=0, printLevel=0)
perm = sb.ordering(); w = sb.bandwidth()
nchk = min(B, 8)
for b in range(nchk):
    d = inst[b]
    ro = O.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
    print(b, "ret", st[b]["returnValue"], ro["ret"], "dx %.2e dy %.2e" % (np.abs(x[b] - ro["x"]).max(), np.abs(y[b] - ro["y"]).max()),
          {k: (st[b][k], ro["stats"][k]) for k in ("iterTotal", "trials", "factorizations", "corrections", "admmIter", "status")})
