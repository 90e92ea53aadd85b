"""Soak of the sparse engine's scheduler: one large batch solved several times on the same handle -- every run has to give the same bits
(instances are regrouped differently by the persistent wavefronts each time), every instance has to be solved, and a sample has to match
the CPU oracle.   python tools/micro/sparse_soak.py [B] [runs]"""
import os, sys, time, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la
from lcqpow_amd import synth_sparse as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, nC, nK = 4096, 2048, 512
Qpat, Apat, qo, eo = S.sparse_pattern_arrays(n, nC, nK)
sb = la.SparseBatchLCQP(B, n, nC, nK, Qpat, Apat, opt=la.default_options(perturbStep=0, printLevel=0))
for c0 in range(0, B, 1024):
    inst = [S.sparse_values(i, n, nC, nK, orders=(qo, eo)) for i in range(c0, min(B, c0 + 1024))]
    sb.load(c0, len(inst), np.stack([d["Qx"] for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["Ex"] for d in inst]),
            lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]))
ref = None
for r in range(runs):
    t0 = time.perf_counter(); sb.run(); sb.synchronize(); dt = time.perf_counter() - t0
    x, y, st = sb.solution()
    h = hashlib.sha256(x.tobytes() + y.tobytes()).hexdigest()[:16]
    solved = sum(1 for s in st if s["returnValue"] == 0)
    it = np.array([s["iterTotal"] for s in st])
    print(f"run {r}: {B / dt:8.0f} LCQPs/s  solved {solved}/{B}  iterates mean {it.mean():.3f} min {it.min()} max {it.max()}  sha256(x, y) {h}", flush=True)
    if ref is None: ref = h
    assert h == ref and solved == B
import oracle_py as oracle
import problems as P
opt = oracle.default_options(perturbStep=0)
worst = 0.0
for b in (0, B // 3, B - 1):
    d = P.sparse_instance(b, n, nC, nK, span=6)
    ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
    worst = max(worst, float(np.abs(x[b] - ro["x"]).max()))
    assert ro["ret"] == 0 and np.abs(x[b] - ro["x"]).max() < 1e-9
print(f"identical bits in {runs} runs; oracle sample max |dx| = {worst:.2e}")
sb.close()
