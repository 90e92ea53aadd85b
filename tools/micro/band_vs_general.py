"""One banded sparse problem (BASELINE config 5 shape) alone and in small batches: the band engine against the general LDL' (LCQP_SPARSE_GENERAL=1).
usage: python tools/micro/band_vs_general.py [B ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la
from lcqpow_amd import synth_sparse as S
n, nC, nK = 4096, 2048, 512
Qpat, Apat, qo, eo = S.sparse_pattern_arrays(n, nC, nK)
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    inst = [S.sparse_values(i, n, nC, nK, orders=(qo, eo)) for i in range(B)]
    for gen in (0, 1):
        if gen: os.environ["LCQP_SPARSE_GENERAL"] = "1"
        else: os.environ.pop("LCQP_SPARSE_GENERAL", None)
        sb = la.SparseBatchLCQP(B, n, nC, nK, Qpat, Apat, opt=la.default_options(perturbStep=0, printLevel=0))
        sb.load(0, B, np.stack([d["Qx"] for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["Ex"] for d in inst]), lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]))
        sb.run(); sb.synchronize()
        t = time.time(); sb.run(); sb.synchronize(); dt = time.time() - t
        x, y, st = sb.solution()
        print(f"B {B:4d} {'general LDL (fronts %d)' % sb.fronts() if gen else 'band engine (w %d, %d lanes)' % (sb.bandwidth(), sb.lanes())}: {1e3 * dt:8.1f} ms, solved {sum(s['returnValue'] == 0 for s in st)}, mean iterates {np.mean([s['iterTotal'] for s in st]):.1f}", flush=True)
        if gen == 0: x0 = x
        else: print(f"       max |x_general - x_band| {np.abs(x - x0).max():.2e}")
        sb.close()
