"""One sparse problem alone (the band workload of bench.py, n = 4096) at every lane-group width: LCQP_SPARSE_LANES is read when the batch object is
created.  python tools/micro/sparse_single_lanes.py [B ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lcqpow_amd import capi as la
from lcqpow_amd import synth_sparse as S
la.request_hw_queues(8)
opt = la.default_options(perturbStep=0, printLevel=0)
ns, nCs, nKs = 4096, 2048, 512
Qpat, Apat, qo, eo = S.sparse_pattern_arrays(ns, nCs, nKs)


def make(Bs):
    sb = la.SparseBatchLCQP(Bs, ns, nCs, nKs, Qpat, Apat, device=0, opt=opt)
    inst = [S.sparse_values(i, ns, nCs, nKs, orders=(qo, eo)) for i in range(Bs)]
    rc = sb.load(0, Bs, np.stack([d["Qx"] for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["Ex"] for d in inst]),
                 lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]))
    assert rc == 0
    return sb


for B in [int(v) for v in sys.argv[1:]] or [1, 8, 64]:
    for lanes in (8, 16, 32, 64):
        if lanes == 8: os.environ.pop("LCQP_SPARSE_LANES", None)
        else: os.environ["LCQP_SPARSE_LANES"] = str(lanes)
        sb = make(B)
        sb.run(); sb.synchronize()
        ts = []
        for r in range(3):
            t0 = time.perf_counter(); sb.run(); sb.synchronize(); ts.append(time.perf_counter() - t0)
        x, _, st = sb.solution()
        print(f"B = {B:3d} lanes {sb.lanes():2d}: {1e3 * min(ts):8.1f} ms per step (best of 3), solved {sum(s['returnValue'] == 0 for s in st)}/{B}, "
              f"iterates {np.mean([s['iterTotal'] for s in st]):.1f}", flush=True)
        sb.close()
