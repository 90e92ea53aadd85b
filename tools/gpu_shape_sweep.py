"""Robustness sweep: synthetic LCQPs of many shapes, HIP batch vs CPU oracle (prints, no asserts)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, oracle_py as O
shapes = [(512, 256, 128), (512, 1024, 256), (384, 700, 100), (100, 0, 50), (33, 17, 16), (256, 1500, 64), (2, 0, 1), (129, 64, 1), (200, 300, 100), (64, 640, 8)]
for (n, nC, nComp) in shapes:
    B = 4
    try:
        bt = la.BatchLCQP(B, n, nC, nComp, opt=la.default_options(perturbStep=0))
        bt.generate_synthetic(0)
        t0 = time.time(); bt.run(); x, y, st = bt.solution(); dt = time.time() - t0
        res = []
        for b in range(B):
            d = bt.read_problem(b)
            ro = O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], A=d["A"] if nC else None, lbA=d["lbA"] if nC else None, ubA=d["ubA"] if nC else None,
                              opt=O.default_options(perturbStep=0), nV=n, nC=nC, nComp=nComp)
            res.append((st[b]["returnValue"], ro["ret"], st[b]["iterTotal"], ro["stats"]["iterTotal"], float(np.abs(ro["x"] - x[b]).max()) if ro["ret"] == 0 and st[b]["returnValue"] == 0 else None))
        bt.close()
        print((n, nC, nComp), "%.0f ms" % (dt * 1e3), res, flush=True)
    except Exception as e:
        print((n, nC, nComp), "EXC", e, flush=True)
