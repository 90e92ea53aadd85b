import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, oracle_py as O
B, n, nC, nComp = 256, 256, 512, 64
d = [O.synth_generate(i, n, nC, nComp) for i in range(8)]
pack = lambda k: np.ascontiguousarray(np.stack([d[i % 8][k] for i in range(B)]))
Q, g, L, R, A, lbA, ubA = (pack(k) for k in ("Q", "g", "L", "R", "A", "lbA", "ubA"))
bt = la.BatchLCQP(B, n, nC, nComp, opt=la.default_options(perturbStep=0))
bt.load(0, B, Q, g, L, R, A=A, lbA=lbA, ubA=ubA)
t0 = time.perf_counter(); rc = bt.load(0, B, Q, g, L, R, A=A, lbA=lbA, ubA=ubA); dt = time.perf_counter() - t0
byts = B * 8.0 * (n * n + (nC + 2 * nComp) * n)
print(f"load rc {rc}: {B} instances in {dt*1e3:.1f} ms = {B/dt:.0f} instances/s, {byts/dt/1e9:.2f} GB/s of problem data")
bt.run(); x, y, st = bt.solution(); print("solved", sum(s["returnValue"] == 0 for s in st))
