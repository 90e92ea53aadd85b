"""Print the per-iterate trace (|stat|, phi, rho, alpha, obj, merit, |p|, qp iterations) of one named test problem from the HIP
batch loop beside the oracle's.  usage: python tools/gpu_trace_diff.py warm_up_w_A"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as hip
import oracle_py as oracle
import problems as P

name = sys.argv[1] if len(sys.argv) > 1 else "warm_up_w_A"
d = getattr(P, name)()
r = P.hip_solve(hip, d, hip.default_options(perturbStep=0, storeSteps=1), trace=True)
ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0, storeSteps=1), trace=256)
th = (r["trace_scalars"], r["trace_x"]); to = (ro["trace_scalars"], ro["trace_x"])
print("hip", r["ret"], r["stats"]); print("orc", ro["ret"], ro["stats"])
sh, so = th[0], to[0]
for k in range(max(len(sh), len(so))):
    a = sh[k] if k < len(sh) else None
    b = so[k] if k < len(so) else None
    print(k, "H", None if a is None else " ".join("%.6g" % v for v in a), "x", None if a is None else th[1][k][:4])
    print(k, "O", None if b is None else " ".join("%.6g" % v for v in b), "x", None if b is None else to[1][k][:4])
