"""Where the HIP and oracle homotopies of one synthetic instance part: [stat, phi, rho, alpha] per stored iterate around the
first iterate whose scalars differ (diagnostic for the one-cycle differences reported by tools/gpu_full_parity.py).
usage: python tools/gpu_trace_tail.py [instance ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, oracle_py as O, problems as P
np.set_printoptions(linewidth=220, precision=6)
for inst in [int(a) for a in sys.argv[1:]] or [2]:
    d = O.synth_generate(inst, 256, 512, 64)
    ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=400)
    rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
    so, sh = ro["trace_scalars"], rh["trace_scalars"]
    n = min(len(so), len(sh))
    rel = np.abs(so[:n] - sh[:n]) / (1e-300 + np.abs(so[:n]))
    bad = np.where((rel[:, 1] > 1e-6) | (so[:n, 2] != sh[:n, 2]) | (rel[:, 3] > 1e-6))[0]
    k = int(bad[0]) if len(bad) else n
    print("instance", inst, "iterates cpu", ro["stats"]["iterTotal"], "gpu", rh["stats"]["iterTotal"], "first differing stored iterate", k)
    lo, hi = max(0, k - 5), min(n, k + 4)
    print(" cpu:"); print(so[lo:hi])
    print(" gpu:"); print(sh[lo:hi])
    print(" max|dx| per iterate before the split:", np.abs(ro["trace_x"][:k] - rh["trace_x"][:k]).max(axis=1)[-6:] if k else None)
