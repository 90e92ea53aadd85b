import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, oracle_py as O, problems as P
for name in ("circle", "warm_up_binary", "synthetic", "example_data"):
    d = O.synth_generate(1, 64, 96, 16) if name == "synthetic" else getattr(P, name)()
    ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=400)
    rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
    so, sh = ro["trace_scalars"], rh["trace_scalars"]
    n = min(len(so), len(sh))
    dx = np.abs(ro["trace_x"][:n] - rh["trace_x"][:n]).max(axis=1)
    print(name, "iters", len(so), len(sh), "rho equal", np.array_equal(so[:n, 2], sh[:n, 2]), "max|dalpha| %.2e" % np.abs(so[:n, 3] - sh[:n, 3]).max(),
          "max|dphi| %.2e" % np.abs(so[:n, 1] - sh[:n, 1]).max(), "max|dstat| %.2e" % np.abs(so[:n, 0] - sh[:n, 0]).max(), "max dx per iterate %.2e" % dx.max(), "at", int(dx.argmax()))
