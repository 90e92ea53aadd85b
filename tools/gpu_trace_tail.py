"""Last iterates of synthetic instances whose iterate count differs between HIP and oracle: [stat, phi, rho, alpha] per
stored iterate (diagnostic for the one-penalty-cycle differences reported by tools/gpu_full_parity.py).
usage: python tools/gpu_trace_tail.py [how many instances to scan] [how many to print]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, oracle_py as O, problems as P
np.set_printoptions(linewidth=200, precision=3)
scan = int(sys.argv[1]) if len(sys.argv) > 1 else 64
show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bt = la.BatchLCQP(scan, 256, 512, 64, opt=la.default_options(perturbStep=0, printLevel=0))
bt.generate_synthetic(0); bt.run(); x, y, st = bt.solution(); bt.close()
ok, xo, yo, so = O.synth_batch_solve(0, scan, 256, 512, 64, opt=O.default_options(perturbStep=0, printLevel=0), threads=len(os.sched_getaffinity(0)))
diff = [b for b in range(scan) if st[b]["iterTotal"] != so[b]["iterTotal"]]
print("instances with a different iterate count:", diff)
for inst in diff[:show]:
    d = O.synth_generate(inst, 256, 512, 64)
    ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=400)
    rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
    so_, sh_ = ro["trace_scalars"], rh["trace_scalars"]
    print("instance", inst, "iterates cpu", ro["stats"]["iterTotal"], "gpu", rh["stats"]["iterTotal"], "outer", ro["stats"]["iterOuter"], rh["stats"]["iterOuter"])
    k0 = max(0, min(len(so_), len(sh_)) - 7)
    print(" cpu tail:"); print(so_[k0:])
    print(" gpu tail:"); print(sh_[k0:])
