"""Quick check of a library variant against the CPU oracle: python tools/gpu_quick_parity.py path/to/lib.so [B] [n nC nComp]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd.capi as capi
capi._SO = os.path.abspath(sys.argv[1])
import oracle_py as O
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n, nC, nComp = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (256, 512, 64)
bt = capi.BatchLCQP(B, n, nC, nComp, opt=capi.default_options(perturbStep=0, printLevel=0))
bt.generate_synthetic(0)
bt.run(); bt.synchronize()
x, y, st = bt.solution()
print("timing", bt.last_timing(), "solved", sum(s["returnValue"] == 0 for s in st), "/", B)
ok, xo, yo, so = O.synth_batch_solve(0, B, n, nC, nComp, opt=O.default_options(perturbStep=0, printLevel=0), threads=8)
print("oracle solved", ok, "max|dx|", np.abs(x - xo).max(), "max|dy|", np.abs(y - yo).max())
for k in ("iterTotal", "trials", "reserved", "factorizations", "corrections", "admmIter"):
    print(k, np.mean([s[k] for s in st]), np.mean([s[k] for s in so]))
bad = [b for b in range(B) if st[b]["returnValue"] != so[b]["returnValue"] or np.abs(x[b] - xo[b]).max() > 1e-8]
print("instances differing:", bad[:10])
