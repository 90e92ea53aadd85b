"""Quick check of a library variant on the synthetic workload against the oracle: solved counts, max|dx|, iterate histogram,
work counters.  usage: python tools/gpu_quick.py lib.so [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_ab import load_variant
import oracle_py as O
m = load_variant("v", sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
O.build(); O.lib()
bt = m.BatchLCQP(N, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
bt.generate_synthetic(0)
bt.run(); bt.run()
x, y, st = bt.solution()
ok, xo, yo, so = O.synth_batch_solve(0, N, 256, 512, 64, opt=O.default_options(perturbStep=0, printLevel=0), threads=len(os.sched_getaffinity(0)))
it_g = np.array([s["iterTotal"] for s in st]); it_c = np.array([s["iterTotal"] for s in so])
dd = it_g - it_c
mean = lambda k, S: float(np.mean([s[k] for s in S]))
print("timing", bt.last_timing(), "solved gpu", sum(s["returnValue"] == 0 for s in st), "cpu", ok, "of", N)
print("max|dx| %.2e max|dy| %.2e" % (np.abs(x - xo).max(), np.abs(y - yo).max()), "hist", {int(k): int((dd == k).sum()) for k in np.unique(dd)})
for k in ("iterTotal", "trials", "reserved", "factorizations", "corrections", "admmIter"):
    print(f"  {k:16s} gpu {mean(k, st):8.2f} cpu {mean(k, so):8.2f}")
ws = bt.work_sums() / N
print("  work sums per LCQP:", ws, " alg MB per LCQP %.1f" % (bt.algorithmic_bytes() / N / 1e6))
