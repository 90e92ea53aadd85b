"""CPU side of the iterate-count comparison: the oracle on the first N synthetic instances against the iterate counts a GPU run
dumped (tools/gpu.py dump_iters).  usage: python tools/cmp_iters.py gpurun_out/r3/base_iters.npz [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O
G = np.load(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else len(G["it"])
O.build(); O.lib()
t0 = time.time()
ok, xo, yo, so = O.synth_batch_solve(0, N, 256, 512, 64, opt=O.default_options(perturbStep=0, printLevel=0), threads=len(os.sched_getaffinity(0)))
it_c = np.array([s["iterTotal"] for s in so]); it_g = G["it"][:N]
dd = it_g - it_c
print(f"{N} instances in {time.time() - t0:.1f} s; mean iterates gpu {it_g.mean():.3f} cpu {it_c.mean():.3f}; histogram gpu-cpu "
      f"{ {int(k): int((dd == k).sum()) for k in np.unique(dd)} }; max|dx| {np.abs(G['x'][:N] - xo).max():.2e} max|dy| {np.abs(G['y'][:N] - yo).max():.2e}; "
      f"cpu trials {np.mean([s['trials'] for s in so]):.2f} sweeps {np.mean([s['reserved'] for s in so]):.2f} (gpu {G['trials'][:N].mean():.2f} {G['sweeps'][:N].mean():.2f})")
