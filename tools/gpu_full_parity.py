"""Whole-workload parity: every instance of the node-sized synthetic job (8192 LCQPs, BASELINE configs[3]) solved by the
batched HIP path on one GPU and by the CPU oracle on all host cores; prints the largest primal / dual difference and how
many instances took a different number of iterates.   usage: python tools/gpu_full_parity.py [instances] [chunk]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la  # noqa: E402
import oracle_py as O  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n, nC, nComp = 256, 512, 64
threads = len(os.sched_getaffinity(0))
O.build(); O.lib()
bt = la.BatchLCQP(chunk, n, nC, nComp, opt=la.default_options(perturbStep=0, printLevel=0))
oopt = O.default_options(perturbStep=0, printLevel=0)
dx = dy = 0.0
n_iter_diff = n_ret_diff = n_ok = 0
hist = {}
sum_g = sum_c = 0
worst = None
t0 = time.time()
for first in range(0, total, chunk):
    bt.generate_synthetic(first)
    bt.run()
    x, y, st = bt.solution()
    ok, xo, yo, so = O.synth_batch_solve(first, chunk, n, nC, nComp, opt=oopt, threads=threads)
    for b in range(chunk):
        if st[b]["returnValue"] != so[b]["returnValue"]:
            n_ret_diff += 1
            continue
        n_ok += st[b]["returnValue"] == 0
        ex, ey = float(np.abs(x[b] - xo[b]).max()), float(np.abs(y[b] - yo[b]).max())
        if ex > dx:
            dx, worst = ex, first + b
        dy = max(dy, ey)
        n_iter_diff += (st[b]["iterTotal"], st[b]["iterOuter"]) != (so[b]["iterTotal"], so[b]["iterOuter"])
        dd = st[b]["iterTotal"] - so[b]["iterTotal"]
        hist[dd] = hist.get(dd, 0) + 1
        sum_g += st[b]["iterTotal"]; sum_c += so[b]["iterTotal"]
    print(f"instances {first}..{first + chunk - 1}: max|dx| {dx:.2e} max|dy| {dy:.2e} iterate-count differences {n_iter_diff} "
          f"return-code differences {n_ret_diff} ({time.time() - t0:.0f} s)", flush=True)
bt.close()
print(f"SUMMARY: {total} instances, {n_ok} solved on both sides, max|x_gpu - x_cpu| = {dx:.3e} (instance {worst}), "
      f"max|y_gpu - y_cpu| = {dy:.3e}, {n_iter_diff} with a different iterate count, {n_ret_diff} with a different return code")
print(f"iterTotal(gpu) - iterTotal(cpu) histogram: {dict(sorted(hist.items()))}; mean iterates gpu {sum_g / max(1, total):.2f} cpu {sum_c / max(1, total):.2f}")
