"""Dump the device traces (scalars and xk per iterate) of some synthetic instances to gpurun_out/r3/traces.npz.
usage: python tools/gpu_dump_traces.py lib.so id id ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_ab import load_variant
m = load_variant("v", sys.argv[1])
ids = [int(a) for a in sys.argv[2:]]
B = max(ids) + 1
bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0, storeSteps=1))
bt.generate_synthetic(0)
bt.run()
x, y, st = bt.solution()
out = {}
for b in ids:
    s, xs = bt.trace(b, 128)
    out[f"s{b}"] = s; out[f"x{b}"] = xs; out[f"y{b}"] = y[b]
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "r3", "traces.npz"), **out)
print("dumped", ids)
