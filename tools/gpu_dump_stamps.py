"""Diagnostic (-DLCQP_PROFILE_STAMPS build): per-iterate clock stamps of every instance -> gpurun_out/r3/<tag>_stamps.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_ab import load_variant
m = load_variant("v", sys.argv[1]); tag = sys.argv[2]; B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0, storeSteps=1))
bt.generate_synthetic(0); bt.run(); bt.run()
x, y, st = bt.solution()
stamps = np.zeros((B, 80))
for b in range(B):
    s, _ = bt.trace(b, 80)
    stamps[b, :len(s)] = s[:, 7]
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3", tag + "_stamps.npz"), stamps=stamps, it=np.array([s["iterTotal"] for s in st]), timing=np.array(bt.last_timing()))
print("dumped", tag, bt.last_timing())
