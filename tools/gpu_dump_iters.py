"""Dump iterTotal / iterOuter / rhoOpt / x of the first N synthetic instances solved by a library variant to gpurun_out/r3/<tag>_iters.npz.
usage: python tools/gpu_dump_iters.py lib.so tag [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_ab import load_variant
m = load_variant("v", sys.argv[1])
tag = sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
bt = m.BatchLCQP(N, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
bt.generate_synthetic(0)
bt.run()
x, y, st = bt.solution()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3", tag + "_iters.npz"), it=np.array([s["iterTotal"] for s in st]), outer=np.array([s["iterOuter"] for s in st]),
         rho=np.array([s["rhoOpt"] for s in st]), ret=np.array([s["returnValue"] for s in st]), trials=np.array([s["trials"] for s in st]),
         sweeps=np.array([s["reserved"] for s in st]), x=x.astype(np.float64), y=y)
print(tag, "mean iterates", np.mean([s["iterTotal"] for s in st]), "timing", bt.last_timing())
