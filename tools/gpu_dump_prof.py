"""Dump the per-instance phase cycle counters of a -DLCQP_PROFILE library and the instance statistics to gpurun_out/r3/<tag>_prof.npz.
usage: python tools/gpu_dump_prof.py lib.so tag [B]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_ab import load_variant
m = load_variant("v", sys.argv[1]); tag = sys.argv[2]; B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
bt.generate_synthetic(0); bt.run(); bt.run()
x, y, st = bt.solution()
prof = np.zeros((B, 16), dtype=np.uint64)
m.lib().lcqp_hip_batch_read_profile.argtypes = [C.c_void_p, C.c_void_p]
m.lib().lcqp_hip_batch_read_profile(bt.h, prof.ctypes.data_as(C.c_void_p))
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
keys = ("iterTotal", "trials", "reserved", "factorizations", "corrections", "admmIter")
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3", tag + "_prof.npz"), prof=prof, **{k: np.array([s[k] for s in st]) for k in keys}, timing=np.array(bt.last_timing()))
print("dumped", tag, bt.last_timing())
