import ctypes as C, os, sys, numpy as np
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tools")
from gpu_ab import load_variant
m = load_variant("v", sys.argv[1]); B = int(sys.argv[2])
bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
bt.generate_synthetic(0); bt.run(); bt.run()
prof = np.zeros((B, 16), dtype=np.uint64)
m.lib().lcqp_hip_batch_read_profile.argtypes = [C.c_void_p, C.c_void_p]
m.lib().lcqp_hip_batch_read_profile(bt.h, prof.ctypes.data_as(C.c_void_p))
p = prof[:, :5].astype(float).mean(axis=0)
print("B", B, "timing", bt.last_timing(), "bulk stages mean cycles per LCQP: gather %.3e chol %.3e zero+diag %.3e inverse %.3e slots %.3e total %.3e; rebuilds %.2f" % (*p, p.sum(), prof[:, 11].mean()))
