"""Where does k_factor spend its time alone and in a full machine?  A -DLCQP_FACTOR_PROFILE build (python tools/build_variants.py
fprof:-DLCQP_FACTOR_PROFILE) stamps the phases of wg_chol with the shader clock (thread 0 of every workgroup).
    python tools/micro/factor_phases.py build/ab/fprof.so 64 256 512 1024"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LCQPOW_HIP_LIBRARY"] = os.path.abspath(sys.argv[1])
from lcqpow_amd import capi as m
NAMES = ["tile load", "16x16 chains", "panel + update in LDS", "inversion of the tile", "write D", "panel tiles (MFMA)", "trailing tiles (MFMA)", "Q -> F1 copy", "whole kernel"]
for B in [int(v) for v in sys.argv[2:]] or [64, 256, 512, 1024]:
    bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
    bt.generate_synthetic(0)
    for rep in range(2):
        m._check(m.lib().lcqp_hip_batch_setup(bt.h), "setup"); bt.synchronize()
    prof = np.zeros((B, 16), dtype=np.uint64)
    m.lib().lcqp_hip_batch_read_profile.argtypes = [C.c_void_p, C.c_void_p]
    m._check(m.lib().lcqp_hip_batch_read_profile(bt.h, prof.ctypes.data_as(C.c_void_p)), "read_profile")
    p = prof.astype(np.float64).mean(axis=0)
    print(f"B = {B}: mean shader clocks per workgroup")
    for k, nme in enumerate(NAMES): print(f"   {nme:26s} {p[k]:10.0f}  {100 * p[k] / p[8]:5.1f} %")
    st = prof[:, 10].astype(np.int64); en = prof[:, 9].astype(np.int64)
    t0 = st.min()
    print(f"   workgroup starts after the first one, microseconds (100 MHz wall clock): median {np.median(st - t0) / 100:.1f}, 75 % {np.percentile(st - t0, 75) / 100:.1f}, "
          f"90 % {np.percentile(st - t0, 90) / 100:.1f}, max {(st - t0).max() / 100:.1f}; last end {(en.max() - t0) / 100:.1f}; mean life {np.mean(en - st) / 100:.1f}")
    bt.close()
