"""Where do the wavefronts of k_sparse_sched spend their time?  Needs a -DLCQP_SCHED_PROFILE build of the library:
   python tools/build_variants.py schedprof:-DLCQP_SCHED_PROFILE ; python tools/micro/sparse_sched_profile.py build/ab/schedprof.so [B]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LCQPOW_HIP_LIBRARY"] = os.path.abspath(sys.argv[1])
import lcqpow_amd as la
from lcqpow_amd import synth_sparse as S
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
n, nC, nK = 4096, 2048, 512
Qpat, Apat, qo, eo = S.sparse_pattern_arrays(n, nC, nK)
sb = la.SparseBatchLCQP(B, n, nC, nK, Qpat, Apat, opt=la.default_options(perturbStep=0, printLevel=0))
for c0 in range(0, B, 1024):
    inst = [S.sparse_values(i, n, nC, nK, orders=(qo, eo)) for i in range(c0, min(B, c0 + 1024))]
    sb.load(c0, len(inst), np.stack([d["Qx"] for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["Ex"] for d in inst]),
            lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]))
L = la.lib()
L.lcqp_hip_sparse_sched_profile.argtypes = [C.c_void_p, C.c_void_p]
sb.run(); sb.synchronize()
p0 = np.zeros(21, dtype=np.uint64); L.lcqp_hip_sparse_sched_profile(sb.h, p0.ctypes.data_as(C.c_void_p))
t0 = time.perf_counter(); sb.run(); sb.synchronize(); dt = time.perf_counter() - t0
p1 = np.zeros(21, dtype=np.uint64); L.lcqp_hip_sparse_sched_profile(sb.h, p1.ctypes.data_as(C.c_void_p))
p = (p1 - p0).astype(float).reshape(7, 3)
print(f"B = {B}: {B / dt:.0f} LCQPs/s, {dt * 1e3:.0f} ms; lanes per instance {sb.lanes()}")
tot = p[:, 0].sum()
for k, nm in enumerate(("start", "round (ADMM preamble)", "trial head", "factorisation", "correction", "QP end + LCQP iterate", "polls without work")):
    ticks, steps, served = p[k]
    print(f"  {nm:24s} {100 * ticks / tot:5.1f} % of the wavefront time; {steps:10.0f} steps, {served / max(steps, 1):4.2f} instances per step, {ticks / max(steps, 1) / 100:8.1f} x 100 clocks per step")
sb.close()
