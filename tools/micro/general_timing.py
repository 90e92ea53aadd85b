"""Timing of the general sparse LDL' engine on 2-D grid problems: create (symbolic analysis), load, run.  usage: python tools/micro/general_timing.py g [g ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, problems as P
for g in [int(a) for a in sys.argv[1:]]:
    nK, nC = max(60, g * g // 14), max(40, g * g // 20)
    d = P.grid_lcqp(g, nK, nC)
    n = d["nV"]
    Qc, Ec = d["Q"].tocsc(), d["E"].tocsc(); Qc.sort_indices(); Ec.sort_indices()
    t0 = time.time()
    sb = la.SparseBatchLCQP(1, n, nC, nK, Qc, Ec, opt=la.default_options(perturbStep=0, printLevel=0))
    t1 = time.time()
    sb.load(0, 1, Qc.data[None, :], d["g"][None, :], Ec.data[None, :], lbA=d["lbA"][None, :], ubA=d["ubA"][None, :])
    t2 = time.time()
    sb.run(); sb.synchronize()
    t3 = time.time()
    x, y, st = sb.solution()
    s = st[0]
    print(f"g {g} n {n} m {nC + 2 * nK}: fronts {sb.fronts()} create {t1 - t0:.2f} s load {t2 - t1:.2f} s run {t3 - t2:.2f} s (device timing {sb.last_timing()}) ret {s['returnValue']} iter {s['iterTotal']} "
          f"factorizations {s['factorizations']} corrections {s['corrections']} trials {s['trials']}", flush=True)
    import ctypes as C
    out = np.zeros(8)
    la.lib().lcqp_hip_sparse_read_profile.argtypes = [C.c_void_p, C.c_void_p]
    if la.lib().lcqp_hip_sparse_read_profile(sb.h, out.ctypes.data_as(C.c_void_p)) == 0 and out.sum() > 0:      # a -DLCQP_PROFILE build (LCQPOW_HIP_LIBRARY=lcqpow_amd/liblcqpow_hip_prof.so)
        names = ["sparse products", "status test", "factorisation", "forward sweeps", "backward sweeps", "vector operations", "LCQP level", "rhs of a correction"]
        print("   profile (100 MHz ticks: %.0f ms): " % (out.sum() / 1e5) + ", ".join(f"{nm} {100 * v / out.sum():.1f} %" for nm, v in zip(names, out)), flush=True)
    sb.close()
