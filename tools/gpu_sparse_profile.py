"""Diagnostic: where does k_sparse_run spend its time?  Runs a -DLCQP_PROFILE build of the library (s_memtime stamps between phases,
per instance) on the sparse BASELINE workload and prints the share of each phase.  Shares only -- the stamped build is not the
measured build.   usage: python tools/gpu_sparse_profile.py --so=ab_tmp/libprof.so [B]
(build the library first, here or on the box:  python -c "import __graft_entry__ as g; g.build_hip(True, 'ab_tmp/libprof.so', ['-DLCQP_PROFILE'], 2)")"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd.capi as la
pre = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--so=")]
if pre:
    la._SO = os.path.abspath(pre[0])
import problems as P
args = [a for a in sys.argv[1:] if a.isdigit()]
B = int(args[0]) if args else 1024
n, nC, nK = 4096, 2048, 512
Qp, Ap = P.sparse_pattern(n, nC, nK)
base = [P.sparse_instance(i, n, nC, nK) for i in range(min(B, 64))]
inst = [base[i % len(base)] for i in range(B)]
sb = la.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=la.default_options(perturbStep=0, printLevel=0))
assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
               lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst])) == 0
sb.run(); sb.synchronize(); sb.run(); sb.synchronize()
x, y, st = sb.solution()
print("B", B, "lanes per instance", sb.lanes(), "timing (setup ms, solve ms)", sb.last_timing(), "solved", sum(s["returnValue"] == 0 for s in st),
      "-> %.0f LCQPs/s" % (B / (1e-3 * sum(sb.last_timing()))))
out = np.zeros(8)
la.lib().lcqp_hip_sparse_read_profile.argtypes = [C.c_void_p, C.c_void_p]
rc = la.lib().lcqp_hip_sparse_read_profile(sb.h, out.ctypes.data_as(C.c_void_p))
if rc != 0:
    print("library was not built with -DLCQP_PROFILE (rc %d)" % rc)
else:
    names = ["sparse products", "KKT assembly", "band factorisation", "forward sweeps", "backward sweeps", "vector operations", "LCQP level", "rhs of a correction (2 passes)"]
    print("mean ticks per instance %.3e (100 MHz clock: %.1f ms)" % (out.sum(), out.sum() / 1e5))
    for k in range(8):
        print("  %-20s %6.2f %%" % (names[k], 100 * out[k] / out.sum()))
for key in ("iterTotal", "trials", "factorizations", "corrections", "reserved"):
    v = np.array([s[key] for s in st], dtype=float)
    print("  %s mean %.1f min %d max %d" % (key, v.mean(), v.min(), v.max()))
