"""Diagnostic: where does k_lcqp_run spend its cycles?  Builds a -DLCQP_PROFILE copy of the library
(clock64 stamps between phases, thread 0 of every workgroup), runs the BASELINE batch once and prints the
share of each phase.  Shares only -- the stamped build is not the measured build."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "gpurun_out", "liblcqpow_hip_prof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
pre = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--so=")]      # a prebuilt -DLCQP_PROFILE library (e.g. under ab_tmp/)
if pre:
    so = os.path.abspath(pre[0])
elif not os.path.exists(so) or "--rebuild" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DLCQP_PROFILE", *extra,
                           "-Wno-pass-failed", "-o", so, os.path.join(ROOT, "lcqpow_amd", "csrc", "lcqp_hip.hip")])
import lcqpow_amd.capi as capi
capi._SO = so
la = capi
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
okw = {a.split("=")[0][2:]: int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--") and "=" in a and not a.startswith("--so=")}
print("options override:", okw, "defines:", extra)
bt = la.BatchLCQP(B, 256, 512, 64, opt=la.default_options(perturbStep=0, **okw))
bt.generate_synthetic(0)
bt.run(); bt.run()
x, y, st = bt.solution()
print("timing (setup ms, solve ms):", bt.last_timing(), "solved", sum(s["returnValue"] == 0 for s in st))
prof = np.zeros((B, 16), dtype=np.uint64)
la.lib().lcqp_hip_batch_read_profile.argtypes = [C.c_void_p, C.c_void_p]
la.lib().lcqp_hip_batch_read_profile(bt.h, prof.ctypes.data_as(C.c_void_p))
names = ["lcqp-level sweeps", "trial residual (Q+E sweep)", "factor: one-piece rebuild", "factor: appends", "corr: L1 trsv", "corr: rows of Et", "corr: pass over T", "admm", "misc/logic", "factor: rotations (deletes)", "factor: working-set bookkeeping"]
tot = prof[:, :11].sum(axis=1).astype(float)
print("mean cycles per instance: %.3e  (max %.3e, min %.3e)" % (tot.mean(), tot.max(), tot.min()))
qs = np.percentile(tot, [10, 50, 90, 99])
print("percentiles 10/50/90/99: %.3e %.3e %.3e %.3e;  mean/max = %.3f (share of the launch an average workgroup slot is busy)" % (*qs, tot.mean() / tot.max()))
it = np.array([s["iterTotal"] for s in st], dtype=float)
print("correlation of cycles with LCQP iterates: %.3f; iterates min/mean/max %d/%.1f/%d" % (np.corrcoef(tot, it)[0, 1], it.min(), it.mean(), it.max()))
for k, nme in enumerate(names):
    print(f"  {nme:28s} {100 * prof[:, k].astype(float).sum() / tot.sum():6.2f} %")
print("  per LCQP: one-piece rebuilds %.2f (mean rows %.0f), rotations %.1f, appends %.1f" % (prof[:, 11].mean(), prof[:, 12].sum() / max(1, prof[:, 11].sum()), prof[:, 13].mean(), prof[:, 14].mean()))
order = np.argsort(-tot)[:6]
for b in order:
    s = st[b]
    print(f"  slow instance {b}: cycles {tot[b]:.3e} iter {s['iterTotal']} trials {s['trials']} sweeps {s['reserved']} updates {s['factorizations']} corr {s['corrections']} admm {s['admmIter']} ret {s['returnValue']}"
          f" | shares: " + " ".join(f"{100 * prof[b, k] / tot[b]:.0f}" for k in range(9)))
