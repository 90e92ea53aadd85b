"""One fuzz problem on both sides, with the statistics of either (run on the GPU box): python tools/fuzz_case.py [--trace] SEED ID [ID ...]
--trace: the oracle prints one line per round / trial of every QP (orc_qp_set_trace); a library built with -DLCQP_TRACE_QP
(LCQPOW_HIP_LIBRARY=build/ab/trace.so) prints the same lines from the device."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import lcqpow_amd as la, oracle_py as O, problems as P, gpu_fuzz, bench
args = sys.argv[1:]
trace_qp = "--trace" in args
args = [a for a in args if a != "--trace"]
seed = int(args[0]); ids = [int(a) for a in args[1:]]
print("kernel sources", bench.kernel_source_hash(), flush=True)
O.build(); O.lib(); O.lcqp_set_robust(1)
rng = np.random.default_rng(seed)
for k in range(max(ids) + 1):
    d = gpu_fuzz.make(rng)
    if k not in ids:
        continue
    ev = np.linalg.eigvalsh(d["Q"])
    print(f"=== seed {seed} id {k}: n={d['nV']} nC={d['nC']} nComp={d['nComp']} keys={sorted(set(d) - {'Q','g','L','R','nV','nC','nComp'})} null(Q) {(ev < 1e-9 * max(ev.max(), 1e-300)).sum()}", flush=True)
    O.qp_set_trace(trace_qp)
    sys.stderr.flush()
    ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=1200)
    O.qp_set_trace(0)
    sys.stderr.flush()
    rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
    for tag, r in (("orc", ro), ("hip", rh)):
        s = r["stats"]; ts = r["trace_scalars"]
        print(f"  {tag}: ret {r['ret']} iter {s['iterTotal']} outer {s['iterOuter']} rho {s['rhoOpt']:g} flag {s['qpSolverExitFlag']} qpiter {s['subproblemIter']} admm {s['admmIter']} trials {s['trials']}")
        for row in ts[-4:]:
            print(f"       stat {row[0]:.2e} phi {row[1]:.2e} rho {row[2]:g} alpha {row[3]:.4g} |p| {row[6]:.2e} qpit {row[7]:g}")
    so, sh, xo, xh = ro["trace_scalars"], rh["trace_scalars"], ro["trace_x"], rh["trace_x"]
    kk = min(len(so), len(sh))
    if kk == 0:
        print("  (one side recorded no iterate: the first QP failed there)", flush=True)
        continue
    dxs = np.array([np.abs(xo[i] - xh[i]).max() for i in range(kk)])
    big = dxs > 1e-6 * (1 + np.abs(xo[:kk]).max())
    first = int(np.argmax(big)) if big.any() else kk
    print(f"  first diverging iterate {first} of {len(so)}/{len(sh)}")
    for i in range(max(0, first - 2), min(kk, first + 2)):
        print(f"    it {i}: |dx| {dxs[i]:.2e} orc [stat {so[i,0]:.2e} phi {so[i,1]:.2e} rho {so[i,2]:g} a {so[i,3]:.4g} qpit {so[i,7]:g}] hip [stat {sh[i,0]:.2e} phi {sh[i,1]:.2e} rho {sh[i,2]:g} a {sh[i,3]:.4g} qpit {sh[i,7]:g}]", flush=True)
