"""examples/OptimizeOnCircle.cpp at larger N (nV = 2 + 2N up to 1002): host loop over SubsolverHIP, batch of one, CPU oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, lcqpow_amd.lcqpow as lcqpow, oracle_py as O, problems as P
O.build(); O.lib()
for N in (255, 500):
    d = P.circle(N)
    lc = lcqpow.LCQProblem(nV=d["nV"], nC=d["nC"], nComp=d["nComp"])
    o = lcqpow.Options(); o.setPrintLevel(0); o.setPerturbStep(False); lc.setOptions(o)
    lc.loadLCQP(Q=d["Q"], g=d["g"], L=d["L"], R=d["R"], A=d["A"], lbA=d["lbA"], ubA=d["ubA"], x0=d["x0"], order="C")
    t0 = time.time(); rc = lc.runSolver(); th = time.time() - t0
    x = lc.getPrimalSolution(); st = lcqpow.OutputStatistics(); lc.getOutputStatistics(st)
    t0 = time.time(); rb = P.hip_solve(la, d, la.default_options(perturbStep=0)); tb = time.time() - t0
    t0 = time.time(); ro = P.oracle_solve(O, d, O.default_options(perturbStep=0)); to = time.time() - t0
    print(f"circle N={N} (nV={d['nV']}): host loop ret {int(rc)} {st.getIterTotal()} iterates {th:.2f} s x[:2]={x[:2]}; batch of one ret {rb['ret']} {rb['stats']['iterTotal']} iterates {tb:.2f} s (incl. create/load);"
          f" oracle ret {ro['ret']} {ro['stats']['iterTotal']} iterates {to:.2f} s; |x_host - x_oracle| {np.abs(x - ro['x']).max():.1e} |x_batch - x_oracle| {np.abs(rb['x'] - ro['x']).max():.1e}", flush=True)
