"""Latency of one LCQP on the two single-problem paths: the reference's host loop over SubsolverHIP (one kernel launch per QP) and
a batch of one (whole homotopy in one launch).   usage: python tools/gpu_single_latency.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la, lcqpow_amd.lcqpow as lcqpow, oracle_py as O, problems as P

def host_loop(d):
    lc = lcqpow.LCQProblem(nV=d["nV"], nC=d["nC"], nComp=d["nComp"])
    o = lcqpow.Options(); o.setPrintLevel(0); o.setPerturbStep(False); lc.setOptions(o)
    lc.loadLCQP(Q=d["Q"], g=d["g"], L=d["L"], R=d["R"], A=d.get("A"), order="C", **{k: d[k] for k in ("lbL", "ubL", "lbR", "ubR", "lbA", "ubA", "lb", "ub", "x0", "y0") if k in d})
    t0 = time.perf_counter(); rc = lc.runSolver(); dt = time.perf_counter() - t0
    st = lcqpow.OutputStatistics(); lc.getOutputStatistics(st)
    return dt, int(rc), st.getIterTotal(), st.getSubproblemIter()

def batch_one(d):
    with_box = d.get("lb") is not None or d.get("ub") is not None
    bt = la.BatchLCQP(1, d["nV"], d["nC"], d["nComp"], with_box=with_box, opt=la.default_options(perturbStep=0, printLevel=0))
    bt.load(0, 1, d["Q"], d["g"], d["L"], d["R"], **{k: d.get(k) for k in P.KEYS})
    bt.run(); bt.synchronize()
    t0 = time.perf_counter(); bt.run(); bt.synchronize(); dt = time.perf_counter() - t0
    x, y, st = bt.solution(); bt.close()
    return dt, st[0]["returnValue"], st[0]["iterTotal"], st[0]["subproblemIter"]

O.build(); O.lib()
for name, d in (("circle N=100", P.circle(100)), ("example_data", P.example_data()), ("synthetic n=256", O.synth_generate(0, 256, 512, 64))):
    host_loop(d)     # warm up (library load, first launches)
    h = host_loop(d); b = batch_one(d)
    t0 = time.perf_counter(); ro = P.oracle_solve(O, d, O.default_options(perturbStep=0)); to = time.perf_counter() - t0
    print(f"{name:18s} host loop {1e3 * h[0]:8.1f} ms (ret {h[1]}, {h[2]} iterates, {h[3]} subproblem its)   batch of one {1e3 * b[0]:8.1f} ms ({b[2]} iterates)   CPU oracle {1e3 * to:8.1f} ms")
