import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import lcqpow_amd as la, oracle_py as O, problems as P
d = P.example_data()
ro = P.oracle_solve(O, d, O.default_options(perturbStep=0)); rh = P.hip_solve(la, d, la.default_options(perturbStep=0))
n, nC, nK = d["nV"], d["nC"], d["nComp"]
print("status", ro["stats"]["status"], rh["stats"]["status"], "iters", ro["stats"]["iterTotal"], rh["stats"]["iterTotal"], "rho", ro["stats"]["rhoOpt"], rh["stats"]["rhoOpt"])
for nm, r in (("orc", ro), ("hip", rh)):
    x, y, rho = r["x"], r["y"], r["stats"]["rhoOpt"]
    Lx, Rx = d["L"] @ x, d["R"] @ x
    yL = y[n + nC:n + nC + nK] + rho * Rx; yR = y[n + nC + nK:] + rho * Lx      # untransformed
    weak = np.nonzero((Lx <= 2.2e-13) & (Rx <= 2.2e-13))[0]
    print(nm, "weak pairs", weak.tolist())
    for i in weak:
        vL = np.nonzero(d["L"][i])[0]; vR = np.nonzero(d["R"][i])[0]
        print("   pair", i, "yL %.3e yR %.3e" % (yL[i], yR[i]), "L vars", vL.tolist(), d["L"][i, vL].tolist(), "R vars", vR.tolist(), d["R"][i, vR].tolist(),
              "box y on those", y[vL].tolist(), y[vR].tolist(), "lb", (d.get("lb")[vL].tolist() if d.get("lb") is not None else None), (d.get("lb")[vR].tolist() if d.get("lb") is not None else None))
