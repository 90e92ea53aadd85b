import sys, numpy as np
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import lcqpow_amd as hip, oracle_py as O, problems as P
for name in ("example_data", "circle"):
    d = getattr(P, name)()
    ro = P.oracle_solve(O, d, O.default_options(perturbStep=0))
    rh = P.hip_solve(hip, d, hip.default_options(perturbStep=0))
    dy = np.abs(ro["y"] - rh["y"])
    print(name, "max|dx|", np.abs(ro["x"]-rh["x"]).max(), "max|dy|", dy.max(), "rows with |dy|>1e-6:", np.nonzero(dy > 1e-6)[0][:40], "n", d["nV"], d["nC"], d["nComp"])
