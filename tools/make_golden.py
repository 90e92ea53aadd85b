"""Generates the committed fixtures under tests/golden/ (run in the build container only).

  example_data.npz   the reference's data files examples/example_data/*.txt (data, not source)
  oracle_golden.npz  outputs of the CPU oracle on the reference's test problems and on a few synthetic
                     instances: pins the oracle against regressions and lets the GPU box check the
                     HIP path against committed numbers as well as against a live oracle run.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

REF = "/root/reference/examples/example_data"
if os.path.isdir(REF):
    names = ["Q", "g", "L", "R", "A", "lbA", "ubA", "lbL", "ubL", "lbR", "ubR", "lb", "ub", "x0"]
    np.savez_compressed(os.path.join(GOLD, "example_data.npz"), **{k: np.loadtxt(os.path.join(REF, k + ".txt")) for k in names})
    print("wrote example_data.npz")

import oracle_py as O
import problems as P
out = {}
cases = dict(warm_up=P.warm_up(), warm_up_x0=P.warm_up_x0(), warm_up_w_A=P.warm_up_w_A(), warm_up_binary=P.warm_up_binary(),
             circle=P.circle(), example_data=P.example_data())
for name, d in cases.items():
    r = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=200)
    s = r["stats"]
    out[name + "_x"] = r["x"]; out[name + "_y"] = r["y"]
    out[name + "_stats"] = np.array([r["ret"], s["iterTotal"], s["iterOuter"], s["status"], s["rhoOpt"]], dtype=float)
    out[name + "_trace"] = r["trace_scalars"]
    print(name, r["ret"], s["iterTotal"], s["iterOuter"], s["status"], s["rhoOpt"])
for inst in range(4):
    for (n, nC, nComp) in ((64, 96, 16), (256, 512, 64)):
        d = O.synth_generate(inst, n, nC, nComp)
        r = O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], A=d["A"], lbA=d["lbA"], ubA=d["ubA"], opt=O.default_options(perturbStep=0))
        s = r["stats"]
        key = f"synth_{n}_{inst}"
        out[key + "_x"] = r["x"]; out[key + "_y"] = r["y"]
        out[key + "_stats"] = np.array([r["ret"], s["iterTotal"], s["iterOuter"], s["status"], s["rhoOpt"]], dtype=float)
        print(key, r["ret"], s["iterTotal"], s["iterOuter"], s["status"], s["rhoOpt"])
np.savez_compressed(os.path.join(GOLD, "oracle_golden.npz"), **out)
print("wrote oracle_golden.npz")
