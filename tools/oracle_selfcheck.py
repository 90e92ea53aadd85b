"""CPU proxy for the HIP-vs-oracle fuzz: the oracle against ITSELF when only the summation order of E x changes
(orc_qp_set_sum_order 0 / 1) on the problems of tools/gpu_fuzz.py.  Two implementations that do not share every rounding differ
where this one differs from itself; a rule that removes a difference here removes the coin, not its symptom.

usage: python tools/oracle_selfcheck.py COUNT SEED [SEED ...]      (no GPU needed)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_py as O  # noqa: E402
import problems as P  # noqa: E402


def make_problems(count, seed):
    import gpu_fuzz      # (the generator only; importing the package loads no GPU code)
    rng = np.random.default_rng(seed)
    return [gpu_fuzz.make(rng) for _ in range(count)]


def classify(ra, rb):
    if ra["ret"] != rb["ret"]:
        return "return codes differ"
    if ra["ret"] != 0:
        return "same"
    dx = float(np.abs(ra["x"] - rb["x"]).max())
    if dx > 1e-6 * (1 + float(np.abs(ra["x"]).max())):
        return "other stationary point"
    sa, sb = ra["stats"], rb["stats"]
    if (sa["iterTotal"], sa["iterOuter"], sa["status"]) != (sb["iterTotal"], sb["iterOuter"], sb["status"]):
        return "same solution, other iterate count"
    return "same"


def main():
    count = int(sys.argv[1])
    seeds = [int(a) for a in sys.argv[2:]]
    O.build(); O.lib(); O.lcqp_set_robust(1)
    tot = {}
    t0 = time.time()
    for seed in seeds:
        cats = {}
        for k, d in enumerate(make_problems(count, seed)):
            O.qp_set_sum_order(1)
            ra = P.oracle_solve(O, d, O.default_options(perturbStep=0))
            O.qp_set_sum_order(0)
            rb = P.oracle_solve(O, d, O.default_options(perturbStep=0))
            O.qp_set_sum_order(1)
            cat = classify(ra, rb)
            cats[cat] = cats.get(cat, 0) + 1
            if cat in ("return codes differ", "other stationary point"):
                print(f"[seed {seed} id {k}] n={d['nV']} nC={d['nC']} nComp={d['nComp']}: {cat}: device order {ra['ret']} (iter {ra['stats']['iterTotal']}) / plain order {rb['ret']} (iter {rb['stats']['iterTotal']})", flush=True)
        print(f"seed {seed}: {count} problems: {cats}", flush=True)
        for k, v in cats.items():
            tot[k] = tot.get(k, 0) + v
    print(f"TOTAL over {count * len(seeds)} problems ({time.time() - t0:.0f} s): {tot}")


if __name__ == "__main__":
    main()
