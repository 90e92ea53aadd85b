"""Where do the HIP batch loop and the CPU oracle part ways on the synthetic workload?

    python tools/gpu_bias.py [--count N] [name=lib.so ...]

For every library variant (default: the product library): solve instances 0..N-1 with the device trace on, solve the same
instances with the oracle (trace on, all host cores), print the histogram of iterTotal(gpu) - iterTotal(cpu), and for the
first divergent instances the iterate at which the two traces part: which scalar (|stat|, phi, rho, alpha) and by how much.
"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from gpu_ab import load_variant  # noqa: E402

NAMES = ("stat", "phi", "rho", "alpha", "obj", "merit", "step", "qpit")


def oracle_traces(count, n, nC, nComp, threads):
    oopt = O.default_options(perturbStep=0, printLevel=0, storeSteps=1)
    res = [None] * count

    def work(lo, hi):
        for i in range(lo, hi):
            d = O.synth_generate(i, n, nC, nComp)
            res[i] = O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], A=d["A"], lbA=d["lbA"], ubA=d["ubA"], opt=oopt, trace=128)
    nth = min(threads, count)
    th = [threading.Thread(target=work, args=(k * count // nth, (k + 1) * count // nth)) for k in range(nth)]
    [t.start() for t in th]; [t.join() for t in th]
    return res


def main():
    count, variants = 1024, []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--count": count = int(args.pop(0))
        else: variants.append(a.split("=", 1))
    if not variants:
        variants = [["product", os.path.join(ROOT, "lcqpow_amd", "liblcqpow_hip.so")]]
    n, nC, nComp = 256, 512, 64
    O.build(); O.lib()
    threads = len(os.sched_getaffinity(0))
    ref = oracle_traces(count, n, nC, nComp, threads)
    it_c = np.array([r["stats"]["iterTotal"] for r in ref])
    print(f"oracle: {count} instances, mean iterates {it_c.mean():.3f}", flush=True)
    for nme, pth in variants:
        m = load_variant(nme, pth)
        bt = m.BatchLCQP(count, n, nC, nComp, opt=m.default_options(perturbStep=0, printLevel=0, storeSteps=1))
        bt.generate_synthetic(0)
        bt.run()
        x, y, st = bt.solution()
        it_g = np.array([s["iterTotal"] for s in st])
        dd = it_g - it_c
        hist = {int(k): int((dd == k).sum()) for k in np.unique(dd)}
        dx = max(float(np.abs(x[b] - ref[b]["x"]).max()) for b in range(count))
        print(f"== {nme}: mean iterates gpu {it_g.mean():.3f} cpu {it_c.mean():.3f}; histogram gpu-cpu {hist}; max|dx| {dx:.2e}", flush=True)
        first_kind = {}
        shown = 0
        for b in np.nonzero(dd)[0]:
            sg, xg = bt.trace(int(b), 128)
            so = ref[b]["trace_scalars"]
            k = 0
            kind = "length"
            while k < min(len(sg), len(so)):
                # rho differing = a penalty decision flipped at iterate k - 1 or k; alpha / stat / phi relative 1e-6 = the iterates left each other
                if sg[k][2] != so[k][2]: kind = "rho"; break
                if abs(sg[k][7] - so[k][7]) > 0: kind = "qpit"
                k += 1
            if kind == "rho" or k < min(len(sg), len(so)):
                pass
            first_kind[kind] = first_kind.get(kind, 0) + 1
            if shown < 12:
                shown += 1
                k0 = max(0, k - 3)
                print(f"-- instance {b}: iterates gpu {it_g[b]} cpu {it_c[b]}; first rho difference at iterate {k}")
                for j in range(k0, min(k + 2, min(len(sg), len(so)))):
                    print("   %3d G " % j + " ".join(f"{nm}={v:.9g}" for nm, v in zip(NAMES, sg[j])))
                    print("   %3d C " % j + " ".join(f"{nm}={v:.9g}" for nm, v in zip(NAMES, so[j])))
        print(f"   divergence kinds: {first_kind}", flush=True)
        bt.close()


if __name__ == "__main__":
    main()
