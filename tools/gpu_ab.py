"""A/B timing of library variants on one GPU, interleaved rounds in one process (cdna_hip_programming.md §5.4 rule 24).

    python tools/gpu_ab.py [--rounds R] [--batch B] name=path/to/lib.so ...

Each variant is a build of lcqpow_amd/csrc/lcqp_hip.hip (e.g. with -DLCQP_ONLY_NCH=2 -DLCQP_DEPTH=..), loaded through its own
ctypes handle.  Prints per variant: median / min of the setup and homotopy kernel times (HIP events on the launch stream),
solved count, work counters, and the largest difference of the solutions from the first variant.
"""
import ctypes as C, os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_variant(name, path):
    spec = importlib.util.spec_from_file_location("capi_" + name, os.path.join(ROOT, "lcqpow_amd", "capi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod._SO = os.path.abspath(path)
    mod.lib()
    return mod


def main():
    rounds, B, shape = 5, 1024, (256, 512, 64)
    variants = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--rounds": rounds = int(args.pop(0))
        elif a == "--batch": B = int(args.pop(0))
        elif a == "--shape": shape = tuple(int(v) for v in args.pop(0).split(","))
        else:
            nme, pth = a.split("=", 1)
            variants.append((nme, pth))
    mods = [(nme, load_variant(nme, pth)) for nme, pth in variants]
    bts = []
    for nme, m in mods:
        bt = m.BatchLCQP(B, *shape, opt=m.default_options(perturbStep=0, printLevel=0))
        bt.generate_synthetic(0)
        bt.run(); bt.synchronize()
        bts.append(bt)
    times = {nme: [] for nme, _ in mods}
    for r in range(rounds):
        for (nme, m), bt in zip(mods, bts):
            bt.run()
            times[nme].append(bt.last_timing())
    x0 = None
    for (nme, m), bt in zip(mods, bts):
        x, y, st = bt.solution()
        t = np.array(times[nme])
        ok = sum(s["returnValue"] == 0 for s in st)
        ab = bt.algorithmic_bytes()
        if x0 is None: x0 = x
        mean = lambda k: float(np.mean([s[k] for s in st]))
        print(f"{nme:12s} kernel ms median {np.median(t[:,1]):8.3f} min {t[:,1].min():8.3f} | setup ms median {np.median(t[:,0]):6.3f} | solved {ok}/{B} "
              f"| alg GB {ab/1e9:7.1f} -> {ab/np.median(t[:,1])/1e6/8000:5.3f} of 8 TB/s | iter {mean('iterTotal'):.2f} trials {mean('trials'):.1f} sweeps {mean('reserved'):.1f} "
              f"fact {mean('factorizations'):.1f} corr {mean('corrections'):.1f} | max|dx| vs first {np.abs(x - x0).max():.2e}", flush=True)
        bt.close()


if __name__ == "__main__":
    main()
