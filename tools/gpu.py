"""One entry point for the GPU-side diagnostics (run on the GPU box from the repo root, e.g. through gpurun):

    python tools/gpu.py <command> [arguments]

Every command is a former one-off script tools/gpu_<command>.py, kept verbatim as a function; `python tools/gpu.py` lists them with
their first docstring line.  tools/gpu_ab.py (interleaved A/B timing of library variants, load_variant) and tools/gpu_fuzz.py (random
degenerate LCQPs, imported by the tests) stay modules of their own and are reachable here as `ab` and `fuzz`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

COMMANDS = {}


def command(name, doc):
    def deco(fn):
        COMMANDS[name] = (fn, doc)
        return fn
    return deco

@command("bias", "Where do the HIP batch loop and the CPU oracle part ways on the synthetic workload?")
def cmd_bias():
    """Where do the HIP batch loop and the CPU oracle part ways on the synthetic workload?

        python tools/gpu.py bias [--count N] [name=lib.so ...]

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


    main()


@command("bulk_profile", "bulk_profile")
def cmd_bulk_profile():
    import ctypes as C, os, sys, numpy as np
    sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tools")
    from gpu_ab import load_variant
    m = load_variant("v", sys.argv[1]); B = int(sys.argv[2])
    bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
    bt.generate_synthetic(0); bt.run(); bt.run()
    prof = np.zeros((B, 16), dtype=np.uint64)
    m.lib().lcqp_hip_batch_read_profile.argtypes = [C.c_void_p, C.c_void_p]
    m.lib().lcqp_hip_batch_read_profile(bt.h, prof.ctypes.data_as(C.c_void_p))
    p = prof[:, :5].astype(float).mean(axis=0)
    print("B", B, "timing", bt.last_timing(), "bulk stages mean cycles per LCQP: gather %.3e chol %.3e zero+diag %.3e inverse %.3e slots %.3e total %.3e; rebuilds %.2f" % (*p, p.sum(), prof[:, 11].mean()))


@command("checks", "Run every GPU building block against the CPU oracle / numpy and print the errors (no asserts).")
def cmd_checks():
    """Run every GPU building block against the CPU oracle / numpy and print the errors (no asserts).
    Used for bring-up on a gpurun box:  python tools/gpu.py checks [quick]"""
    import os, sys, time
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la
    import oracle_py as O

    def hdr(s): print("\n=== " + s, flush=True)

    def main():
        quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
        rng = np.random.default_rng(0)
        print("devices:", la.device_count())
        hdr("util symv / gemv / gemv_t / symm_product")
        for n, m in ((3, 2), (100, 37), (256, 640), (300, 50)):
            A = rng.standard_normal((2, n, n)); A = A + A.transpose(0, 2, 1)
            b = rng.standard_normal((2, n)); c = rng.standard_normal((2, n))
            d = la.util_symv(2.0, A, b, c)
            print(f"symv n={n}: err {np.abs(d - (2.0 * np.einsum('bij,bj->bi', A, b) + c)).max():.2e}")
            E = rng.standard_normal((2, m, n)); x = rng.standard_normal((2, n)); y = rng.standard_normal((2, m))
            print(f"gemv m={m} n={n}: err {np.abs(la.util_gemv(E, x) - np.einsum('bij,bj->bi', E, x)).max():.2e}")
            print(f"gemv_t m={m} n={n}: err {np.abs(la.util_gemv_t(E, y) - np.einsum('bij,bi->bj', E, y)).max():.2e}")
            L = rng.standard_normal((2, m, n)); R = rng.standard_normal((2, m, n))
            Cm = la.util_symm_product(L, R)
            ref = np.einsum('bki,bkj->bij', L, R); ref = ref + ref.transpose(0, 2, 1)
            print(f"symm_product m={m} n={n}: err {np.abs(Cm - ref).max():.2e}")
        hdr("cholesky + back-solve")
        for n in (5, 64, 100, 256, 300, 512):
            M = rng.standard_normal((3, n, n)); K = np.einsum('bij,bkj->bik', M, M) / n + np.eye(n)
            b = rng.standard_normal((3, n))
            x, ms = la.chol_solve(K, b, repeat=3)
            ref = np.linalg.solve(K, b[..., None])[..., 0]
            print(f"chol_solve n={n}: err {np.abs(x - ref).max():.2e}  ({ms:.3f} ms per back-solve launch)")
        hdr("QP subsolver vs oracle")
        for (n, m, seed) in ((2, 2, 1), (20, 30, 2), (64, 100, 3), (256, 640, 4)):
            r2 = np.random.default_rng(seed)
            M = r2.standard_normal((n, n)); Q = M.T @ M / n + np.eye(n)
            A = r2.standard_normal((m, n)) / np.sqrt(n); xs = r2.standard_normal(n)
            lbA = A @ xs - r2.uniform(0.1, 1, m); ubA = A @ xs + r2.uniform(0.1, 1, m)
            lbA[: m // 8] = ubA[: m // 8]          # some equalities
            ubA[m // 8: m // 4] = np.inf           # some one-sided
            g = 3 * r2.standard_normal(n)
            opt = la.default_options(); oopt = O.default_options()
            qo = O.QP(Q, A, oopt); ro = qo.solve(True, g, lbA, ubA, np.zeros(n)); xo, yo = qo.solution()
            qh = la.SubsolverHIP(n, m, Q, A, opt); rh = qh.solve(True, g, lbA, ubA, np.zeros(n)); xh, yh = qh.getSolution()
            print(f"QP n={n} m={m}: oracle {ro} {qo.counters()} | hip {rh} {qh.counters()} | dx {np.abs(xo - xh).max():.2e} dy {np.abs(yo - yh).max():.2e}")
            g2 = g + 0.3 * r2.standard_normal(n)
            ro = qo.solve(False, g2, lbA, ubA); xo, yo = qo.solution()
            rh = qh.solve(False, g2, lbA, ubA); xh, yh = qh.getSolution()
            print(f"   hot: oracle {ro} | hip {rh} | dx {np.abs(xo - xh).max():.2e} dy {np.abs(yo - yh).max():.2e}")
            if n == 20:
                lb = -0.3 * np.ones(n); ub = np.full(n, np.inf); ub[:5] = 0.2
                qo = O.QP(Q, A, oopt); ro = qo.solve(True, g, lbA, ubA, np.zeros(n), None, lb, ub); xo, yo = qo.solution()
                qh = la.SubsolverHIP(n, m, Q, A, opt); rh = qh.solve(True, g, lbA, ubA, np.zeros(n), None, lb, ub); xh, yh = qh.getSolution()
                print(f"   box: oracle {ro} | hip {rh} | dx {np.abs(xo - xh).max():.2e} dy {np.abs(yo - yh).max():.2e}")
        hdr("batched LCQP vs oracle (synthetic)")
        for (B, n, nC, nComp) in ((4, 64, 96, 16), (8, 256, 512, 64)) if not quick else ((2, 64, 96, 16),):
            opt = la.default_options(perturbStep=0); oopt = O.default_options(perturbStep=0)
            bt = la.BatchLCQP(B, n, nC, nComp, opt=opt)
            bt.generate_synthetic(0)
            t0 = time.time(); bt.run(); bt.synchronize(); dt = time.time() - t0
            x, y, st = bt.solution()
            print(f"batch B={B} n={n}: wall {dt*1e3:.1f} ms, timing {bt.last_timing()}, alg bytes {bt.algorithmic_bytes():.3e}")
            for b in range(min(B, 4)):
                d = bt.read_problem(b)
                ro = O.lcqp_solve(d['Q'], d['g'], d['L'], d['R'], A=d['A'], lbA=d['lbA'], ubA=d['ubA'], opt=oopt)
                so = ro['stats']; sh = st[b]
                print(f"  inst {b}: oracle ret {ro['ret']} it {so['iterTotal']}/{so['iterOuter']} rho {so['rhoOpt']} trials {so['trials']} fact {so['factorizations']} stat {so['status']}"
                      f" | hip ret {sh['returnValue']} it {sh['iterTotal']}/{sh['iterOuter']} rho {sh['rhoOpt']} trials {sh['trials']} fact {sh['factorizations']} stat {sh['status']} ef {sh['qpSolverExitFlag']}"
                      f" | dx {np.abs(ro['x'] - x[b]).max():.2e} dy {np.abs(ro['y'] - y[b]).max():.2e}")
            bt.close()
        hdr("batched LCQP: reference toy problems")
        Q = 2 * np.eye(2); g = np.array([-2., -2.]); L = np.array([[1., 0.]]); R = np.array([[0., 1.]])
        for ps in (0, 1):
            opt = la.default_options(perturbStep=ps); oopt = O.default_options(perturbStep=ps)
            bt = la.BatchLCQP(1, 2, 0, 1, opt=opt)
            rc = bt.load(0, 1, Q, g, L, R); bt.run(); x, y, st = bt.solution()
            ro = O.lcqp_solve(Q, g, L, R, opt=oopt)
            print(f"warm_up perturb={ps}: load rc {rc} hip x {x[0]} y {y[0]} st {st[0]['returnValue']} it {st[0]['iterTotal']} | oracle x {ro['x']} y {ro['y']} it {ro['stats']['iterTotal']}")
            bt.close()

    main()


@command("circle_sizes", "examples/OptimizeOnCircle.cpp at larger N (nV = 2 + 2N up to 1002): host loop over SubsolverHIP, batch of one, CPU oracle.")
def cmd_circle_sizes():
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


@command("determinism", "determinism")
def cmd_determinism():
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import lcqpow_amd as la
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    bt = la.BatchLCQP(B, 256, 512, 64, opt=la.default_options(perturbStep=0))
    bt.generate_synthetic(0)
    res = []
    for r in range(4):
        bt.run(); x, y, st = bt.solution(); res.append((x.copy(), y.copy(), st))
    for r in range(1, 4):
        dx = np.abs(res[r][0] - res[0][0]); dy = np.abs(res[r][1] - res[0][1])
        bad = np.nonzero(dx.max(axis=1) > 0)[0]
        sd = [b for b in range(B) if res[r][2][b] != res[0][2][b]]
        print(f"run {r} vs 0: max dx {dx.max():.3e} max dy {dy.max():.3e} instances differing {len(bad)} (first {bad[:8]}), stats differing {sd[:8]}")
        if len(bad):
            b = bad[0]; i = np.argmax(dx[b]); print("   e.g. inst", b, "coord", i, res[0][0][b, i], res[r][0][b, i], res[0][2][b], res[r][2][b])


@command("dual_diff", "dual_diff")
def cmd_dual_diff():
    import sys, numpy as np
    sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
    import lcqpow_amd as hip, oracle_py as O, problems as P
    for name in ("example_data", "circle"):
        d = getattr(P, name)()
        ro = P.oracle_solve(O, d, O.default_options(perturbStep=0))
        rh = P.hip_solve(hip, d, hip.default_options(perturbStep=0))
        dy = np.abs(ro["y"] - rh["y"])
        print(name, "max|dx|", np.abs(ro["x"]-rh["x"]).max(), "max|dy|", dy.max(), "rows with |dy|>1e-6:", np.nonzero(dy > 1e-6)[0][:40], "n", d["nV"], d["nC"], d["nComp"])


@command("dump_iters", "Dump iterTotal / iterOuter / rhoOpt / x of the first N synthetic instances solved by a library variant to gpurun_out/r3/<tag>_iters.npz.")
def cmd_dump_iters():
    """Dump iterTotal / iterOuter / rhoOpt / x of the first N synthetic instances solved by a library variant to gpurun_out/r3/<tag>_iters.npz.
    usage: python tools/gpu.py dump_iters lib.so tag [N]"""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_ab import load_variant
    m = load_variant("v", sys.argv[1])
    tag = sys.argv[2]
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    bt = m.BatchLCQP(N, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
    bt.generate_synthetic(0)
    bt.run()
    x, y, st = bt.solution()
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3", tag + "_iters.npz"), it=np.array([s["iterTotal"] for s in st]), outer=np.array([s["iterOuter"] for s in st]),
             rho=np.array([s["rhoOpt"] for s in st]), ret=np.array([s["returnValue"] for s in st]), trials=np.array([s["trials"] for s in st]),
             sweeps=np.array([s["reserved"] for s in st]), x=x.astype(np.float64), y=y)
    print(tag, "mean iterates", np.mean([s["iterTotal"] for s in st]), "timing", bt.last_timing())


@command("dump_prof", "Dump the per-instance phase cycle counters of a -DLCQP_PROFILE library and the instance statistics to gpurun_out/r3/<tag>_prof.npz.")
def cmd_dump_prof():
    """Dump the per-instance phase cycle counters of a -DLCQP_PROFILE library and the instance statistics to gpurun_out/r3/<tag>_prof.npz.
    usage: python tools/gpu.py dump_prof lib.so tag [B]"""
    import ctypes as C, os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_ab import load_variant
    m = load_variant("v", sys.argv[1]); tag = sys.argv[2]; B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
    bt.generate_synthetic(0); bt.run(); bt.run()
    x, y, st = bt.solution()
    prof = np.zeros((B, 16), dtype=np.uint64)
    m.lib().lcqp_hip_batch_read_profile.argtypes = [C.c_void_p, C.c_void_p]
    m.lib().lcqp_hip_batch_read_profile(bt.h, prof.ctypes.data_as(C.c_void_p))
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
    keys = ("iterTotal", "trials", "reserved", "factorizations", "corrections", "admmIter")
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3", tag + "_prof.npz"), prof=prof, **{k: np.array([s[k] for s in st]) for k in keys}, timing=np.array(bt.last_timing()))
    print("dumped", tag, bt.last_timing())


@command("dump_stamps", "Diagnostic (-DLCQP_PROFILE_STAMPS build): per-iterate clock stamps of every instance -> gpurun_out/r3/<tag>_stamps.npz")
def cmd_dump_stamps():
    """Diagnostic (-DLCQP_PROFILE_STAMPS build): per-iterate clock stamps of every instance -> gpurun_out/r3/<tag>_stamps.npz"""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_ab import load_variant
    m = load_variant("v", sys.argv[1]); tag = sys.argv[2]; B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0, storeSteps=1))
    bt.generate_synthetic(0); bt.run(); bt.run()
    x, y, st = bt.solution()
    stamps = np.zeros((B, 80))
    for b in range(B):
        s, _ = bt.trace(b, 80)
        stamps[b, :len(s)] = s[:, 7]
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3", tag + "_stamps.npz"), stamps=stamps, it=np.array([s["iterTotal"] for s in st]), timing=np.array(bt.last_timing()))
    print("dumped", tag, bt.last_timing())


@command("dump_traces", "Dump the device traces (scalars and xk per iterate) of some synthetic instances to gpurun_out/r3/traces.npz.")
def cmd_dump_traces():
    """Dump the device traces (scalars and xk per iterate) of some synthetic instances to gpurun_out/r3/traces.npz.
    usage: python tools/gpu.py dump_traces lib.so id id ..."""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_ab import load_variant
    m = load_variant("v", sys.argv[1])
    ids = [int(a) for a in sys.argv[2:]]
    B = max(ids) + 1
    bt = m.BatchLCQP(B, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0, storeSteps=1))
    bt.generate_synthetic(0)
    bt.run()
    x, y, st = bt.solution()
    out = {}
    for b in ids:
        s, xs = bt.trace(b, 128)
        out[f"s{b}"] = s; out[f"x{b}"] = xs; out[f"y{b}"] = y[b]
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "r3", "traces.npz"), **out)
    print("dumped", ids)


@command("full_parity", "Whole-workload parity: every instance of the node-sized synthetic job (8192 LCQPs, BASELINE configs[3]) solved by the")
def cmd_full_parity():
    """Whole-workload parity: every instance of the node-sized synthetic job (8192 LCQPs, BASELINE configs[3]) solved by the
    batched HIP path on one GPU and by the CPU oracle on all host cores; prints the largest primal / dual difference and how
    many instances took a different number of iterates.   usage: python tools/gpu.py full_parity [instances] [chunk]"""
    import os
    import sys
    import time

    import numpy as np

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la  # noqa: E402
    import oracle_py as O  # noqa: E402

    total = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n, nC, nComp = 256, 512, 64
    threads = len(os.sched_getaffinity(0))
    O.build(); O.lib()
    bt = la.BatchLCQP(chunk, n, nC, nComp, opt=la.default_options(perturbStep=0, printLevel=0))
    oopt = O.default_options(perturbStep=0, printLevel=0)
    dx = dy = 0.0
    n_iter_diff = n_ret_diff = n_ok = 0
    hist = {}
    sum_g = sum_c = 0
    worst = None
    t0 = time.time()
    for first in range(0, total, chunk):
        bt.generate_synthetic(first)
        bt.run()
        x, y, st = bt.solution()
        ok, xo, yo, so = O.synth_batch_solve(first, chunk, n, nC, nComp, opt=oopt, threads=threads)
        for b in range(chunk):
            if st[b]["returnValue"] != so[b]["returnValue"]:
                n_ret_diff += 1
                continue
            n_ok += st[b]["returnValue"] == 0
            ex, ey = float(np.abs(x[b] - xo[b]).max()), float(np.abs(y[b] - yo[b]).max())
            if ex > dx:
                dx, worst = ex, first + b
            dy = max(dy, ey)
            n_iter_diff += (st[b]["iterTotal"], st[b]["iterOuter"]) != (so[b]["iterTotal"], so[b]["iterOuter"])
            dd = st[b]["iterTotal"] - so[b]["iterTotal"]
            hist[dd] = hist.get(dd, 0) + 1
            sum_g += st[b]["iterTotal"]; sum_c += so[b]["iterTotal"]
        print(f"instances {first}..{first + chunk - 1}: max|dx| {dx:.2e} max|dy| {dy:.2e} iterate-count differences {n_iter_diff} "
              f"return-code differences {n_ret_diff} ({time.time() - t0:.0f} s)", flush=True)
    bt.close()
    print(f"SUMMARY: {total} instances, {n_ok} solved on both sides, max|x_gpu - x_cpu| = {dx:.3e} (instance {worst}), "
          f"max|y_gpu - y_cpu| = {dy:.3e}, {n_iter_diff} with a different iterate count, {n_ret_diff} with a different return code")
    print(f"iterTotal(gpu) - iterTotal(cpu) histogram: {dict(sorted(hist.items()))}; mean iterates gpu {sum_g / max(1, total):.2f} cpu {sum_c / max(1, total):.2f}")


@command("leak_check", "Create / load / run / destroy many batch, QP and CSC objects and watch the free device memory (hipMemGetInfo via torch).")
def cmd_leak_check():
    """Create / load / run / destroy many batch, QP and CSC objects and watch the free device memory (hipMemGetInfo via torch)."""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import lcqpow_amd as la, lcqpow_amd.lcqpow as lcqpow, problems as P
    d = P.circle(20)
    free0 = None
    for k in range(151):
        if k == 1:
            free0 = torch.cuda.mem_get_info()[0]      # after the first cycle: the runtime's one-time scratch / code-object allocations are in
        bt = la.BatchLCQP(16, 64, 96, 16, opt=la.default_options(perturbStep=0)); bt.generate_synthetic(k); bt.run(); bt.solution(); bt.close()
        q = la.SubsolverHIP(d["nV"], d["nC"] + 2 * d["nComp"], d["Q"], np.vstack([d["A"], d["L"], d["R"]]))
        q.solve(True, d["g"], np.r_[d["lbA"], np.zeros(2 * d["nComp"])], np.r_[d["ubA"], np.full(2 * d["nComp"], np.inf)], d["x0"]); q.close()
        lc = lcqpow.LCQProblem(nV=2, nC=0, nComp=1); o = lcqpow.Options(); o.setPrintLevel(0); lc.setOptions(o)
        lc.loadLCQP(Q=2 * np.eye(2), g=np.array([-2., -2.]), L=np.array([[1., 0.]]), R=np.array([[0., 1.]]), order="C"); lc.runSolver(); del lc
        if k % 50 == 0 and k:
            print(k, "cycles: free memory change %.1f MiB" % ((torch.cuda.mem_get_info()[0] - free0) / 2**20), flush=True)
    free1 = torch.cuda.mem_get_info()[0]
    print("leak check:", "OK" if abs(free1 - free0) < 64 * 2**20 else "LEAK", "(%.1f MiB)" % ((free1 - free0) / 2**20))


@command("load_rate", "load_rate")
def cmd_load_rate():
    import os, sys, time
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la, oracle_py as O
    B, n, nC, nComp = 256, 256, 512, 64
    d = [O.synth_generate(i, n, nC, nComp) for i in range(8)]
    pack = lambda k: np.ascontiguousarray(np.stack([d[i % 8][k] for i in range(B)]))
    Q, g, L, R, A, lbA, ubA = (pack(k) for k in ("Q", "g", "L", "R", "A", "lbA", "ubA"))
    bt = la.BatchLCQP(B, n, nC, nComp, opt=la.default_options(perturbStep=0))
    bt.load(0, B, Q, g, L, R, A=A, lbA=lbA, ubA=ubA)
    t0 = time.perf_counter(); rc = bt.load(0, B, Q, g, L, R, A=A, lbA=lbA, ubA=ubA); dt = time.perf_counter() - t0
    byts = B * 8.0 * (n * n + (nC + 2 * nComp) * n)
    print(f"load rc {rc}: {B} instances in {dt*1e3:.1f} ms = {B/dt:.0f} instances/s, {byts/dt/1e9:.2f} GB/s of problem data")
    bt.run(); x, y, st = bt.solution(); print("solved", sum(s["returnValue"] == 0 for s in st))


@command("microbench", "microbench")
def cmd_microbench():
    import ctypes as C, os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import lcqpow_amd as la
    L = la.lib()
    L.lcqp_hip_bench_rows.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
    for (B, m, n) in ((1024, 640, 256), (4096, 640, 256), (1024, 140, 256)):
        for mode, name in ((1, "dots  A x "), (2, "axpy  A'y "), (3, "both      ")):
            ms = C.c_float(0)
            rc = L.lcqp_hip_bench_rows(B, m, n, mode, 20, C.byref(ms))
            by = B * m * n * 8.0
            print(f"B={B} m={m} n={n} {name}: rc {rc} {ms.value:.4f} ms  {by / ms.value / 1e6:.0f} GB/s")


@command("nch8_check", "nch8_check")
def cmd_nch8_check():
    import sys, numpy as np
    sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests"); sys.path.insert(0,"/root/repo/tools")
    from gpu_ab import load_variant
    import oracle_py as O
    m = load_variant("v", sys.argv[1])
    for (n, nC, nComp) in ((513, 0, 50), (600, 300, 100), (1024, 600, 256)):
        B = 2
        bt = m.BatchLCQP(B, n, nC, nComp, opt=m.default_options(perturbStep=0))
        bt.generate_synthetic(0); bt.run()
        x, y, st = bt.solution()
        for b in range(B):
            d = bt.read_problem(b)
            ro = O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], A=d["A"] if nC else None, lbA=d["lbA"] if nC else None, ubA=d["ubA"] if nC else None, opt=O.default_options(perturbStep=0), nV=n, nC=nC, nComp=nComp)
            print(sys.argv[1], (n, nC, nComp), b, "ret", st[b]["returnValue"], ro["ret"], "iters", st[b]["iterTotal"], ro["stats"]["iterTotal"], "trials", st[b]["trials"], ro["stats"]["trials"], "dx %.2e" % np.abs(ro["x"] - x[b]).max(), "timing", bt.last_timing())
        bt.close()


@command("phase_profile", "Diagnostic: where does k_lcqp_run spend its cycles?  Builds a -DLCQP_PROFILE copy of the library")
def cmd_phase_profile():
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
    pre = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--so=")]      # a prebuilt -DLCQP_PROFILE library (e.g. under build/ab/)
    if pre:
        so = os.path.abspath(pre[0])
    elif not os.path.exists(so) or "--rebuild" in sys.argv:
        import __graft_entry__ as ge      # the product's own build recipe (host unit + per-size kernel units), with the profile switch
        ge.build_hip(force=True, out=so, defines=["-DLCQP_PROFILE", *extra], only_nch=2)
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


@command("quick", "Quick check of a library variant on the synthetic workload against the oracle: solved counts, max|dx|, iterate histogram,")
def cmd_quick():
    """Quick check of a library variant on the synthetic workload against the oracle: solved counts, max|dx|, iterate histogram,
    work counters.  usage: python tools/gpu.py quick lib.so [N]"""
    import os, sys, time
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_ab import load_variant
    import oracle_py as O
    m = load_variant("v", sys.argv[1])
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    O.build(); O.lib()
    bt = m.BatchLCQP(N, 256, 512, 64, opt=m.default_options(perturbStep=0, printLevel=0))
    bt.generate_synthetic(0)
    bt.run(); bt.run()
    x, y, st = bt.solution()
    ok, xo, yo, so = O.synth_batch_solve(0, N, 256, 512, 64, opt=O.default_options(perturbStep=0, printLevel=0), threads=len(os.sched_getaffinity(0)))
    it_g = np.array([s["iterTotal"] for s in st]); it_c = np.array([s["iterTotal"] for s in so])
    dd = it_g - it_c
    mean = lambda k, S: float(np.mean([s[k] for s in S]))
    print("timing", bt.last_timing(), "solved gpu", sum(s["returnValue"] == 0 for s in st), "cpu", ok, "of", N)
    print("max|dx| %.2e max|dy| %.2e" % (np.abs(x - xo).max(), np.abs(y - yo).max()), "hist", {int(k): int((dd == k).sum()) for k in np.unique(dd)})
    for k in ("iterTotal", "trials", "reserved", "factorizations", "corrections", "admmIter"):
        print(f"  {k:16s} gpu {mean(k, st):8.2f} cpu {mean(k, so):8.2f}")
    ws = bt.work_sums() / N
    print("  work sums per LCQP:", ws, " alg MB per LCQP %.1f" % (bt.algorithmic_bytes() / N / 1e6))


@command("quick_parity", "Quick check of a library variant against the CPU oracle: python tools/gpu.py quick_parity path/to/lib.so [B] [n nC nComp]")
def cmd_quick_parity():
    """Quick check of a library variant against the CPU oracle: python tools/gpu.py quick_parity path/to/lib.so [B] [n nC nComp]"""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd.capi as capi
    capi._SO = os.path.abspath(sys.argv[1])
    import oracle_py as O
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    n, nC, nComp = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (256, 512, 64)
    bt = capi.BatchLCQP(B, n, nC, nComp, opt=capi.default_options(perturbStep=0, printLevel=0))
    bt.generate_synthetic(0)
    bt.run(); bt.synchronize()
    x, y, st = bt.solution()
    print("timing", bt.last_timing(), "solved", sum(s["returnValue"] == 0 for s in st), "/", B)
    ok, xo, yo, so = O.synth_batch_solve(0, B, n, nC, nComp, opt=O.default_options(perturbStep=0, printLevel=0), threads=8)
    print("oracle solved", ok, "max|dx|", np.abs(x - xo).max(), "max|dy|", np.abs(y - yo).max())
    for k in ("iterTotal", "trials", "reserved", "factorizations", "corrections", "admmIter"):
        print(k, np.mean([s[k] for s in st]), np.mean([s[k] for s in so]))
    bad = [b for b in range(B) if st[b]["returnValue"] != so[b]["returnValue"] or np.abs(x[b] - xo[b]).max() > 1e-8]
    print("instances differing:", bad[:10])


@command("shape_sweep", "Robustness sweep: synthetic LCQPs of many shapes, HIP batch vs CPU oracle (prints, no asserts).")
def cmd_shape_sweep():
    """Robustness sweep: synthetic LCQPs of many shapes, HIP batch vs CPU oracle (prints, no asserts)."""
    import os, sys, time
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la, oracle_py as O
    shapes = [(512, 256, 128), (512, 1024, 256), (384, 700, 100), (100, 0, 50), (33, 17, 16), (256, 1500, 64), (2, 0, 1), (129, 64, 1), (200, 300, 100), (64, 640, 8)]
    for (n, nC, nComp) in shapes:
        B = 4
        try:
            bt = la.BatchLCQP(B, n, nC, nComp, opt=la.default_options(perturbStep=0))
            bt.generate_synthetic(0)
            t0 = time.time(); bt.run(); x, y, st = bt.solution(); dt = time.time() - t0
            res = []
            for b in range(B):
                d = bt.read_problem(b)
                ro = O.lcqp_solve(d["Q"], d["g"], d["L"], d["R"], A=d["A"] if nC else None, lbA=d["lbA"] if nC else None, ubA=d["ubA"] if nC else None,
                                  opt=O.default_options(perturbStep=0), nV=n, nC=nC, nComp=nComp)
                res.append((st[b]["returnValue"], ro["ret"], st[b]["iterTotal"], ro["stats"]["iterTotal"], float(np.abs(ro["x"] - x[b]).max()) if ro["ret"] == 0 and st[b]["returnValue"] == 0 else None))
            bt.close()
            print((n, nC, nComp), "%.0f ms" % (dt * 1e3), res, flush=True)
        except Exception as e:
            print((n, nC, nComp), "EXC", e, flush=True)


@command("single_latency", "Latency of one LCQP on the two single-problem paths: the reference's host loop over SubsolverHIP (one kernel launch per QP) and")
def cmd_single_latency():
    """Latency of one LCQP on the two single-problem paths: the reference's host loop over SubsolverHIP (one kernel launch per QP) and
    a batch of one (whole homotopy in one launch).   usage: python tools/gpu.py single_latency"""
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


@command("sparse_check", "Sparse arm on the GPU against the sparse CPU oracle: python tools/gpu.py sparse_check [B] [n nC nComp]")
def cmd_sparse_check():
    """Sparse arm on the GPU against the sparse CPU oracle: python tools/gpu.py sparse_check [B] [n nC nComp]"""
    import os, sys, time
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la
    import oracle_py as O, problems as P
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n, nC, nK = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 2048, 512)
    Qp, Ap = P.sparse_pattern(n, nC, nK)
    inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
    sb = la.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=la.default_options(perturbStep=0, printLevel=0))
    print("half bandwidth", sb.bandwidth())
    rc = sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                 lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]))
    assert rc == 0, rc
    sb.run(); sb.synchronize()
    t0 = time.perf_counter(); sb.run(); sb.synchronize(); dt = time.perf_counter() - t0
    x, y, st = sb.solution()
    print("timing (setup ms, solve ms)", sb.last_timing(), "wall %.1f ms" % (1e3 * dt), "solved", sum(s["returnValue"] == 0 for s in st), "/", B,
          "alg GB %.3f" % (sb.algorithmic_bytes() / 1e9))
    opt = O.default_options(perturbStep=0, printLevel=0)
    perm = sb.ordering(); w = sb.bandwidth()
    nchk = min(B, 8)
    for b in range(nchk):
        d = inst[b]
        ro = O.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        print(b, "ret", st[b]["returnValue"], ro["ret"], "dx %.2e dy %.2e" % (np.abs(x[b] - ro["x"]).max(), np.abs(y[b] - ro["y"]).max()),
              {k: (st[b][k], ro["stats"][k]) for k in ("iterTotal", "trials", "factorizations", "corrections", "admmIter", "status")})


@command("sparse_profile", "Diagnostic: where does k_sparse_sched spend its time?  Runs a -DLCQP_PROFILE build of the library (s_memtime stamps between phases,")
def cmd_sparse_profile():
    """Diagnostic: where does k_sparse_sched spend its time?  Runs a -DLCQP_PROFILE build of the library (s_memtime stamps between phases,
    per instance) on the sparse BASELINE workload and prints the share of each phase.  Shares only -- the stamped build is not the
    measured build.   usage: python tools/gpu.py sparse_profile --so=build/ab/libprof.so [B]
    (build the library first, here or on the box:  python -c "import __graft_entry__ as g; g.build_hip(True, 'build/ab/libprof.so', ['-DLCQP_PROFILE'], 2)")"""
    import ctypes as C, os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd.capi as la
    pre = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--so=")]
    if pre:
        la._SO = os.path.abspath(pre[0])
    import problems as P
    args = [a for a in sys.argv[1:] if a.isdigit()]
    B = int(args[0]) if args else 1024
    n, nC, nK = 4096, 2048, 512
    Qp, Ap = P.sparse_pattern(n, nC, nK)
    base = [P.sparse_instance(i, n, nC, nK) for i in range(min(B, 64))]
    inst = [base[i % len(base)] for i in range(B)]
    sb = la.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=la.default_options(perturbStep=0, printLevel=0))
    assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                   lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst])) == 0
    sb.run(); sb.synchronize(); sb.run(); sb.synchronize()
    x, y, st = sb.solution()
    print("B", B, "lanes per instance", sb.lanes(), "timing (setup ms, solve ms)", sb.last_timing(), "solved", sum(s["returnValue"] == 0 for s in st),
          "-> %.0f LCQPs/s" % (B / (1e-3 * sum(sb.last_timing()))))
    out = np.zeros(8)
    la.lib().lcqp_hip_sparse_read_profile.argtypes = [C.c_void_p, C.c_void_p]
    rc = la.lib().lcqp_hip_sparse_read_profile(sb.h, out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        print("library was not built with -DLCQP_PROFILE (rc %d)" % rc)
    else:
        names = ["sparse products", "status test", "band factorisation", "forward sweeps", "backward sweeps", "vector operations", "LCQP level", "rhs of a correction (2 passes)"]
        print("mean ticks per instance %.3e (100 MHz clock: %.1f ms)" % (out.sum(), out.sum() / 1e5))
        for k in range(8):
            print("  %-20s %6.2f %%" % (names[k], 100 * out[k] / out.sum()))
    for key in ("iterTotal", "trials", "factorizations", "corrections", "reserved"):
        v = np.array([s[key] for s in st], dtype=float)
        print("  %s mean %.1f min %d max %d" % (key, v.mean(), v.min(), v.max()))


@command("spmv_bench", "SpMV rate of the device CSC products (lcqp_hip_csc_apply) at the sizes of BASELINE config 5 (n = 4096) and beyond.")
def cmd_spmv_bench():
    """SpMV rate of the device CSC products (lcqp_hip_csc_apply) at the sizes of BASELINE config 5 (n = 4096) and beyond.
    Bytes per product: 12 per non-zero (8 value + 4 index) + 4 per column pointer + 8 per input and output entry.
    usage: python tools/gpu.py spmv_bench"""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la


    def csc_from_dense_pattern(m, n, per_col, rng):
        p = [0]; i = []; x = []
        for c in range(n):
            rows = np.sort(rng.choice(m, size=min(per_col, m), replace=False))
            i.extend(rows.tolist()); x.extend(rng.standard_normal(rows.size).tolist()); p.append(len(i))
        return np.array(p, dtype=np.int32), np.array(i, dtype=np.int32), np.array(x)


    rng = np.random.default_rng(0)
    print("shape, nnz, per product: microseconds, GB/s (algorithmic bytes), fraction of 8 TB/s")
    for (m, n, per_col) in [(4096, 4096, 3), (4096, 4096, 20), (6142, 4096, 3), (65536, 65536, 20), (262144, 262144, 20), (1048576, 1048576, 16)]:
        p, i, x = csc_from_dense_pattern(m, n, per_col, rng)
        M = la.CSCMatrix(m, n, p, i, x)
        b = rng.standard_normal(n)
        d = M.apply(b, repeat=50); ms = M.last_ms
        ref = np.zeros(m); np.add.at(ref, i, x * np.repeat(b, np.diff(p)))
        assert np.abs(d - ref).max() < 1e-9 * (1 + np.abs(ref).max())
        bt = rng.standard_normal(m)
        dt = M.apply(bt, transposed=True, repeat=50); mst = M.last_ms
        nbytes = 12.0 * len(x) + 4.0 * (n + 1) + 8.0 * (m + n)
        for tag, t in (("A b ", ms), ("A'b ", mst)):
            print(f"{m}x{n} nnz {len(x):9d} {tag}: {1e3 * t:9.1f} us  {nbytes / (t * 1e-3) / 1e9:8.1f} GB/s  {nbytes / (t * 1e-3) / 8e12:6.3f}")
        M.close()


@command("trace_check", "trace_check")
def cmd_trace_check():
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la, oracle_py as O, problems as P
    for name in ("circle", "warm_up_binary", "synthetic", "example_data"):
        d = O.synth_generate(1, 64, 96, 16) if name == "synthetic" else getattr(P, name)()
        ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=400)
        rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
        so, sh = ro["trace_scalars"], rh["trace_scalars"]
        n = min(len(so), len(sh))
        dx = np.abs(ro["trace_x"][:n] - rh["trace_x"][:n]).max(axis=1)
        print(name, "iters", len(so), len(sh), "rho equal", np.array_equal(so[:n, 2], sh[:n, 2]), "max|dalpha| %.2e" % np.abs(so[:n, 3] - sh[:n, 3]).max(),
              "max|dphi| %.2e" % np.abs(so[:n, 1] - sh[:n, 1]).max(), "max|dstat| %.2e" % np.abs(so[:n, 0] - sh[:n, 0]).max(), "max dx per iterate %.2e" % dx.max(), "at", int(dx.argmax()))


@command("trace_diff", "Print the per-iterate trace (|stat|, phi, rho, alpha, obj, merit, |p|, qp iterations) of one named test problem from the HIP")
def cmd_trace_diff():
    """Print the per-iterate trace (|stat|, phi, rho, alpha, obj, merit, |p|, qp iterations) of one named test problem from the HIP
    batch loop beside the oracle's.  usage: python tools/gpu.py trace_diff warm_up_w_A"""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as hip
    import oracle_py as oracle
    import problems as P

    name = sys.argv[1] if len(sys.argv) > 1 else "warm_up_w_A"
    d = getattr(P, name)()
    r = P.hip_solve(hip, d, hip.default_options(perturbStep=0, storeSteps=1), trace=True)
    ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0, storeSteps=1), trace=256)
    th = (r["trace_scalars"], r["trace_x"]); to = (ro["trace_scalars"], ro["trace_x"])
    print("hip", r["ret"], r["stats"]); print("orc", ro["ret"], ro["stats"])
    sh, so = th[0], to[0]
    for k in range(max(len(sh), len(so))):
        a = sh[k] if k < len(sh) else None
        b = so[k] if k < len(so) else None
        print(k, "H", None if a is None else " ".join("%.6g" % v for v in a), "x", None if a is None else th[1][k][:4])
        print(k, "O", None if b is None else " ".join("%.6g" % v for v in b), "x", None if b is None else to[1][k][:4])


@command("trace_tail", "Where the HIP and oracle homotopies of one synthetic instance part: [stat, phi, rho, alpha] per stored iterate around the")
def cmd_trace_tail():
    """Where the HIP and oracle homotopies of one synthetic instance part: [stat, phi, rho, alpha] per stored iterate around the
    first iterate whose scalars differ (diagnostic for the one-cycle differences reported by `python tools/gpu.py full_parity`).
    usage: python tools/gpu.py trace_tail [instance ...]"""
    import os, sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lcqpow_amd as la, oracle_py as O, problems as P
    np.set_printoptions(linewidth=220, precision=6)
    for inst in [int(a) for a in sys.argv[1:]] or [2]:
        d = O.synth_generate(inst, 256, 512, 64)
        ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=400)
        rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
        so, sh = ro["trace_scalars"], rh["trace_scalars"]
        n = min(len(so), len(sh))
        rel = np.abs(so[:n] - sh[:n]) / (1e-300 + np.abs(so[:n]))
        bad = np.where((rel[:, 1] > 1e-6) | (so[:n, 2] != sh[:n, 2]) | (rel[:, 3] > 1e-6))[0]
        k = int(bad[0]) if len(bad) else n
        print("instance", inst, "iterates cpu", ro["stats"]["iterTotal"], "gpu", rh["stats"]["iterTotal"], "first differing stored iterate", k)
        lo, hi = max(0, k - 5), min(n, k + 4)
        print(" cpu:"); print(so[lo:hi])
        print(" gpu:"); print(sh[lo:hi])
        print(" max|dx| per iterate before the split:", np.abs(ro["trace_x"][:k] - rh["trace_x"][:k]).max(axis=1)[-6:] if k else None)



@command("ab", "interleaved A/B timing of library variants (tools/gpu_ab.py)")
def cmd_ab():
    import gpu_ab
    gpu_ab.main()


@command("fuzz", "random degenerate LCQPs, HIP vs oracle (tools/gpu_fuzz.py)")
def cmd_fuzz():
    import gpu_fuzz
    sys.exit(gpu_fuzz.main())


@command("fuzz_diverge", "first diverging iterate of HIP and oracle on chosen fuzz problems: python tools/gpu.py fuzz_diverge SEED ID [ID ...]")
def cmd_fuzz_diverge():
    """Root cause of the fuzz problems that end at another stationary point on the two sides (tools/gpu_fuzz.py): regenerates problem ID of
    the given seed, solves it with per-iterate traces on both sides and prints the iterates around the first one where the traces part:
    |xk_hip - xk_orc|, rho, alpha, stationarity, complementarity, QP iterations of either side, and how many candidate next points the QP
    of that iterate has (the QP is re-solved by the oracle's QP solver from both sides' xk: a PSD Hessian has a face of minimisers)."""
    import os
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lcqpow_amd as la, oracle_py as O, problems as P, gpu_fuzz
    seed = int(sys.argv[1]); ids = [int(a) for a in sys.argv[2:]]
    O.build(); O.lib(); O.lcqp_set_robust(1)
    rng = np.random.default_rng(seed)
    probs = {}
    for k in range(max(ids) + 1):
        d = gpu_fuzz.make(rng)
        if k in ids:
            probs[k] = d
    np.set_printoptions(linewidth=200, precision=3)
    for k in ids:
        d = probs[k]
        ro = P.oracle_solve(O, d, O.default_options(perturbStep=0), trace=1000)
        rh = P.hip_solve(la, d, la.default_options(perturbStep=0, storeSteps=1), trace=True)
        so, sh, xo, xh = ro["trace_scalars"], rh["trace_scalars"], ro["trace_x"], rh["trace_x"]
        Qe = np.linalg.eigvalsh(d["Q"])
        print(f"=== seed {seed} id {k}: n={d['nV']} nC={d['nC']} nComp={d['nComp']} keys={sorted(set(d) - {'Q','g','L','R','nV','nC','nComp'})} "
              f"eig(Q) min {Qe.min():.2e} max {Qe.max():.2e} rank-deficient dims {(Qe < 1e-9 * max(Qe.max(), 1e-300)).sum()}")
        print(f"    ret {ro['ret']}/{rh['ret']} iter {len(so)}/{len(sh)} final |dx| {np.abs(ro['x'] - rh['x']).max():.2e} status {ro['stats']['status']}/{rh['stats']['status']}")
        kk = min(len(so), len(sh))
        dxs = np.array([np.abs(xo[i] - xh[i]).max() for i in range(kk)])
        first = int(np.argmax(dxs > 1e-6 * (1 + np.abs(xo[:kk]).max()))) if (dxs > 1e-6 * (1 + np.abs(xo[:kk]).max())).any() else kk
        for i in range(max(0, first - 2), min(kk, first + 3)):
            print(f"    it {i}: |dx| {dxs[i]:.2e}  orc [stat {so[i,0]:.2e} phi {so[i,1]:.2e} rho {so[i,2]:g} alpha {so[i,3]:.6g} obj {so[i,4]:.10g} |p| {so[i,6]:.2e} qpit {so[i,7]:g}]"
                  f"  hip [stat {sh[i,0]:.2e} phi {sh[i,1]:.2e} rho {sh[i,2]:g} alpha {sh[i,3]:.6g} obj {sh[i,4]:.10g} |p| {sh[i,6]:.2e} qpit {sh[i,7]:g}]")
        if 0 < first < kk:
            # the QP that produced iterate `first`: min 1/2 x'Qx + (g_tilde + rho C xk)'x at xk = iterate first-1 -- objective values of the
            # two sides' next points in the penalised merit of THAT iterate: equal values = two minimisers of one QP (a face), not an error
            i = first - 1
            rho = so[i + 1, 2] if so[i + 1, 2] == sh[i + 1, 2] else so[i, 2]
            Cm = d["L"].T @ d["R"] + d["R"].T @ d["L"]
            lbL = d.get("lbL", np.zeros(d["nComp"])); lbR = d.get("lbR", np.zeros(d["nComp"]))
            gphi = -(d["R"].T @ lbL + d["L"].T @ lbR)
            for side, xs in (("orc", xo), ("hip", xh)):
                xk, xn = xs[i], xs[i + 1]
                gk = d["g"] + rho * gphi + rho * (Cm @ xk)
                f = lambda v: 0.5 * v @ d["Q"] @ v + gk @ v
                print(f"    {side}: QP at iterate {i} (rho {rho:g}): f(xk) {f(xk):.12g} f(x_next) {f(xn):.12g}; f(other side's x_next) {f((xh if side == 'orc' else xo)[i + 1]):.12g}; |xk_orc - xk_hip| {np.abs(xo[i] - xh[i]).max():.2e}")


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        print(__doc__)
        for k in sorted(COMMANDS):
            print(f"  {k:16s} {COMMANDS[k][1]}")
        sys.exit(0 if len(sys.argv) < 2 else 2)
    cmd = sys.argv.pop(1)
    sys.argv[0] = "gpu.py " + cmd
    COMMANDS[cmd][0]()
