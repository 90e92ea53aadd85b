"""Run every GPU building block against the CPU oracle / numpy and print the errors (no asserts).
Used for bring-up on a gpurun box:  python tools/gpu_checks.py [quick]"""
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

if __name__ == "__main__":
    main()
