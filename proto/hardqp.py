import numpy as np, scipy.linalg as sla, sys
def mk(n,m,seed):
    r2 = np.random.default_rng(seed)
    M = r2.standard_normal((n, n)); Q = M.T @ M / n + np.eye(n)
    A = r2.standard_normal((m, n)) / np.sqrt(n); xs = r2.standard_normal(n)
    lbA = A @ xs - r2.uniform(0.1, 1, m); ubA = A @ xs + r2.uniform(0.1, 1, m)
    lbA[: m // 8] = ubA[: m // 8]; ubA[m // 8: m // 4] = np.inf
    g = 3 * r2.standard_normal(n)
    return Q,A,lbA,ubA,g
Q,A,l,u,g = mk(256,640,4)
n,m = 256,640
def admm(rho0, adapt, iters=2000, sigma=1e-6, alpha=1.6, report=100):
    rhov = np.full(m, rho0); eq = l==u; rhov[eq]*=1e3
    def fact(): return sla.cho_factor(Q + sigma*np.eye(n) + (A.T*rhov)@A)
    F = fact(); x=np.zeros(n); z=np.clip(A@x,l,u); y=np.zeros(m); nf=1
    for it in range(1,iters+1):
        xt = sla.cho_solve(F, sigma*x - g + A.T@(rhov*z - y))
        zt = A@xt
        xn = alpha*xt+(1-alpha)*x; zr = alpha*zt+(1-alpha)*z
        zn = np.clip(zr + y/rhov, l, u); y = y + rhov*(zr-zn); x, z = xn, zn
        if it % 25 == 0:
            rp = np.abs(A@x - z).max(); rd = np.abs(Q@x + g + A.T@y).max()
            if it % report == 0:
                act = ((z-l < -y)&np.isfinite(l)) | ((u-z < y)&np.isfinite(u)) | eq
                print(f"  it {it} rp {rp:.2e} rd {rd:.2e} nact {act.sum()} rho {rhov[~eq][0]:.3g}")
            if rp < 1e-7 and rd < 1e-7: print("  converged at", it, "nfact", nf); return x,y,z
            if adapt:
                pn = rp/max(np.abs(A@x).max(), np.abs(z).max(), 1e-10); dn = rd/max(np.abs(Q@x).max(), np.abs(A.T@y).max(), np.abs(g).max(), 1e-10)
                f = np.sqrt(pn/max(dn,1e-30))
                if f > 5 or f < 0.2:
                    rhov *= np.clip(f, 1e-3, 1e3); rhov = np.clip(rhov, 1e-6, 1e6); F = fact(); nf+=1
    return x,y,z
for rho0, adapt in ((0.1, False), (0.1, True), (1.0, False), (3.0, False)):
    print("rho0", rho0, "adapt", adapt); admm(rho0, adapt, iters=1500, report=250)
