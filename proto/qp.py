import numpy as np, scipy.linalg as sla
INF = np.inf

class QPADMM:
    """min 1/2 x'Qx + g'x  s.t. lbA<=Ax<=ubA, lb<=x<=ub ; reduced-KKT ADMM + active-set polish.
    Dual convention returned: Qx+g - A'yA - yB = 0 (qpOASES sign)."""
    def __init__(s, Q, A, sigma=1e-6, rho=0.1, alpha=1.6, rho_eq_mult=1e3):
        s.Q, s.A = Q, A
        s.n = Q.shape[0]; s.m = A.shape[0]
        s.sigma, s.rho0, s.alpha, s.rho_eq_mult = sigma, rho, alpha, rho_eq_mult
        s.fact = None
        s.M = Q + sigma*np.eye(s.n)
        s.L1 = np.linalg.cholesky(s.M)
        s.At = sla.solve_triangular(s.L1, A.T, lower=True).T   # A L1^-T  (m x n)
        s.nfact = 0; s.x = None
    def setup(s, lbA, ubA, lb, ub):
        n, m = s.n, s.m
        s.l = np.concatenate([lbA, lb]); s.u = np.concatenate([ubA, ub])
        s.rhov = np.full(m+n, s.rho0)
        s.rhov[s.l == s.u] *= s.rho_eq_mult
        free = np.isinf(s.l) & np.isinf(s.u)
        s.rhov[free] = 0.0   # unconstrained rows drop out
        s.free = free
        rA, rB = s.rhov[:m], s.rhov[m:]
        K = s.Q + s.sigma*np.eye(n) + (s.A.T * rA) @ s.A + np.diag(rB)
        s.LK = np.linalg.cholesky(K); s.nfact += 1
    def Aop(s, x): return np.concatenate([s.A @ x, x])
    def ATop(s, v): return s.A.T @ v[:s.m] + v[s.m:]
    def solve(s, g, x0=None, y0=None, max_iter=4000, check_every=10, eps=1e-5, verbose=False):
        n, m = s.n, s.m
        l, u, rhov = s.l, s.u, s.rhov
        if (l > u).any(): return None, None, dict(status='infeasible', iters=0, polish=0)
        x = np.zeros(n) if x0 is None else x0.copy()
        y = np.zeros(n+m) if y0 is None else y0.copy()   # OSQP sign internally
        z = np.clip(s.Aop(x), l, u)
        info = dict(iters=0, polish=0, status='maxiter')
        for it in range(1, max_iter+1):
            rhs = s.sigma*x - g + s.ATop(rhov*z - y)
            xt = sla.cho_solve((s.LK, True), rhs)
            zt = s.Aop(xt)
            xn = s.alpha*xt + (1-s.alpha)*x
            zr = s.alpha*zt + (1-s.alpha)*z
            with np.errstate(invalid='ignore', divide='ignore'):
                w = np.where(rhov > 0, zr + y/np.where(rhov>0, rhov, 1), zr)
            zn = np.clip(w, l, u)
            yn = np.where(rhov > 0, y + rhov*(zr - zn), 0.0)
            x, z, y = xn, zn, yn
            if it % check_every == 0:
                Ax = s.Aop(x)
                rp = np.abs(Ax - z).max()
                rd = np.abs(s.Q @ x + g + s.ATop(y)).max()
                if verbose: print(it, rp, rd)
                if rp < eps and rd < eps:
                    ok, xp, yp = s.polish(g, x, y, z)
                    info['polish'] += 1
                    if ok:
                        info.update(iters=it, status='solved'); s.x = xp
                        return xp, -yp, info
        info['iters'] = max_iter
        return x, -y, info
    def polish(s, g, x, y, z, tol=1e-9):
        n, m = s.n, s.m
        l, u = s.l, s.u
        Ax = s.Aop(x)
        # active-set guess (OSQP rule): lower active if z-l < -y ; upper if u-z < y
        for trial in range(8):
            actL = (z - l < -y) & np.isfinite(l)
            actU = (u - z < y) & np.isfinite(u)
            eq = (l == u)
            actL |= eq; actU &= ~eq
            act = actL | actU
            b = np.where(actL, l, u)[act]
            idx = np.nonzero(act)[0]
            # build active matrix rows
            Aact = np.vstack([s.A[idx[idx < m]], np.eye(n)[idx[idx >= m]-m]]) if len(idx) else np.zeros((0, n))
            na = len(idx)
            delta = 1e-9
            K = np.block([[s.M, Aact.T], [Aact, -delta*np.eye(na)]])
            Kt = np.block([[s.Q, Aact.T], [Aact, np.zeros((na, na))]])
            rhs = np.concatenate([-g, b])
            lu = sla.lu_factor(K)
            sol = sla.lu_solve(lu, rhs)
            for _ in range(5):
                r = rhs - Kt @ sol
                sol = sol + sla.lu_solve(lu, r)
            xp = sol[:n]; ya = sol[n:]
            yp = np.zeros(n+m); yp[idx] = ya
            # verify KKT
            Axp = s.Aop(xp)
            scale = 1.0
            pviol = np.maximum(l - Axp, Axp - u)
            pviol[act] = np.abs(Axp[act] - b)
            dres = np.abs(s.Q @ xp + g + s.ATop(yp)).max()
            # dual sign: lower-active => y<=0 ; upper-active => y>=0 (OSQP sign)
            badsign = np.zeros(n+m, bool)
            badsign[actL & ~eq] = yp[actL & ~eq] > tol
            badsign[actU] = yp[actU] < -tol
            viol = pviol > tol
            if not viol.any() and not badsign.any() and dres < tol:
                return True, xp, yp
            # PDAS-style update: treat polished point as the new estimate
            x, y = xp, yp
            z = np.clip(Axp, l, u)
            # for constraints with bad sign: move z strictly inside so they're dropped
            y = np.where(badsign, 0.0, y)
            zin = np.where(badsign, np.clip(Axp, l, u), z)
            # violated constraints: add by giving them a multiplier of the right sign
            y = np.where(viol & (Axp < l), -1.0, y)
            y = np.where(viol & (Axp > u), 1.0, y)
            z = zin
        return False, x, y

def kkt_check(Q, g, A, lbA, ubA, lb, ub, x, yfull):
    n = Q.shape[0]
    yB, yA = yfull[:n], yfull[n:]
    stat = np.abs(Q@x + g - A.T@yA - yB).max()
    Ax = A@x
    pf = max(np.maximum(lbA-Ax, Ax-ubA).max(initial=0), np.maximum(lb-x, x-ub).max(initial=0), 0)
    # complementarity: y>0 => at lower; y<0 => at upper
    cs = 0.0
    for yy, v, lo, hi in ((yA, Ax, lbA, ubA), (yB, x, lb, ub)):
        with np.errstate(invalid='ignore'):
            cs = max(cs, np.abs(np.where(yy > 0, yy*np.where(np.isfinite(lo), v-lo, np.inf), 0)).max(initial=0))
            cs = max(cs, np.abs(np.where(yy < 0, yy*np.where(np.isfinite(hi), hi-v, np.inf), 0)).max(initial=0))
    return stat, pf, cs
