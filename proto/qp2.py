"""Prototype of the exact QP algorithm to be restated in C (oracle) and HIP.
min 1/2 x'Qx + g'x  s.t. l <= E x <= u,  E = [A_stacked ; I_fin] (box rows only for finite bounds).
Reduced-KKT ADMM rounds (factor once) + primal-dual active-set polish via Schur complement on M=Q+sp*I."""
import numpy as np, scipy.linalg as sla
INACT, LOWER, UPPER, EQ = 0, 1, 2, 3

def safe_chol(S, tau):
    """right-looking Cholesky; a pivot <= tau*S_kk(original) marks row k dependent: L_kk=1e150 (y_k -> 0)."""
    n = S.shape[0]; L = np.tril(S).copy(); d0 = np.diag(S).copy(); ndep = 0
    for k in range(n):
        p = L[k, k]
        if not (p > tau*d0[k]) or p <= 0:
            L[k, k] = 1e150; L[k+1:, k] = 0.0; ndep += 1
            continue
        L[k, k] = np.sqrt(p)
        L[k+1:, k] /= L[k, k]
        L[k+1:, k+1:] -= np.tril(np.outer(L[k+1:, k], L[k+1:, k]))
    return L, ndep

class QP2:
    def __init__(s, Q, A, sigma=1e-6, rho=0.1, alpha=1.6, rho_eq_mult=1e3, sp=1e-8, delta=1e-12,
                 admm_first=20, admm_hot=2, max_trials=8, max_rounds=200, nref=6, tol=1e-9):
        s.Q, s.A = Q, A; s.n = Q.shape[0]
        s.sigma, s.rho0, s.alpha, s.rho_eq_mult, s.sp, s.delta = sigma, rho, alpha, rho_eq_mult, sp, delta
        s.admm_first, s.admm_hot, s.max_trials, s.max_rounds, s.nref, s.tol = admm_first, admm_hot, max_trials, max_rounds, nref, tol
        s.tol_res = 1e-12; s.stat = dict(admm=0, trials=0, refine=0, rounds=0, na=[])
    def setup(s, lbA, ubA, lb, ub):
        n = s.n
        fin = np.isfinite(lb) | np.isfinite(ub)
        s.boxidx = np.nonzero(fin)[0]
        s.E = np.vstack([s.A, np.eye(n)[s.boxidx]])
        s.l = np.concatenate([lbA, lb[s.boxidx]]); s.u = np.concatenate([ubA, ub[s.boxidx]])
        s.mA = s.A.shape[0]; s.m = s.E.shape[0]
        s.rhov = np.full(s.m, s.rho0); s.rhov[s.l == s.u] *= s.rho_eq_mult
        s.rhov[np.isinf(s.l) & np.isinf(s.u)] = 0.0
        scale = max(np.abs(np.diag(s.Q)).max(), 1e-300)
        s.spv = s.sp*scale
        K = s.Q + s.sigma*np.eye(n) + (s.E.T*s.rhov) @ s.E
        s.LK = np.linalg.cholesky(K)
        s.L1 = np.linalg.cholesky(s.Q + s.spv*np.eye(n))
        s.Et = sla.solve_triangular(s.L1, s.E.T, lower=True).T
        s.first = True
    def solve(s, g, x0, y0ref):
        """y0ref: [box duals (n); row duals (mA)] in reference sign, or None. returns x, yref, info"""
        n, m, E, l, u, rhov = s.n, s.m, s.E, s.l, s.u, s.rhov
        if (l > u).any(): return None, None, dict(status='infeasible')
        x = x0.copy()
        y = np.zeros(m)
        if y0ref is not None:
            y[:s.mA] = -y0ref[n:]; y[s.mA:] = -y0ref[:n][s.boxidx]
        z = np.clip(E @ x, l, u)
        iters = s.admm_first if s.first else s.admm_hot
        s.first = False
        for rnd in range(s.max_rounds):
            s.stat['rounds'] += 1
            for it in range(iters):
                rhs = s.sigma*x - g + E.T @ (rhov*z - y)
                xt = sla.cho_solve((s.LK, True), rhs)
                zt = E @ xt
                xn = s.alpha*xt + (1-s.alpha)*x
                zr = s.alpha*zt + (1-s.alpha)*z
                rs = np.where(rhov > 0, rhov, 1.0)
                zn = np.clip(zr + y/rs, l, u)
                y = np.where(rhov > 0, y + rhov*(zr - zn), 0.0)
                x, z = xn, zn
                s.stat['admm'] += 1
            # active-set guess
            st = np.full(m, INACT)
            st[(z - l < -y) & np.isfinite(l)] = LOWER
            st[(u - z < y) & np.isfinite(u)] = UPPER
            st[l == u] = EQ
            ok, xp, yp = s.pdas(g, st, x, y)
            if ok:
                yref = np.zeros(n + s.mA)
                yref[n:] = -yp[:s.mA]; yref[s.boxidx] = -yp[s.mA:]
                return xp, yref, dict(status='solved', rounds=rnd+1)
            iters = min(max(2*iters, 10), 400)
        return None, None, dict(status='maxrounds')
    def pdas(s, g, st, xc, yc):
        """primal-dual active-set trials in correction (iterative-refinement) form.
        xc, yc: ADMM iterate (OSQP sign) used as starting point."""
        n, m, E, l, u = s.n, s.m, s.E, s.l, s.u
        gs = 1 + np.abs(g).max()
        tol = s.tol
        x = xc.copy(); yfull = np.where(st != INACT, yc, 0.0)
        prev_idx = None
        for trial in range(s.max_trials):
            s.stat['trials'] += 1
            idx = np.nonzero(st != INACT)[0]; na = len(idx)
            s.stat['na'].append(na)
            if na > min(2*n, m): return False, None, None
            b = np.where(st[idx] == UPPER, u[idx], l[idx])
            # residual evaluation: one Q pass + one fused E pass
            Ex = E @ x
            r1 = -g - s.Q @ x - E.T @ yfull
            r2 = b - Ex[idx]
            res_stat = np.abs(r1).max(); res_eq = np.abs(r2).max(initial=0)
            newst = st.copy()
            ftol = tol*(1 + np.abs(Ex))
            newst[(st == INACT) & (Ex < l - ftol)] = LOWER
            newst[(st == INACT) & (Ex > u + ftol)] = UPPER
            ytol = tol*gs
            newst[(st == LOWER) & (yfull > ytol)] = INACT
            newst[(st == UPPER) & (yfull < -ytol)] = INACT
            changed = (newst != st).any()
            if trial > 0 and not changed and res_stat <= s.tol_res*gs and res_eq <= s.tol_res*(1+np.abs(b).max(initial=0)):
                return True, x, yfull
            if changed and trial > 0:
                st = newst
                yfull = np.where(st != INACT, yfull, 0.0)
                idx = np.nonzero(st != INACT)[0]; na = len(idx)
                if na > min(2*n, m): return False, None, None
                b = np.where(st[idx] == UPPER, u[idx], l[idx])
                r1 = -g - s.Q @ x - E.T @ yfull     # (on device: recomputed cheaply from stored pieces)
                r2 = b - Ex[idx]
            Eta = s.Et[idx]
            same = prev_idx is not None and len(prev_idx) == na and (prev_idx == idx).all()
            if not same:
                S = Eta @ Eta.T
                LS, ndep = safe_chol(S, s.delta)
                s.stat['ndep'] = s.stat.get('ndep', 0) + ndep
                s.stat['fact'] = s.stat.get('fact', 0) + 1
            c = sla.solve_triangular(s.L1, r1, lower=True)
            dy = sla.cho_solve((LS, True), Eta @ c - r2) if na else np.zeros(0)
            dx = sla.solve_triangular(s.L1.T, c - Eta.T @ dy, lower=False)
            x = x + dx; yfull = yfull.copy(); yfull[idx] += dy
            prev_idx = idx
        return False, None, None
