import numpy as np
from collections import deque
EPS = 2.221e-16
class Opt:
    complTol = 1e3*EPS; statTol = 1e6*EPS; rho0 = 0.01; beta = 2.0
    zeroPenFirst = True; perturb = False; maxIter = 1000; maxRho = 1e8; nDyn = 3; eta = 0.9

def run_lcqp(d, qpsolver_factory, opt=Opt, x0=None, y0=None, lb=None, ub=None, log=None):
    Q, g, L, R, A = d['Q'], d['g'], d['L'], d['R'], d['A']
    n, nC, nComp = d['n'], d['nC'], d['nComp']
    lbL = d.get('lbL'); lbR = d.get('lbR'); ubL = d.get('ubL'); ubR = d.get('ubR')
    As = np.vstack([A.reshape(nC, n), L, R])
    lbA = np.concatenate([d['lbA'] if d.get('lbA') is not None else np.full(nC, -np.inf),
                          lbL if lbL is not None else np.zeros(nComp), lbR if lbR is not None else np.zeros(nComp)])
    ubA = np.concatenate([d['ubA'] if d.get('ubA') is not None else np.full(nC, np.inf),
                          ubL if ubL is not None else np.full(nComp, np.inf), ubR if ubR is not None else np.full(nComp, np.inf)])
    C = L.T@R + R.T@L
    lb = np.full(n, -np.inf) if lb is None else lb
    ub = np.full(n, np.inf) if ub is None else ub
    xk = np.zeros(n) if x0 is None else x0.copy()
    yk = None if y0 is None else y0.copy()
    qp = qpsolver_factory(Q, As); qp.setup(lbA, ubA, lb, ub)
    g_tilde = g.copy()
    phi_const = 0.0; g_phi = None
    if lbL is not None or lbR is not None:
        phi_const = lbL @ lbR
        g_phi = -(R.T@lbL + L.T@lbR)
    alphak = 1.0; rho = opt.rho0
    outer = inner = total = 0
    stats = dict(qpit=0, polish=0, qps=0)
    hist = deque()
    def getphi():
        return phi_const + (g_phi@xk if g_phi is not None else 0.0) + 0.5*xk@(C@xk)
    def solveqp(gk, first):
        nonlocal yk
        # warm start: x = xk ; y = yk (convert to OSQP internal: [A rows; box rows], sign flip)
        y0i = None
        if yk is not None:
            y0i = -np.concatenate([yk[n:], yk[:n]])
        x, y, info = qp.solve(gk, x0=xk, y0=y0i)
        stats['qpit'] += info['iters']; stats['polish'] += info['polish']; stats['qps'] += 1
        if info['status'] != 'solved':
            return None, info
        m = As.shape[0]
        yk = np.concatenate([y[m:], y[:m]])
        return x, info
    gk = g.copy() if opt.zeroPenFirst else rho*(C@xk) + g_tilde
    xnew, info = solveqp(gk, True)
    if xnew is None: return dict(ret='SUBPROBLEM_SOLVER_ERROR', info=info)
    pk = xnew - xk
    while True:
        xk = xk + alphak*pk
        Qk = Q + rho*C
        ykA = yk[n:]
        statk = Qk@xk + g_tilde - As.T@ykA - yk[:n]
        total += 1; inner += 1
        if log is not None: log.append((total, outer, np.abs(statk).max(), getphi(), rho, alphak, xk.copy()))
        # leyffer
        def leyffer():
            nd = opt.nDyn
            if nd <= 0: return False
            cur = getphi()
            if len(hist) < nd: hist.append(cur); return False
            if cur < opt.complTol:
                hist.popleft(); hist.append(cur); return False
            flag = True
            for i in range(nd):
                if cur < opt.eta*hist[i]: flag = False; break
            hist.popleft(); hist.append(cur)
            return flag
        def updpen():
            nonlocal rho, g_tilde
            hist.clear(); rho *= opt.beta
            if g_phi is not None: g_tilde = g + rho*g_phi
        if leyffer():
            updpen(); outer += 1; inner = 0
        gk = rho*(C@xk) + g_tilde
        if np.abs(statk).max() < opt.statTol:
            if getphi() < opt.complTol:
                # transform duals
                yk = yk.copy()
                yk[n+nC:n+nC+nComp] -= rho*(R@xk)
                yk[n+nC+nComp:] -= rho*(L@xk)
                return dict(ret='SUCCESS', x=xk, y=yk, rho=rho, total=total, outer=outer, **stats)
            else:
                updpen(); outer += 1; inner = 0
        if total > opt.maxIter: return dict(ret='MAX_ITER', total=total, **stats)
        if rho > opt.maxRho: return dict(ret='MAX_PEN', total=total, **stats)
        gk = rho*(C@xk) + g_tilde
        xnew, info = solveqp(gk, False)
        if xnew is None: return dict(ret='SUBPROBLEM_SOLVER_ERROR', info=info, total=total, **stats)
        pk = xnew - xk
        Qk = Q + rho*C
        qk = pk@(Qk@pk); lk = pk@(Qk@xk + g_tilde)
        alphak = 1.0
        if qk > 0 and lk < 0: alphak = min(-lk/qk, 1.0)
