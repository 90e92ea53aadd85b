import numpy as np, scipy.linalg as sla, sys
import qp2, refprobs, lcqp
np.set_printoptions(linewidth=200, precision=4)
which = sys.argv[1]
if which=='circle':
    d, x0 = refprobs.circle(); n=d['n']; lb=np.full(n,-np.inf); ub=np.full(n,np.inf)
else:
    d, x0, lb, ub = refprobs.example_data(); n=d['n']
nC,nComp=d['nC'],d['nComp']
As = np.vstack([d['A'], d['L'], d['R']])
lbL = d.get('lbL', np.zeros(nComp)); lbR = d.get('lbR', np.zeros(nComp)); ubL=d.get('ubL', np.full(nComp,np.inf)); ubR=d.get('ubR', np.full(nComp,np.inf))
lbA = np.concatenate([d['lbA'], lbL, lbR]); ubA = np.concatenate([d['ubA'], ubL, ubR])
q = qp2.QP2(d['Q'], As, admm_first=int(sys.argv[2]), admm_hot=2)
q.setup(lbA, ubA, lb, ub)
print('n', n, 'm', q.m, 'eq rows', (q.l==q.u).sum(), 'spv', q.spv, 'Q diag', np.diag(d['Q']).min(), np.diag(d['Q']).max())
# verbose pdas
def pdas(g, st, xc):
    s=q
    n, m, E, l, u = s.n, s.m, s.E, s.l, s.u
    gs = 1 + np.abs(g).max(); tol=s.tol
    xprev = xc.copy(); yprev=None; prev_idx=None
    for trial in range(s.max_trials):
        idx = np.nonzero(st != 0)[0]; na = len(idx)
        b = np.where(st[idx] == 2, u[idx], l[idx])
        Ea, Eta = E[idx], s.Et[idx]
        S = Eta @ Eta.T
        dlt=0.0
        LS, ndep = qp2.safe_chol(S, s.delta)
        if ndep: print('   ndep', ndep)
        yprev = np.zeros(na) if (prev_idx is None or len(prev_idx)!=na or (prev_idx!=idx).any()) else yprev
        r1 = -g + s.spv*xprev; r2 = b - dlt*yprev
        c = sla.solve_triangular(s.L1, r1, lower=True)
        y = sla.cho_solve((LS, True), Eta @ c - r2)
        x = sla.solve_triangular(s.L1.T, c - Eta.T @ y, lower=False)
        Ex = E@x; yfull=np.zeros(m); yfull[idx]=y
        res_stat = s.spv*np.abs(x-xprev).max(); res_eq=np.abs(Ex[idx]-b).max(initial=0)
        true_stat = np.abs(s.Q@x+g+Ea.T@y).max()
        newst=st.copy(); ftol=tol*(1+np.abs(Ex))
        a1=(st==0)&(Ex<l-ftol); a2=(st==0)&(Ex>u+ftol); ytol=tol*gs
        d1=(st==1)&(yfull>ytol); d2=(st==2)&(yfull<-ytol)
        newst[a1]=1; newst[a2]=2; newst[d1|d2]=0
        print('   trial', trial, 'na', na, 'res_stat %.2e true %.2e res_eq %.2e'%(res_stat,true_stat,res_eq), 'add', a1.sum()+a2.sum(), 'drop', (d1|d2).sum(), '|x|', np.abs(x).max(), '|y|', np.abs(y).max(initial=0), 'maxviol', max((l-Ex).max(), (Ex-u).max()))
        if not (newst!=st).any() and res_stat<=s.tol_res*gs and res_eq<=s.tol_res*(1+np.abs(b).max(initial=0)): return True,x,yfull
        prev_idx=idx; xprev=x; yprev=y; st=newst
    return False,None,None
q.pdas=pdas; q.max_rounds=int(sys.argv[3]) if len(sys.argv)>3 else 3
x,y,info=q.solve(d['g'], x0, None); print(info)
