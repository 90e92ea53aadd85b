import numpy as np, scipy.linalg as sla
from hardqp import mk
import io, contextlib
Q,A,l,u,g = mk(256,640,4); n,m=256,640
def run(rho0=0.1, sigma=1e-6, alpha=1.6, eps=1e-6):
    rhov = np.full(m, rho0); eq = l==u; rhov[eq]*=1e3
    F = sla.cho_factor(Q + sigma*np.eye(n) + (A.T*rhov)@A); x=np.zeros(n); z=np.clip(A@x,l,u); y=np.zeros(m)
    n_admm=10; it=0
    for rnd in range(12):
        yprev = y.copy()
        for _ in range(n_admm):
            xt = sla.cho_solve(F, sigma*x - g + A.T@(rhov*z - y)); zt = A@xt
            xn = alpha*xt+(1-alpha)*x; zr = alpha*zt+(1-alpha)*z
            zn = np.clip(zr + y/rhov, l, u); y = y + rhov*(zr-zn); x, z = xn, zn; it+=1
        dy = y - yprev; nrm = np.abs(dy).max()
        if nrm > 0:
            c1 = np.abs(A.T@dy).max() <= eps*nrm
            lu = np.where(dy>0, np.where(np.isfinite(u),u,0)*dy, 0).sum() + np.where(dy<0, np.where(np.isfinite(l),l,0)*dy, 0).sum()
            badinf = ((dy>eps*nrm)&~np.isfinite(u)).any() or ((dy<-eps*nrm)&~np.isfinite(l)).any()
            print(f"round {rnd} it {it} |A'dy|/|dy| {np.abs(A.T@dy).max()/nrm:.2e} support {lu/nrm:.3e} badinf {badinf}")
            if c1 and lu < -eps*nrm and not badinf: print("  -> primal infeasible certificate"); return
        n_admm = min(max(2*n_admm,10),400)
run()
