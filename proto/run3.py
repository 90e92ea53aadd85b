import numpy as np, time, sys
import synth, qp, lcqp
trials=[]
class S(qp.QPADMM):
    kw = {}
    def solve(s, g, x0=None, y0=None, **k):
        return super().solve(g, x0, y0, **S.kw)
    def polish(s, g, x, y, z, tol=1e-9):
        # count trials by wrapping lu_factor
        import scipy.linalg as sla
        cnt=[0]; orig=sla.lu_factor
        def f(K):
            cnt[0]+=1; s.lastsize=K.shape[0]; return orig(K)
        sla.lu_factor=f
        try: r = super().polish(g,x,y,z,tol)
        finally: sla.lu_factor=orig
        trials.append((cnt[0], s.lastsize - s.n, r[0]))
        return r
for ce in [1,2,5]:
    S.kw = dict(eps=1e9, check_every=ce)
    for inst in range(3):
        trials.clear()
        d = synth.gen(inst)
        r = lcqp.run_lcqp(d, lambda Q,A: S(Q,A,rho=0.3))
        print(ce, inst, r['ret'], r.get('total'), r.get('qpit'), 'trials', [t[0] for t in trials], 'nact', [t[1] for t in trials][:5], trials[-1][1])
