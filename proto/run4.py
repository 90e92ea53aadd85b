import numpy as np, time, sys
import qp, lcqp, refprobs
class S(qp.QPADMM):
    kw = dict(eps=1e9, check_every=int(sys.argv[1]) if len(sys.argv)>1 else 10)
    def solve(s, g, x0=None, y0=None, **k):
        return super().solve(g, x0, y0, **S.kw)
f = lambda Q,A: S(Q,A,rho=0.3)
class O(lcqp.Opt): pass
for name, d, kw in [('warm_up x0=0', refprobs.warm_up(), {}), ('warm_up x0=(1,1)', refprobs.warm_up(), dict(x0=np.ones(2), y0=np.zeros(4))),
                ('w_A', refprobs.warm_up(variant='w_A'), {}), ('binary', refprobs.warm_up(variant='binary'), dict(x0=np.zeros(2)))]:
    r = lcqp.run_lcqp(d, f, **kw)
    print(name, r['ret'], r.get('x'), r.get('y'), r.get('total'), r.get('outer'), r.get('rho'), r.get('qpit'))
d, x0 = refprobs.circle()
t=time.time(); r = lcqp.run_lcqp(d, f, x0=x0); print('circle', r['ret'], None if r.get('x') is None else r['x'][:2], r.get('total'), r.get('outer'), r.get('rho'), r.get('qpit'), r.get('polish'), time.time()-t, r.get('info'))
d, x0, lb, ub = refprobs.example_data()
t=time.time(); r = lcqp.run_lcqp(d, f, x0=x0, lb=lb, ub=ub); print('exdata', r['ret'], r.get('total'), r.get('outer'), r.get('rho'), r.get('qpit'), r.get('polish'), time.time()-t, r.get('info'))
