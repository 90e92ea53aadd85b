import numpy as np, time, sys
import synth, qp, lcqp
class S(qp.QPADMM):
    kw = {}
    def solve(s, g, x0=None, y0=None, **k):
        return super().solve(g, x0, y0, **S.kw)
for cfg in [dict(eps=1e-3, check_every=10), dict(eps=1e9, check_every=10), dict(eps=1e9, check_every=5), dict(eps=1e9, check_every=25)]:
  for rho in [0.1, 0.3, 1.0]:
    S.kw = cfg
    tot = []
    for inst in range(2):
        d = synth.gen(inst)
        t=time.time()
        r = lcqp.run_lcqp(d, lambda Q,A: S(Q,A,rho=rho))
        tot.append((r['ret'], r.get('total'), r.get('qpit'), r.get('polish'), round(time.time()-t,1)))
    print(cfg, rho, tot)
