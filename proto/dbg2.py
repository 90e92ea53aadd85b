import numpy as np, scipy.linalg as sla, sys
import qp2, refprobs, lcqp
np.set_printoptions(linewidth=200, precision=4)
which = sys.argv[1]
if which=='circle':
    d, x0 = refprobs.circle(); lb=ub=None
else:
    d, x0, lb, ub = refprobs.example_data()
class W:
    def __init__(s, Q, A):
        s.q = qp2.QP2(Q, A, admm_first=20, admm_hot=2); W.last = s.q
        orig = s.q.pdas; q=s.q
        def pd(g, st):
            t0=q.stat['trials']; r0=q.stat['refine']
            r = orig(g, st)
            if not r[0] and q.stat['rounds']<8: print('  pdas fail: trials', q.stat['trials']-t0, 'refine', q.stat['refine']-r0, 'na', q.stat['na'][-3:])
            return r
        q.pdas = pd
        q.max_rounds=8
    def setup(s, *a): s.q.setup(*a)
    def solve(s, g, x0=None, y0=None):
        n = s.q.n; yref = None
        if y0 is not None:
            m = s.q.mA; yref = -np.concatenate([y0[m:], y0[:m]])
        x, y, info = s.q.solve(g, x0, yref)
        print('QP', info)
        if x is None: return None, None, dict(status=info['status'], iters=0, polish=0)
        m = s.q.mA
        return x, np.concatenate([y[n:], y[:n]]), dict(status='solved', iters=0, polish=0)
r = lcqp.run_lcqp(d, W, x0=x0, lb=lb, ub=ub)
print(r['ret'])
