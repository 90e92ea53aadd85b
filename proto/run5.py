import numpy as np, time, sys
import synth, qp2, lcqp, refprobs
kw = dict(admm_first=int(sys.argv[1]), admm_hot=int(sys.argv[2]))
class W:
    def __init__(s, Q, A): s.q = qp2.QP2(Q, A, **kw); W.last = s.q
    def setup(s, *a): s.q.setup(*a)
    def solve(s, g, x0=None, y0=None):
        # lcqp.py passes y0 in internal order [rows; box] with flipped sign: undo
        n = s.q.n; yref = None
        if y0 is not None:
            m = s.q.mA; yref = -np.concatenate([y0[m:], y0[:m]])
        x, y, info = s.q.solve(g, x0, yref)
        if x is None: return None, None, dict(status=info['status'], iters=0, polish=0)
        m = s.q.mA
        return x, np.concatenate([y[n:], y[:n]]), dict(status='solved', iters=0, polish=0)
def rep(name, r, t0):
    st = W.last.stat
    print(name, r['ret'], 'iter', r.get('total'), 'outer', r.get('outer'), 'rho', r.get('rho'), 'qps', r.get('qps'), 'admm', st['admm'], 'trials', st['trials'], 'refine', st['refine'], 'rounds', st['rounds'], 'na', st['na'][-1] if st['na'] else None, 't %.2f' % (time.time()-t0))
    return r
f = W
t=time.time(); r = rep('warm_up', lcqp.run_lcqp(refprobs.warm_up(), f), t); print('   ', r.get('x'), r.get('y'))
t=time.time(); r = rep('w_A', lcqp.run_lcqp(refprobs.warm_up(variant='w_A'), f), t); print('   ', r.get('x'), r.get('y'))
t=time.time(); r = rep('binary', lcqp.run_lcqp(refprobs.warm_up(variant='binary'), f, x0=np.zeros(2)), t); print('   ', r.get('x'), r.get('y'))
d, x0 = refprobs.circle(); t=time.time(); r = rep('circle', lcqp.run_lcqp(d, f, x0=x0), t); print('   ', None if r.get('x') is None else r['x'][:2])
d, x0, lb, ub = refprobs.example_data(); t=time.time(); r = rep('exdata', lcqp.run_lcqp(d, f, x0=x0, lb=lb, ub=ub), t)
for inst in range(3):
    d = synth.gen(inst); t=time.time(); r = rep('synth%d'%inst, lcqp.run_lcqp(d, f), t)
