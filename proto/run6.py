import numpy as np, time, sys
import synth, qp2, lcqp, refprobs
def run(kw, probs):
    class W:
        def __init__(s, Q, A): s.q = qp2.QP2(Q, A, **kw); W.last = s.q
        def setup(s, *a): s.q.setup(*a)
        def solve(s, g, x0=None, y0=None):
            n = s.q.n; yref = None
            if y0 is not None:
                m = s.q.mA; yref = -np.concatenate([y0[m:], y0[:m]])
            x, y, info = s.q.solve(g, x0, yref)
            if x is None: return None, None, dict(status=info['status'], iters=0, polish=0)
            return x, np.concatenate([y[n:], y[:n]]), dict(status='solved', iters=0, polish=0)
    out=[]
    for name in probs:
        t=time.time()
        if name=='circle': d,x0=refprobs.circle(); r=lcqp.run_lcqp(d,W,x0=x0)
        elif name=='exdata': d,x0,lb,ub=refprobs.example_data(); r=lcqp.run_lcqp(d,W,x0=x0,lb=lb,ub=ub)
        else: d=synth.gen(int(name)); r=lcqp.run_lcqp(d,W)
        st=W.last.stat
        out.append((name, r['ret'][:4], r.get('total'), st['admm'], st['trials'], st.get('fact'), st.get('ndep'), round(time.time()-t,1), None if r.get('x') is None else np.round(r['x'][:2],4)))
    return out
if __name__=='__main__':
    for sp in [1e-8, 1e-6, 1e-5, 1e-4]:
        for delta in [1e-12, 1e-9]:
            kw=dict(admm_first=20, admm_hot=2, sp=sp, delta=delta, max_rounds=20)
            print(sp, delta, run(kw, ['circle','exdata','0']))
