import numpy as np, scipy.linalg as sla
import qp2, refprobs
np.set_printoptions(linewidth=200, precision=4)
d, x0 = refprobs.circle()
n, nC, nComp = d['n'], d['nC'], d['nComp']
As = np.vstack([d['A'], d['L'], d['R']])
lbA = np.concatenate([d['lbA'], np.zeros(2*nComp)]); ubA = np.concatenate([d['ubA'], np.full(2*nComp, np.inf)])
q = qp2.QP2(d['Q'], As, admm_first=50, admm_hot=2)
q.setup(lbA, ubA, np.full(n,-np.inf), np.full(n, np.inf))
# instrument pdas
orig = q.pdas
def pd(g, st):
    n, m, E, l, u = q.n, q.m, q.E, q.l, q.u
    idx = np.nonzero(st != 0)[0]; na=len(idx)
    print('na', na, 'rank E_act', np.linalg.matrix_rank(E[idx]))
    Eta = q.Et[idx]; S = Eta@Eta.T
    print('S diag range', np.diag(S).min(), np.diag(S).max(), 'eig', np.linalg.eigvalsh(S)[[0,1,-1]])
    r = orig(g, st); print('pdas ->', r[0], 'trials so far', q.stat['trials'], 'refine', q.stat['refine']); return r
q.pdas = pd
q.max_rounds = 3
x, y, info = q.solve(d['g'], x0, None)
print(info)
