import numpy as np
MASK = (1<<64)-1
GAMMA = 0x9E3779B97F4A7C15
SEED0 = 0x4C43515000000001

def sm64(state, k):
    """k-th (0-based) SplitMix64 output for initial state `state` (vectorised over k)."""
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(state) + (k + np.uint64(1)) * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z

def u01(state, k):
    return (sm64(state, k) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

def gen(inst, n=256, nC=512, nComp=64, seed0=SEED0):
    st = (seed0 ^ inst) & MASK
    off = 0
    def take(cnt):
        nonlocal off
        v = u01(st, np.arange(off, off+cnt, dtype=np.uint64)); off += cnt
        return v
    M = (2*take(n*n)-1).reshape(n, n)
    g = 2*take(n)-1
    coin = take(nComp)
    xs = 2*take(n)-1
    A = ((2*take(nC*n)-1)/np.sqrt(n)).reshape(nC, n)
    slo = 0.1 + 0.9*take(nC)
    shi = 0.1 + 0.9*take(nC)
    Q = M.T @ M / n + np.eye(n)
    L = np.zeros((nComp, n)); R = np.zeros((nComp, n))
    L[np.arange(nComp), np.arange(nComp)] = 1
    R[np.arange(nComp), nComp+np.arange(nComp)] = 1
    # feasible point: pair i -> (x_i, x_{nComp+i}); coin<0.5: L side zero
    xl = 0.5*(xs[:nComp]+1); xr = 0.5*(xs[nComp:2*nComp]+1)   # U(0,1)
    xs = xs.copy()
    xs[:nComp] = np.where(coin < 0.5, 0.0, xl)
    xs[nComp:2*nComp] = np.where(coin < 0.5, xr, 0.0)
    Ax = A @ xs
    lbA = Ax - slo; ubA = Ax + shi
    return dict(Q=Q, g=g, L=L, R=R, A=A, lbA=lbA, ubA=ubA, xstar=xs, n=n, nC=nC, nComp=nComp)
