"""ctypes binding of the CPU oracle (oracle/liblcqp_oracle.so).

Test infrastructure only: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package lcqpow_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
_SO = os.path.join(ROOT, "oracle", "liblcqp_oracle.so")

c_double_p = C.POINTER(C.c_double)


class Options(C.Structure):
    """Mirrors orc_options_t / lcqp_options_t (identical field order)."""
    _fields_ = [
        ("complementarityTolerance", C.c_double),
        ("stationarityTolerance", C.c_double),
        ("initialPenaltyParameter", C.c_double),
        ("penaltyUpdateFactor", C.c_double),
        ("maxPenaltyParameter", C.c_double),
        ("etaDynamicPenalty", C.c_double),
        ("solveZeroPenaltyFirst", C.c_int),
        ("perturbStep", C.c_int),
        ("maxIterations", C.c_int),
        ("nDynamicPenalty", C.c_int),
        ("printLevel", C.c_int),
        ("storeSteps", C.c_int),
        ("perturbSeed", C.c_uint64),
        ("admmRho", C.c_double),
        ("admmSigma", C.c_double),
        ("admmAlpha", C.c_double),
        ("rhoEqMult", C.c_double),
        ("proxSmall", C.c_double),
        ("proxBig", C.c_double),
        ("pivotThreshold", C.c_double),
        ("depTau", C.c_double),
        ("feasTol", C.c_double),
        ("resTol", C.c_double),
        ("admmFirst", C.c_int),
        ("admmHot", C.c_int),
        ("maxTrials", C.c_int),
        ("maxRounds", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("iterTotal", C.c_int), ("iterOuter", C.c_int), ("subproblemIter", C.c_int),
        ("status", C.c_int), ("qpSolverExitFlag", C.c_int), ("returnValue", C.c_int),
        ("rhoOpt", C.c_double),
        ("admmIter", C.c_int), ("trials", C.c_int), ("factorizations", C.c_int),
        ("corrections", C.c_int), ("qpSolves", C.c_int), ("reserved", C.c_int),
    ]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(ROOT, p)) > os.path.getmtime(_SO)
            for p in ("oracle/lcqp_oracle.c", "oracle/lcqp_oracle.h", "include/lcqp_synth.h")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_options_default.argtypes = [C.POINTER(Options)]
        L.orc_util_quadform.restype = C.c_double
        L.orc_util_dot.restype = C.c_double
        L.orc_util_maxabs.restype = C.c_double
        L.orc_qp_create.restype = C.c_void_p
        L.orc_qp_create.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.POINTER(Options)]
        L.orc_qp_destroy.argtypes = [C.c_void_p]
        L.orc_qp_solve.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)] + [c_double_p] * 7
        L.orc_qp_get_solution.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.orc_qp_get_counters.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.orc_lcqp_solve.argtypes = ([C.c_int] * 3 + [c_double_p] * 15 + [C.POINTER(Options), c_double_p, c_double_p,
                                     C.POINTER(Stats), C.c_int, c_double_p, c_double_p, C.POINTER(C.c_int)])
        L.orc_synth_generate.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int] + [c_double_p] * 7
        L.orc_synth_batch_solve.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Options),
                                            C.c_int, c_double_p, c_double_p, C.POINTER(Stats)]
        L.orc_synth_bench.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(Options),
                                      c_double_p, c_double_p, C.POINTER(Stats), c_double_p]
        L.orc_host_stream_gbps.restype = C.c_double
        L.orc_host_stream_gbps.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_size_t, C.c_int]
        _lib = L
    return _lib


def default_options(**kw):
    o = Options()
    lib().orc_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(c_double_p)


def _arr(a):
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


SEED0 = 0x4C43515000000001


def lcqp_set_robust(on):
    """dependent-row rules of the oracle's QP solver inside orc_lcqp_solve: on by default (every kernel carries them since round 2);
    0 / False selects the plain polish of round 1"""
    lib().orc_lcqp_set_robust(int(on))


def qp_set_sum_order(device_order):
    """summation order of E x in the oracle's QP solver: 1 = the device's (default), 0 = left to right (oracle/lcqp_oracle.c, dot_lanes)"""
    lib().orc_qp_set_sum_order(int(device_order))


def qp_set_trace(on):
    """one stderr line per round / trial of the oracle's QP solver (tools/fuzz_case.py --trace)"""
    lib().orc_qp_set_trace(int(on))


def qp_set_enter_cap(div):
    """cap on entering rows of a cold polish: max(n / div, 16) rows per trial (default 8, the device's value); 0 switches it off"""
    lib().orc_qp_set_enter_cap(int(div))


def synth_generate(instance, n=256, nC=512, nComp=64, seed0=SEED0):
    Q = np.empty((n, n)); g = np.empty(n); L = np.empty((nComp, n)); R = np.empty((nComp, n))
    A = np.empty((nC, n)); lbA = np.empty(nC); ubA = np.empty(nC)
    lib().orc_synth_generate(seed0, instance, n, nC, nComp, _p(Q), _p(g), _p(L), _p(R), _p(A), _p(lbA), _p(ubA))
    return dict(Q=Q, g=g, L=L, R=R, A=A, lbA=lbA, ubA=ubA, nV=n, nC=nC, nComp=nComp)


def lcqp_solve(Q, g, L, R, lbL=None, ubL=None, lbR=None, ubR=None, A=None, lbA=None, ubA=None, lb=None, ub=None,
               x0=None, y0=None, opt=None, trace=0, nV=None, nC=None, nComp=None):
    """Dense loadLCQP + runSolver on the oracle. Returns dict(ret, x, y, stats, trace?)."""
    Q = _arr(Q); g = _arr(g); L = _arr(L); R = _arr(R); A = _arr(A)
    nV = nV or g.shape[0]
    nComp = nComp or (L.size // nV)
    nC = nC if nC is not None else (0 if A is None else A.size // nV)
    arrs = [_arr(v) for v in (lbL, ubL, lbR, ubR)]
    lbA = _arr(lbA); ubA = _arr(ubA); lb = _arr(lb); ub = _arr(ub); x0 = _arr(x0); y0 = _arr(y0)
    opt = opt or default_options()
    x = np.zeros(nV); y = np.zeros(nV + nC + 2 * nComp)
    st = Stats()
    ts = np.zeros((max(trace, 1), 8)); tx = np.zeros((max(trace, 1), nV)); tl = C.c_int(0)
    ret = lib().orc_lcqp_solve(nV, nC, nComp, _p(Q), _p(g), _p(L), _p(R), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]),
                               _p(arrs[3]), _p(A), _p(lbA), _p(ubA), _p(lb), _p(ub), _p(x0), _p(y0), C.byref(opt),
                               _p(x), _p(y), C.byref(st), trace, _p(ts), _p(tx), C.byref(tl))
    out = dict(ret=ret, x=x, y=y, stats=st.asdict())
    if trace:
        out["trace_scalars"] = ts[:tl.value].copy()
        out["trace_x"] = tx[:tl.value].copy()
    return out


class QP:
    """SubsolverBase-shaped handle on the oracle QP solver."""

    def __init__(self, Q, A, opt=None):
        Q = _arr(Q); self.nV = Q.shape[0]
        A = _arr(A) if A is not None else np.zeros((0, self.nV))
        self.nC = A.size // self.nV
        self.opt = opt or default_options()
        self.h = lib().orc_qp_create(self.nV, self.nC, _p(Q), _p(A), C.byref(self.opt))

    def solve(self, initial, g, lbA=None, ubA=None, x0=None, y0=None, lb=None, ub=None):
        it = C.c_int(0); ef = C.c_int(0)
        a = [_arr(v) for v in (g, lbA, ubA, x0, y0, lb, ub)]
        ret = lib().orc_qp_solve(self.h, int(initial), C.byref(it), C.byref(ef), *[_p(v) for v in a])
        return ret, it.value, ef.value

    def solution(self):
        x = np.zeros(self.nV); y = np.zeros(self.nV + self.nC)
        lib().orc_qp_get_solution(self.h, _p(x), _p(y))
        return x, y

    def counters(self):
        v = [C.c_int(0) for _ in range(4)]
        lib().orc_qp_get_counters(self.h, *[C.byref(t) for t in v])
        return dict(admm=v[0].value, trials=v[1].value, factorizations=v[2].value, corrections=v[3].value)

    def __del__(self):
        try:
            lib().orc_qp_destroy(self.h)
        except Exception:
            pass


def synth_batch_solve(first, count, n=256, nC=512, nComp=64, opt=None, threads=1, seed0=SEED0, want_xy=True):
    opt = opt or default_options()
    nd = n + nC + 2 * nComp
    x = np.zeros((count, n)) if want_xy else None
    y = np.zeros((count, nd)) if want_xy else None
    st = (Stats * count)()
    ok = lib().orc_synth_batch_solve(seed0, first, count, n, nC, nComp, C.byref(opt), threads, _p(x), _p(y), st)
    return ok, x, y, [s.asdict() for s in st]


def host_cpu_topology():
    """(one cpu id per physical core, all cpu ids) of the cpus this process may run on, from the sibling lists in sysfs"""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores, seen = [], set()
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                sib += list(range(int(lo), int(hi or lo) + 1))
            key = tuple(sorted(sib))
        except OSError:
            key = (c,)
        if key not in seen:
            seen.add(key)
            cores.append(c)
    return cores, allowed


def l3_domains(cpus):
    """the given cpu ids grouped by the L3 cache they share (sysfs); one group when sysfs does not say"""
    doms = {}
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list") as f:
                key = f.read().strip()
        except OSError:
            key = "all"
        doms.setdefault(key, []).append(c)
    return list(doms.values())


def host_stream_gbps(cpus, mbytes_per_thread=256, reps=3):
    cp = (C.c_int * len(cpus))(*[int(c) for c in cpus])
    return float(lib().orc_host_stream_gbps(len(cpus), cp, int(mbytes_per_thread) << 20, reps))


def synth_bench(first, threads, per_thread, cpus=None, n=256, nC=512, nComp=64, opt=None, seed0=SEED0, want_xy=True):
    """steady-state CPU timing (orc_synth_bench): returns (solved, seconds, x, y, stats)"""
    opt = opt or default_options()
    count, nd = threads * per_thread, n + nC + 2 * nComp
    x = np.zeros((count, n)) if want_xy else None
    y = np.zeros((count, nd)) if want_xy else None
    st = (Stats * count)()
    sec = np.zeros(1)
    cp = (C.c_int * threads)(*[int(c) for c in cpus[:threads]]) if cpus is not None else None
    ok = lib().orc_synth_bench(seed0, first, threads, per_thread, cp, n, nC, nComp, C.byref(opt), _p(x), _p(y), st, _p(sec))
    return ok, float(sec[0]), x, y, [s.asdict() for s in st]


# ---- Utilities (orc_util_*) ---------------------------------------------------------------------
def util_matmul(A, B, m, n, p):
    A = _arr(A); B = _arr(B); Cm = np.zeros(m * p)
    lib().orc_util_matmul(_p(A), _p(B), _p(Cm), m, n, p)
    return Cm


def util_matmul_t(A, B, m, n, p):
    A = _arr(A); B = _arr(B); Cm = np.zeros(n * p)
    lib().orc_util_matmul_t(_p(A), _p(B), _p(Cm), m, n, p)
    return Cm


def util_symm_product(A, B, m, n):
    A = _arr(A); B = _arr(B); Cm = np.zeros(n * n)
    lib().orc_util_symm_product(_p(A), _p(B), _p(Cm), m, n)
    return Cm


def util_affine(alpha, A, b, c, m, n):
    A = _arr(A); b = _arr(b); c = _arr(c); d = np.zeros(m)
    lib().orc_util_affine(C.c_double(alpha), _p(A), _p(b), _p(c), _p(d), m, n)
    return d


def util_weighted_matadd(alpha, A, beta, B, m, n):
    A = _arr(A); B = _arr(B); Cm = np.zeros(m * n)
    lib().orc_util_weighted_matadd(C.c_double(alpha), _p(A), C.c_double(beta), _p(B), _p(Cm), m, n)
    return Cm


def util_weighted_vecadd(alpha, a, beta, b, m):
    a = _arr(a); b = _arr(b); c = np.zeros(m)
    lib().orc_util_weighted_vecadd(C.c_double(alpha), _p(a), C.c_double(beta), _p(b), _p(c), m)
    return c


def util_quadform(Q, p, m):
    Q = _arr(Q); p = _arr(p)
    return lib().orc_util_quadform(_p(Q), _p(p), m)


def util_dot(a, b, m):
    a = _arr(a); b = _arr(b)
    return lib().orc_util_dot(_p(a), _p(b), m)


def util_maxabs(a, m):
    a = _arr(a)
    return lib().orc_util_maxabs(_p(a), m)


# ---- CSC utilities (orc_csc_*) -------------------------------------------------------------------
class CSC(C.Structure):
    _fields_ = [("nzmax", C.c_int), ("m", C.c_int), ("n", C.c_int), ("p", C.POINTER(C.c_int)), ("i", C.POINTER(C.c_int)),
                ("x", c_double_p), ("nz", C.c_int)]


def _csc_setup():
    L = lib()
    if getattr(L, "_csc_ready", False):
        return L
    P = C.POINTER(CSC)
    L.orc_csc_create.restype = P
    L.orc_csc_create.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_csc_free.argtypes = [P]
    L.orc_csc_upper.restype = P; L.orc_csc_upper.argtypes = [P]
    L.orc_csc_to_dns.restype = c_double_p; L.orc_csc_to_dns.argtypes = [P]
    L.orc_dns_to_csc.restype = P; L.orc_dns_to_csc.argtypes = [c_double_p, C.c_int, C.c_int]
    for f in ("orc_csc_matmul", "orc_csc_matmul_t", "orc_csc_add_matmul_t"):
        getattr(L, f).argtypes = [P, c_double_p, c_double_p]
    L.orc_csc_symm_product.restype = P; L.orc_csc_symm_product.argtypes = [P, P]
    L.orc_csc_affine.argtypes = [C.c_double, P, c_double_p, c_double_p, c_double_p, C.c_int]
    L.orc_csc_quadform.restype = C.c_double; L.orc_csc_quadform.argtypes = [P, c_double_p, C.c_int]
    L._csc_ready = True
    return L


def csc_create(m, n, x, i, p):
    L = _csc_setup()
    x = _arr(x); i = np.ascontiguousarray(i, dtype=np.int32); p = np.ascontiguousarray(p, dtype=np.int32)
    return L.orc_csc_create(m, n, len(x), _p(x), i.ctypes.data_as(C.POINTER(C.c_int)), p.ctypes.data_as(C.POINTER(C.c_int)))


def csc_arrays(M):
    """(m, n, p, i, x) of a CSC* as numpy copies"""
    c = M.contents
    p = np.array([c.p[k] for k in range(c.n + 1)], dtype=np.int32)
    nnz = int(p[-1])
    return c.m, c.n, p, np.array([c.i[k] for k in range(nnz)], dtype=np.int32), np.array([c.x[k] for k in range(nnz)])


def csc_to_dns(M):
    L = _csc_setup()
    c = M.contents
    ptr = L.orc_csc_to_dns(M)
    out = np.array([ptr[k] for k in range(c.m * c.n)]).reshape(c.m, c.n)
    return out


def dns_to_csc(full):
    L = _csc_setup()
    full = _arr(full)
    return L.orc_dns_to_csc(_p(full), full.shape[0], full.shape[1])


def hessian_is_definite_by_diagonal(Qcsr):
    """the test that selects the ordering and the light regularisation of the sparse polish: min Q_ii >= 1e-6 max Q_ii"""
    d = np.asarray(Qcsr.diagonal(), dtype=float)
    return bool(d.size and d.max() > 0 and d.min() >= 1e-6 * np.abs(d).max())


def kkt_ordering(nV, Qp, Qi, Ep, Ei, wmax=63, kbmax=16, rows_follow=False):
    """Ordering of the KKT graph [Q E'; E .] with scipy (independent of the product's own analysis): reverse Cuthill-McKee; while the
    half bandwidth exceeds wmax, the node of highest degree moves to the border (the last positions).  rows_follow: the band is taken backwards and every row that
    would still be eliminated before all of its variables moves to just behind the first of them (the ordering the product uses for batches
    of safely definite Hessians, lcqp_hip_sparse_create); if that widens the band beyond wmax the plain ordering is kept.
    Returns (perm, w, kb): perm[position] = node (node < nV: variable, else row node - nV), w = half bandwidth of the first N - kb positions."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    m = len(Ep) - 1
    N = nV + m
    Qs = sp.csr_matrix((np.ones(len(Qi)), np.asarray(Qi), np.asarray(Qp)), shape=(nV, nV))
    Es = sp.csr_matrix((np.ones(len(Ei)), np.asarray(Ei), np.asarray(Ep)), shape=(m, nV))
    K = sp.csr_matrix(sp.bmat([[Qs, Es.T], [Es, None]], format="csr") + sp.identity(N, format="csr"))
    border = []

    def width(Kk, p):
        ip = np.empty(len(p), dtype=np.int64); ip[p] = np.arange(len(p))
        coo = Kk.tocoo()
        return int(np.abs(ip[coo.row] - ip[coo.col]).max()) if coo.nnz else 0

    while True:
        keep = np.setdiff1d(np.arange(N), border)
        Kk = sp.csr_matrix(K[keep][:, keep])
        p = np.asarray(reverse_cuthill_mckee(Kk, symmetric_mode=True), dtype=np.int64)
        w = width(Kk, p)
        if w <= wmax or len(border) >= kbmax:
            break
        deg = np.asarray(Kk.getnnz(axis=1)).ravel()
        border.append(int(keep[int(np.argmax(deg))]))
    if rows_follow:
        p = p[::-1].copy()      # reverse Cuthill-McKee leaves the multiplier nodes in front of their variables; the band width is the same backwards
        ip = np.empty(len(keep), dtype=np.int64); ip[p] = np.arange(len(keep))
        isrow = keep >= nV
        key = np.arange(len(keep), dtype=float)
        for pos in range(len(keep)):
            loc = p[pos]
            if not isrow[loc]:
                continue
            nb = Kk.indices[Kk.indptr[loc]:Kk.indptr[loc + 1]]
            nb = nb[~isrow[nb]]
            if nb.size and ip[nb].min() > pos:
                key[pos] = ip[nb].min() + 0.5
        p2 = p[np.argsort(key, kind="stable")]
        w2 = width(Kk, p2)
        if w2 <= wmax:
            p, w = p2, w2
    perm = np.concatenate([keep[p], np.array(border, dtype=np.int64)]).astype(np.int32)
    return perm, w, len(border)


def kkt_ordering_general(nV, Qp, Qi, Ep, Ei, leaf=48):
    """Fill-reducing ordering of the KKT graph for the oracle's general sparse LDL' (w = -1), written here with scipy -- independent of the
    product's own analysis: nested dissection by breadth-first level structures (George 1973: a pseudo-peripheral start, the middle level
    is the separator; parts of at most `leaf` nodes are numbered in their breadth-first order), sub-regions first, separators last.
    Returns perm[position] = node (node < nV: variable, else row node - nV)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order, connected_components
    m = len(Ep) - 1
    N = nV + m
    Qs = sp.csr_matrix((np.ones(len(Qi)), np.asarray(Qi), np.asarray(Qp)), shape=(nV, nV))
    Es = sp.csr_matrix((np.ones(len(Ei)), np.asarray(Ei), np.asarray(Ep)), shape=(m, nV))
    K = sp.csr_matrix(sp.bmat([[Qs, Es.T], [Es, None]], format="csr"))
    K = sp.csr_matrix(K + K.T)
    out = []

    def levels(sub, start):
        order, pred = breadth_first_order(sub, start, directed=False, return_predecessors=True)
        lev = np.full(sub.shape[0], -1)
        lev[start] = 0
        for v in order[1:]:
            lev[v] = lev[pred[v]] + 1
        return order, lev

    stack = [np.arange(N)]
    pieces = []          # (nodes, is_separator) in REVERSE elimination order
    while stack:
        nodes = stack.pop()
        if len(nodes) == 0:
            continue
        sub = sp.csr_matrix(K[nodes][:, nodes])
        ncomp, lab = connected_components(sub, directed=False)
        if ncomp > 1:
            for c in range(ncomp):
                stack.append(nodes[lab == c])
            continue
        if len(nodes) <= leaf:
            order, _ = levels(sub, 0)
            pieces.append(nodes[order])
            continue
        order, lev = levels(sub, 0)
        order, lev = levels(sub, int(order[-1]))          # restart from the far end: a pseudo-peripheral node
        mid = int(lev.max()) // 2
        if lev.max() < 2:                                   # a clique-like piece: no separator to be had
            pieces.append(nodes[order])
            continue
        sep = nodes[lev == mid]
        pieces.append(sep)                                  # eliminated after both sides
        stack.append(nodes[lev < mid]); stack.append(nodes[lev > mid])
    perm = np.concatenate(pieces[::-1]).astype(np.int32)
    assert len(perm) == N and len(np.unique(perm)) == N
    return perm


def sparse_lcqp_solve(nV, nC, nComp, Qcsr, g, Ecsr, lbA=None, ubA=None, lbL=None, ubL=None, lbR=None, ubR=None, x0=None, y0=None,
                      perm=None, w=None, kb=0, opt=None):
    """OSQP_SPARSE arm on the oracle (oracle/lcqp_oracle_sparse.c).  Qcsr / Ecsr: scipy CSR matrices (Q full symmetric,
    E = [A; L; R]).  Returns dict(ret, x, y (nC + 2 nComp), stats)."""
    L_ = lib()
    ip = C.POINTER(C.c_int)
    Qcsr = Qcsr.tocsr(); Ecsr = Ecsr.tocsr()
    Qcsr.sort_indices(); Ecsr.sort_indices()
    Qp = np.ascontiguousarray(Qcsr.indptr, dtype=np.int32); Qi = np.ascontiguousarray(Qcsr.indices, dtype=np.int32); Qx = _arr(Qcsr.data)
    Ep = np.ascontiguousarray(Ecsr.indptr, dtype=np.int32); Ei = np.ascontiguousarray(Ecsr.indices, dtype=np.int32); Ex = _arr(Ecsr.data)
    if perm is None:
        perm, w, kb = kkt_ordering(nV, Qp, Qi, Ep, Ei, rows_follow=hessian_is_definite_by_diagonal(Qcsr))
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    opt = opt or default_options()
    m = nC + 2 * nComp
    x = np.zeros(nV); y = np.zeros(m); st = Stats()
    a = [_arr(v) for v in (lbA, ubA, lbL, ubL, lbR, ubR, x0, y0)]
    L_.orc_sparse_lcqp_solve.restype = C.c_int
    ret = L_.orc_sparse_lcqp_solve(nV, nC, nComp, Qp.ctypes.data_as(ip), Qi.ctypes.data_as(ip), _p(Qx), _p(_arr(g)),
                                   Ep.ctypes.data_as(ip), Ei.ctypes.data_as(ip), _p(Ex), *[_p(v) for v in a],
                                   perm.ctypes.data_as(ip), C.c_int(int(w)), C.c_int(int(kb)), C.byref(opt), _p(x), _p(y), C.byref(st))
    return dict(ret=ret, x=x, y=y, stats=st.asdict(), perm=perm, w=int(w), kb=int(kb))
