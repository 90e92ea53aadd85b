"""ctypes binding of the C ABI in include/lcqp_hip.h (liblcqpow_hip.so).  No CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LCQPOW_HIP_LIBRARY: another build of the same HIP library (experiment variants under build/ab/); there is no CPU fallback either way
_SO = os.environ.get("LCQPOW_HIP_LIBRARY") or os.path.join(_HERE, "liblcqpow_hip.so")
c_double_p = C.POINTER(C.c_double)

SEED0 = 0x4C43515000000001
SUCCESSFUL_RETURN = 0
SUBPROBLEM_SOLVER_ERROR = 203
MAX_ITERATIONS_REACHED = 200
MAX_PENALTY_REACHED = 201


class Options(C.Structure):
    """lcqp_options_t (include/lcqp_hip.h); same field order as the oracle's orc_options_t."""
    _fields_ = [
        ("complementarityTolerance", C.c_double), ("stationarityTolerance", C.c_double),
        ("initialPenaltyParameter", C.c_double), ("penaltyUpdateFactor", C.c_double),
        ("maxPenaltyParameter", C.c_double), ("etaDynamicPenalty", C.c_double),
        ("solveZeroPenaltyFirst", C.c_int), ("perturbStep", C.c_int), ("maxIterations", C.c_int),
        ("nDynamicPenalty", C.c_int), ("printLevel", C.c_int), ("storeSteps", C.c_int),
        ("perturbSeed", C.c_uint64),
        ("admmRho", C.c_double), ("admmSigma", C.c_double), ("admmAlpha", C.c_double), ("rhoEqMult", C.c_double),
        ("proxSmall", C.c_double), ("proxBig", C.c_double), ("pivotThreshold", C.c_double), ("depTau", C.c_double),
        ("feasTol", C.c_double), ("resTol", C.c_double),
        ("admmFirst", C.c_int), ("admmHot", C.c_int), ("maxTrials", C.c_int), ("maxRounds", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("iterTotal", C.c_int), ("iterOuter", C.c_int), ("subproblemIter", C.c_int), ("status", C.c_int),
        ("qpSolverExitFlag", C.c_int), ("returnValue", C.c_int), ("rhoOpt", C.c_double),
        ("admmIter", C.c_int), ("trials", C.c_int), ("factorizations", C.c_int), ("corrections", C.c_int),
        ("qpSolves", C.c_int), ("reserved", C.c_int),
    ]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_lib = None


def library_path():
    return _SO


def request_hw_queues(n=8):
    """Ask the HIP runtime for n hardware queues (GPU_MAX_HW_QUEUES) -- an explicit call of the PROGRAM (bench.py, the examples), before
    the first HIP call of the process; the library never changes the environment by itself.  Every batch object has two HIP streams and
    the runtime maps the streams of a process onto 4 queues by default: the two slots of a BatchPipeline can land on one queue and run one
    after the other.  Returns True when the variable was set, False when the environment already holds a value (left alone)."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return True


def lib():
    """Load liblcqpow_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} is missing: the HIP extension must be built (see __graft_entry__.build); "
                               "there is no CPU fallback for the product path")
        L = C.CDLL(_SO)
        L.lcqp_hip_last_error.restype = C.c_char_p
        L.lcqp_hip_options_default.argtypes = [C.POINTER(Options)]
        L.lcqp_hip_qp_create.restype = C.c_void_p
        L.lcqp_hip_qp_create.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.POINTER(Options), C.c_int]
        L.lcqp_hip_qp_clone.restype = C.c_void_p
        L.lcqp_hip_qp_clone.argtypes = [C.c_void_p]
        L.lcqp_hip_qp_destroy.argtypes = [C.c_void_p]
        L.lcqp_hip_qp_solve.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)] + [c_double_p] * 7
        L.lcqp_hip_qp_get_solution.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.lcqp_hip_qp_get_counters.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.lcqp_hip_batch_create.restype = C.c_void_p
        L.lcqp_hip_batch_create.argtypes = [C.c_int] * 6
        L.lcqp_hip_batch_destroy.argtypes = [C.c_void_p]
        L.lcqp_hip_batch_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        L.lcqp_hip_batch_set_overlapped.argtypes = [C.c_void_p, C.c_int]
        L.lcqp_hip_batch_load.argtypes = [C.c_void_p, C.c_int, C.c_int] + [c_double_p] * 15
        L.lcqp_hip_batch_generate_synthetic.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.lcqp_hip_batch_read_problem.argtypes = [C.c_void_p, C.c_int] + [c_double_p] * 7
        L.lcqp_hip_batch_setup.argtypes = [C.c_void_p]
        L.lcqp_hip_batch_run.argtypes = [C.c_void_p]
        L.lcqp_hip_batch_synchronize.argtypes = [C.c_void_p]
        L.lcqp_hip_batch_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.lcqp_hip_batch_get_solution.argtypes = [C.c_void_p, c_double_p, c_double_p, C.POINTER(Stats)]
        L.lcqp_hip_batch_get_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, C.POINTER(C.c_int)]
        L.lcqp_hip_batch_stream.restype = C.c_void_p
        L.lcqp_hip_batch_stream.argtypes = [C.c_void_p]
        L.lcqp_hip_batch_algorithmic_bytes.restype = C.c_double
        L.lcqp_hip_batch_algorithmic_bytes.argtypes = [C.c_void_p]
        L.lcqp_hip_batch_work_sums.argtypes = [C.c_void_p, c_double_p]
        L.lcqp_hip_util_symv.argtypes = [C.c_int, C.c_int, C.c_double] + [c_double_p] * 4
        L.lcqp_hip_util_gemv.argtypes = [C.c_int, C.c_int, C.c_int] + [c_double_p] * 3
        L.lcqp_hip_util_gemv_t.argtypes = [C.c_int, C.c_int, C.c_int] + [c_double_p] * 3
        L.lcqp_hip_util_symm_product.argtypes = [C.c_int, C.c_int, C.c_int] + [c_double_p] * 3
        L.lcqp_hip_util_rows_list.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, C.POINTER(C.c_int), C.c_int] + [c_double_p] * 4
        L.lcqp_hip_csc_create.restype = C.c_void_p
        L.lcqp_hip_csc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), c_double_p, C.c_int]
        L.lcqp_hip_csc_destroy.argtypes = [C.c_void_p]
        L.lcqp_hip_csc_apply.argtypes = [C.c_void_p, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, C.c_int, C.POINTER(C.c_float)]
        L.lcqp_hip_chol_solve.argtypes = [C.c_int, C.c_int] + [c_double_p] * 3 + [C.c_int, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def last_error():
    return lib().lcqp_hip_last_error().decode()


def device_count():
    return lib().lcqp_hip_device_count()


def default_options(**kw):
    o = Options()
    lib().lcqp_hip_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _arr(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}: {last_error()}")


def _sized(name, a, size):
    """the C side reads exactly `size` doubles through the raw pointer: refuse anything else here"""
    if a is not None and a.size != size:
        raise ValueError(f"{name}: expected {size} values, got {a.size} (shape {a.shape})")
    return a


class SubsolverHIP:
    """Python view of the SubsolverBase-shaped QP object (include/SubsolverBase.hpp:28-58)."""

    def __init__(self, nV, nC, Q, A, opt=None, device=0):
        Q = _sized("Q", _arr(Q), nV * nV); A = _sized("A", _arr(A), nC * nV)
        if Q is None or (nC > 0 and A is None):
            raise ValueError("Q (and A when nC > 0) must be given")
        self.nV, self.nC = nV, nC
        self.opt = opt or default_options()
        self.h = lib().lcqp_hip_qp_create(nV, nC, _p(Q), _p(A), C.byref(self.opt), device)
        if not self.h:
            raise RuntimeError("lcqp_hip_qp_create failed: " + last_error())

    def solve(self, initialSolve, g, lbA=None, ubA=None, x0=None, y0=None, lb=None, ub=None):
        it = C.c_int(0); ef = C.c_int(0)
        n, m = self.nV, self.nC
        a = [_sized(nm, _arr(v), sz) for nm, v, sz in (("g", g, n), ("lbA", lbA, m), ("ubA", ubA, m), ("x0", x0, n), ("y0", y0, n + m),
                                                       ("lb", lb, n), ("ub", ub, n))]
        ret = lib().lcqp_hip_qp_solve(self.h, int(bool(initialSolve)), C.byref(it), C.byref(ef), *[_p(v) for v in a])
        return ret, it.value, ef.value

    def getSolution(self):
        x = np.zeros(self.nV); y = np.zeros(self.nV + self.nC)
        lib().lcqp_hip_qp_get_solution(self.h, _p(x), _p(y))
        return x, y

    def counters(self):
        v = [C.c_int(0) for _ in range(4)]
        lib().lcqp_hip_qp_get_counters(self.h, *[C.byref(t) for t in v])
        return dict(admm=v[0].value, trials=v[1].value, factorizations=v[2].value, corrections=v[3].value)

    def close(self):
        if self.h:
            lib().lcqp_hip_qp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_mixed(problems, opt=None, device=0):
    """A list of LCQPs of DIFFERENT shapes in one call: the Python twin of LCQPow::MixedBatchLCQProblem (lcqpow_amd/csrc/host/BatchLCQProblem.hpp).
    problems: dicts with nV, nC, nComp, Q, g, L, R and optionally lbL, ubL, lbR, ubR, A, lbA, ubA, lb, ub, x0, y0 (the argument list of
    LCQProblem::loadLCQP).  They are sorted into buckets of equal (nV, nC, nComp, box bounds or not), one BatchLCQP per bucket created for
    exactly that shape -- every instance gets the bits of its solo run -- and bucket k + 1 is created, loaded and launched while bucket k
    runs.  Within a bucket instances may differ in which optional arguments they carry.  Returns one dict per problem, in input order:
    ret (the reference's ReturnValue), x, y, stats."""
    keys = ("lbL", "ubL", "lbR", "ubR", "A", "lbA", "ubA", "lb", "ub", "x0", "y0")
    buckets = {}
    for i, d in enumerate(problems):
        box = d.get("lb") is not None or d.get("ub") is not None
        buckets.setdefault((d["nV"], d["nC"], d["nComp"], box), []).append(i)
    out = [None] * len(problems)

    def collect(bt, members):
        x, y, st = bt.solution()
        for s_, i in enumerate(members):
            out[i] = dict(ret=st[s_]["returnValue"], x=x[s_], y=y[s_], stats=st[s_])
        bt.close()

    flying = None
    for (nV, nC, nComp, box), members in buckets.items():
        bt = BatchLCQP(len(members), nV, nC, nComp, with_box=box, device=device, opt=opt)
        if len(buckets) > 1: bt.set_overlapped(True)      # its setup runs beside the homotopy of the bucket launched before it
        failed = False
        for s_, i in enumerate(members):
            d = problems[i]
            rc = bt.load(s_, 1, d["Q"], d["g"], d["L"], d["R"], **{k: d.get(k) for k in keys})
            if rc != 0:      # as LCQProblem::loadLCQP: the code is the problem's result, the others of the bucket are not run with a hole among them
                for j in members:
                    out[j] = dict(ret=int(rc) if j == i else 300, x=None, y=None, stats=None)
                failed = True
                break
        if failed:
            bt.close()
            continue
        bt.run()                       # asynchronous on the bucket's own stream
        if flying is not None:
            collect(*flying)           # the bucket before ran while this one was created and loaded
        flying = (bt, members)
    if flying is not None:
        collect(*flying)
    return out


class BatchPipeline:
    """A stream of batches over `depth` BatchLCQP objects (each has its own HIP stream; run() only launches): the host-side twin of
    LCQPow::BatchPipeline (lcqpow_amd/csrc/host/BatchLCQProblem.hpp).  acquire() hands out the object to fill next --
    a free one, else the oldest one in flight after waiting for it (its results are then read with .solution())."""

    def __init__(self, depth, batch, nV, nC, nComp, with_box=False, device=0, opt=None, over=None):
        # over: existing BatchLCQP objects to run the pipeline over (not closed by close()).  HIP maps the streams of a process onto a few
        # hardware queues (4 by default, GPU_MAX_HW_QUEUES); every batch object has two streams, so a process that keeps more than two batch
        # objects alive can find both slots of a pipeline on one queue -- and its batches run one after the other (tools/micro/pipeline_check.py:
        # 35 200 LCQPs/s with two objects alive, 30 750 with an idle third one, 34 800 again with GPU_MAX_HW_QUEUES=8)
        if "GPU_MAX_HW_QUEUES" not in os.environ:
            import warnings
            warnings.warn("BatchPipeline with the HIP runtime's default of 4 hardware queues: with more batch objects alive than the pipeline's, "
                          "two slots can share a queue and run one after the other; call lcqpow_amd.request_hw_queues() before the first HIP "
                          "call of the process, or set GPU_MAX_HW_QUEUES", RuntimeWarning, stacklevel=2)
        self.owned = over is None
        self.slots = list(over) if over is not None else [BatchLCQP(batch, nV, nC, nComp, with_box=with_box, device=device, opt=opt) for _ in range(depth)]
        if len(self.slots) > 1:
            for bt in self.slots: bt.set_overlapped(True)      # the setup of one slot runs beside the homotopy of another
        self.state = [0] * len(self.slots)          # 0 free, 1 in flight, 2 finished
        self.order = []

    def acquire(self):
        # (a slot that was handed out with results and not launched again is free again; with nothing in flight every slot is)
        self.state = [0 if s == 2 else s for s in self.state]
        if not self.order:
            self.state = [0] * len(self.slots)
        for k, s in enumerate(self.state):
            if s == 0:
                return self.slots[k], False
        k = self.order.pop(0)
        self.slots[k].synchronize()
        self.state[k] = 2
        return self.slots[k], True        # (object, it carries a finished run)

    def launch(self, bt):
        k = self.slots.index(bt)
        bt.run()
        self.state[k] = 1
        self.order.append(k)

    def drain(self):
        """the batches still in flight, oldest first"""
        while self.order:
            k = self.order.pop(0)
            self.slots[k].synchronize()
            self.state[k] = 0
            yield self.slots[k]

    def close(self):
        if self.owned:
            for s in self.slots:
                s.close()
        elif len(self.slots) > 1:
            for s in self.slots:
                s.set_overlapped(False)      # the caller's objects run alone again


class BatchLCQP:
    """B independent dense LCQPs of one shape solved on one GPU (lcqp_hip_batch_*)."""

    def __init__(self, batch, nV, nC, nComp, with_box=False, device=0, opt=None):
        self.B, self.nV, self.nC, self.nComp = batch, nV, nC, nComp
        self.nd = nV + nC + 2 * nComp
        self.h = lib().lcqp_hip_batch_create(batch, nV, nC, nComp, int(with_box), device)
        if not self.h:
            raise RuntimeError("lcqp_hip_batch_create failed: " + last_error())
        if opt is not None:
            self.set_options(opt)

    def set_options(self, opt):
        _check(lib().lcqp_hip_batch_set_options(self.h, C.byref(opt)), "set_options")

    def set_overlapped(self, overlapped=True):
        """lcqp_hip_batch_set_overlapped: this object's setup runs beside another object's homotopy kernel (BatchPipeline sets it)."""
        _check(lib().lcqp_hip_batch_set_overlapped(self.h, 1 if overlapped else 0), "set_overlapped")

    def load(self, first, count, Q, g, L, R, lbL=None, ubL=None, lbR=None, ubR=None, A=None, lbA=None, ubA=None,
             lb=None, ub=None, x0=None, y0=None):
        """lcqp_hip_batch_load for instances [first, first + count).  Returns the reference's ReturnValue code (0, or e.g. 116
        INVALID_OBJECTIVE_LINEAR_TERM for g = None) like LCQProblem::loadLCQP does; a wrongly sized array raises ValueError."""
        n, nC, nK = self.nV, self.nC, self.nComp
        if first < 0 or count <= 0 or first + count > self.B:
            raise ValueError(f"instances [{first}, {first + count}) outside the batch of {self.B}")
        sizes = (("Q", Q, n * n), ("g", g, n), ("L", L, nK * n), ("R", R, nK * n), ("lbL", lbL, nK), ("ubL", ubL, nK), ("lbR", lbR, nK),
                 ("ubR", ubR, nK), ("A", A, nC * n), ("lbA", lbA, nC), ("ubA", ubA, nC), ("lb", lb, n), ("ub", ub, n), ("x0", x0, n),
                 ("y0", y0, self.nd))
        a = [_sized(nm, _arr(v), count * sz) for nm, v, sz in sizes]
        return lib().lcqp_hip_batch_load(self.h, first, count, *[_p(v) for v in a])

    def generate_synthetic(self, first_instance=0, seed0=SEED0):
        _check(lib().lcqp_hip_batch_generate_synthetic(self.h, seed0, first_instance), "generate_synthetic")

    def read_problem(self, b):
        n, nC, nComp = self.nV, self.nC, self.nComp
        Q = np.zeros((n, n)); g = np.zeros(n); L = np.zeros((nComp, n)); R = np.zeros((nComp, n))
        A = np.zeros((nC, n)); lbA = np.zeros(nC); ubA = np.zeros(nC)
        _check(lib().lcqp_hip_batch_read_problem(self.h, b, _p(Q), _p(g), _p(L), _p(R), _p(A), _p(lbA), _p(ubA)), "read_problem")
        return dict(Q=Q, g=g, L=L, R=R, A=A, lbA=lbA, ubA=ubA)

    def setup(self):
        _check(lib().lcqp_hip_batch_setup(self.h), "setup")

    def run(self):
        _check(lib().lcqp_hip_batch_run(self.h), "run")

    def synchronize(self):
        _check(lib().lcqp_hip_batch_synchronize(self.h), "synchronize")

    def last_timing(self):
        a = C.c_float(0); b = C.c_float(0)
        _check(lib().lcqp_hip_batch_last_timing(self.h, C.byref(a), C.byref(b)), "last_timing")
        return a.value, b.value

    def solution(self):
        x = np.zeros((self.B, self.nV)); y = np.zeros((self.B, self.nd))
        st = (Stats * self.B)()
        _check(lib().lcqp_hip_batch_get_solution(self.h, _p(x), _p(y), st), "get_solution")
        return x, y, [s.asdict() for s in st]

    def trace(self, instance, cap=1024):
        """per-iterate (|statk|inf, phi, rho, alphak, obj, merit, |pk|inf, QP iterations) and xk of one instance (needs options.storeSteps)"""
        sc = np.zeros((cap, 8)); xs = np.zeros((cap, self.nV)); n = C.c_int(0)
        _check(lib().lcqp_hip_batch_get_trace(self.h, instance, cap, _p(sc), _p(xs), C.byref(n)), "get_trace")
        return sc[:n.value].copy(), xs[:n.value].copy()

    def algorithmic_bytes(self):
        return lib().lcqp_hip_batch_algorithmic_bytes(self.h)

    def work_sums(self):
        """batch totals counted by the kernel: rows of Et read by the corrections, rows x slots over the corrections, bytes and number of
        the working-set updates, rows of E read by the residual sweeps, triangular solves with L1 (include/lcqp_hip.h)"""
        out = np.zeros(6)
        _check(lib().lcqp_hip_batch_work_sums(self.h, _p(out)), "work_sums")
        return out

    def stream(self):
        return lib().lcqp_hip_batch_stream(self.h)

    def close(self):
        if self.h:
            lib().lcqp_hip_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def util_symv(alpha, A, b, c):
    A = _arr(A); b = _arr(b); c = _arr(c)
    batch, n = A.shape[0], A.shape[1]
    d = np.zeros((batch, n))
    _check(lib().lcqp_hip_util_symv(batch, n, alpha, _p(A), _p(b), _p(c), _p(d)), "util_symv")
    return d


def util_gemv(A, b):
    A = _arr(A); b = _arr(b)
    batch, m, n = A.shape
    c = np.zeros((batch, m))
    _check(lib().lcqp_hip_util_gemv(batch, m, n, _p(A), _p(b), _p(c)), "util_gemv")
    return c


def util_gemv_t(A, b):
    A = _arr(A); b = _arr(b)
    batch, m, n = A.shape
    c = np.zeros((batch, n))
    _check(lib().lcqp_hip_util_gemv_t(batch, m, n, _p(A), _p(b), _p(c)), "util_gemv_t")
    return c


def util_rows_list(A, lists, x=None, coef=None, dots0=None):
    """wg_rows through a row list (lcqp_hip_util_rows_list): A [batch][m][n], lists [batch][nlist] -> (dots [batch][m], outT [batch][n])"""
    A = _arr(A); x = _arr(x); coef = _arr(coef)
    batch, m, n = A.shape
    lists = np.ascontiguousarray(lists, dtype=np.int32)
    dots = np.zeros((batch, m)) if dots0 is None else np.ascontiguousarray(dots0, dtype=np.float64).copy()
    out = np.zeros((batch, n))
    _check(lib().lcqp_hip_util_rows_list(batch, m, n, _p(A), lists.ctypes.data_as(C.POINTER(C.c_int)), lists.shape[1], _p(x), _p(coef),
                                         _p(dots) if x is not None else None, _p(out) if coef is not None else None), "util_rows_list")
    return dots, out


def util_symm_product(A, B):
    A = _arr(A); B = _arr(B)
    batch, m, n = A.shape
    Cm = np.zeros((batch, n, n))
    _check(lib().lcqp_hip_util_symm_product(batch, m, n, _p(A), _p(B), _p(Cm)), "util_symm_product")
    return Cm


def chol_solve(K, b, repeat=1):
    K = _arr(K); b = _arr(b)
    batch, n = K.shape[0], K.shape[1]
    x = np.zeros((batch, n)); ms = C.c_float(0)
    _check(lib().lcqp_hip_chol_solve(batch, n, _p(K), _p(b), _p(x), repeat, C.byref(ms)), "chol_solve")
    return x, ms.value


class CSCMatrix:
    """Device copy of a CSC matrix (and its transpose) for the sparse Utilities products."""

    def __init__(self, m, n, p, i, x, device=0):
        self.m, self.n = m, n
        p = np.ascontiguousarray(p, dtype=np.int32); i = np.ascontiguousarray(i, dtype=np.int32); x = _arr(x)
        ip = C.POINTER(C.c_int)
        self.h = lib().lcqp_hip_csc_create(m, n, len(x), p.ctypes.data_as(ip), i.ctypes.data_as(ip), _p(x), device)
        if not self.h:
            raise RuntimeError("lcqp_hip_csc_create failed: " + last_error())

    def apply(self, b, transposed=False, alpha=1.0, c=None, repeat=1):
        b = _arr(b); c = _arr(c)
        d = np.zeros(self.n if transposed else self.m); ms = C.c_float(0)
        _check(lib().lcqp_hip_csc_apply(self.h, int(transposed), alpha, _p(b), _p(c), _p(d), repeat, C.byref(ms)), "csc_apply")
        self.last_ms = ms.value
        return d

    def close(self):
        if self.h:
            lib().lcqp_hip_csc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseBatchLCQP:
    """B independent sparse LCQPs of one sparsity pattern on one GPU (lcqp_hip_sparse_*): the reference's OSQP_SPARSE arm.
    Qpat / Apat: scipy-like CSC pattern objects with .indptr / .indices (Q full symmetric nV x nV; A the stacked [A; L; R],
    (nC + 2 nComp) x nV).  Values are loaded per instance in the CSC order of these patterns."""

    def __init__(self, batch, nV, nC, nComp, Qpat, Apat, device=0, opt=None):
        ip = C.POINTER(C.c_int)
        L = lib()
        L.lcqp_hip_sparse_create.restype = C.c_void_p
        L.lcqp_hip_sparse_create.argtypes = [C.c_int] * 4 + [ip] * 4 + [C.c_int]
        L.lcqp_hip_sparse_last_error.restype = C.c_char_p
        L.lcqp_hip_sparse_destroy.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_bandwidth.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_lanes.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_border.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_fronts.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_get_ordering.argtypes = [C.c_void_p, ip]
        L.lcqp_hip_sparse_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        L.lcqp_hip_sparse_load.argtypes = [C.c_void_p, C.c_int, C.c_int] + [c_double_p] * 11
        L.lcqp_hip_sparse_run.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_synchronize.argtypes = [C.c_void_p]
        L.lcqp_hip_sparse_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.lcqp_hip_sparse_get_solution.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_void_p]
        L.lcqp_hip_sparse_get_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, C.POINTER(C.c_int)]
        L.lcqp_hip_sparse_algorithmic_bytes.restype = C.c_double
        L.lcqp_hip_sparse_algorithmic_bytes.argtypes = [C.c_void_p]
        self.B, self.nV, self.nC, self.nComp, self.m = batch, nV, nC, nComp, nC + 2 * nComp
        self._pat = [np.ascontiguousarray(a, dtype=np.int32) for a in (Qpat.indptr, Qpat.indices, Apat.indptr, Apat.indices)]
        if self._pat[0].size != nV + 1 or self._pat[2].size != nV + 1:
            raise ValueError("pattern column pointers must have nV + 1 entries (CSC)")
        if self._pat[1].size != int(self._pat[0][-1]) or self._pat[3].size != int(self._pat[2][-1]):
            raise ValueError("pattern index arrays must have indptr[-1] entries")        # the C side reads exactly that many
        self.nnzQ, self.nnzA = int(self._pat[0][-1]), int(self._pat[2][-1])
        self.h = L.lcqp_hip_sparse_create(batch, nV, nC, nComp, *[a.ctypes.data_as(ip) for a in self._pat], device)
        if not self.h:
            raise RuntimeError("lcqp_hip_sparse_create failed: " + L.lcqp_hip_sparse_last_error().decode())
        if opt is not None:
            self.set_options(opt)

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with code {rc}: {lib().lcqp_hip_sparse_last_error().decode()}")

    def bandwidth(self):
        return lib().lcqp_hip_sparse_bandwidth(self.h)

    def lanes(self):
        """lanes of a wavefront per instance (8, 16, 32 or 64: the smallest above the half bandwidth)"""
        return lib().lcqp_hip_sparse_lanes(self.h)

    def border(self):
        """border nodes of the bordered band: the last positions of ordering() (0: plain band)"""
        return lib().lcqp_hip_sparse_border(self.h)

    def fronts(self):
        """fronts of the general sparse LDL' (0: one of the band engines runs this pattern)"""
        return lib().lcqp_hip_sparse_fronts(self.h)

    def ordering(self):
        perm = np.zeros(self.nV + self.m, dtype=np.int32)
        self._chk(lib().lcqp_hip_sparse_get_ordering(self.h, perm.ctypes.data_as(C.POINTER(C.c_int))), "get_ordering")
        return perm

    def set_options(self, opt):
        self._chk(lib().lcqp_hip_sparse_set_options(self.h, C.byref(opt)), "set_options")

    def load(self, first, count, Qx, g, Ax, lbA=None, ubA=None, lbL=None, ubL=None, lbR=None, ubR=None, x0=None, y0=None):
        n, nC, nK = self.nV, self.nC, self.nComp
        sizes = (("Qx", Qx, self.nnzQ), ("g", g, n), ("Ax", Ax, self.nnzA), ("lbA", lbA, nC), ("ubA", ubA, nC), ("lbL", lbL, nK), ("ubL", ubL, nK),
                 ("lbR", lbR, nK), ("ubR", ubR, nK), ("x0", x0, n), ("y0", y0, self.m))
        a = [_sized(nm, _arr(v), count * sz) for nm, v, sz in sizes]
        return lib().lcqp_hip_sparse_load(self.h, first, count, *[_p(v) for v in a])

    def run(self):
        self._chk(lib().lcqp_hip_sparse_run(self.h), "run")

    def synchronize(self):
        self._chk(lib().lcqp_hip_sparse_synchronize(self.h), "synchronize")

    def last_timing(self):
        a = C.c_float(0); b = C.c_float(0)
        self._chk(lib().lcqp_hip_sparse_last_timing(self.h, C.byref(a), C.byref(b)), "last_timing")
        return a.value, b.value

    def solution(self):
        x = np.zeros((self.B, self.nV)); y = np.zeros((self.B, self.m))
        st = (Stats * self.B)()
        self._chk(lib().lcqp_hip_sparse_get_solution(self.h, _p(x), _p(y), st), "get_solution")
        return x, y, [s.asdict() for s in st]

    def trace(self, instance, cap=1024):
        """per-iterate (|statk|inf, phi, rho, alphak, obj, merit, |pk|inf, QP iterations) and xk of one instance (needs options.storeSteps)"""
        sc = np.zeros((cap, 8)); xs = np.zeros((cap, self.nV)); n = C.c_int(0)
        self._chk(lib().lcqp_hip_sparse_get_trace(self.h, instance, cap, _p(sc), _p(xs), C.byref(n)), "get_trace")
        return sc[:n.value].copy(), xs[:n.value].copy()

    def algorithmic_bytes(self):
        return lib().lcqp_hip_sparse_algorithmic_bytes(self.h)

    def close(self):
        if self.h:
            lib().lcqp_hip_sparse_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
