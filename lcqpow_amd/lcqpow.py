"""The reference's Python surface (``import lcqpow``) on the MI355X backend:  ``import lcqpow_amd.lcqpow as lcqpow``.

Same classes, method names, keyword arguments and return codes as the pybind11 module of the reference
(interfaces/python/lcqpow/LCQProblem.cpp:70-176, Options.cpp:12-43, OutputStatistics.cpp:14-31,
Utilities.cpp:12-77), bound with ctypes to the C ABI of include/lcqp_host.h (liblcqpow_host.so, which drives
liblcqpow_hip.so).  No CPU fallback: importing this module without the built libraries raises.

Matrix layout.  The reference converts dense inputs to ``Eigen::MatrixXd`` (column-major) and hands ``.data()``
to a C++ core that reads row-major (src/Utilities.cpp:43), so its Python callers pass ``L.T, R.T, A.T``
(interfaces/python/examples/OptimizeOnCircle.py:76).  ``loadLCQP`` here reproduces that by default
(``order="F"``: the array's column-major element order is what the solver reads as row-major), so scripts
written for the reference run unchanged; pass ``order="C"`` to hand over ordinary row-major
``(nComp, nV)`` / ``(nC, nV)`` arrays without transposing.
"""
import ctypes as C
import enum
import os

import numpy as np

from . import capi as _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblcqpow_host.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class ReturnValue(enum.IntEnum):
    """include/Utilities.hpp:37-87"""
    NOT_YET_IMPLEMENTED = -1
    SUCCESSFUL_RETURN = 0
    INVALID_ARGUMENT = 100
    INVALID_PENALTY_UPDATE_VALUE = 101
    INVALID_COMPLEMENTARITY_TOLERANCE = 102
    INVALID_INITIAL_PENALTY_VALUE = 103
    INVALID_MAX_ITERATIONS_VALUE = 104
    INVALID_STATIONARITY_TOLERANCE = 105
    INVALID_NUMBER_OF_OPTIM_VARS = 106
    INVALID_NUMBER_OF_COMP_VARS = 107
    INVALID_NUMBER_OF_CONSTRAINT_VARS = 108
    INVALID_QPSOLVER = 109
    INVALID_OSQP_BOX_CONSTRAINTS = 110
    INVALID_TOTAL_ITER_COUNT = 111
    INVALID_TOTAL_OUTER_ITER = 112
    IVALID_SUBPROBLEM_ITER = 113
    INVALID_RHO_OPT = 114
    INVALID_PRINT_LEVEL_VALUE = 115
    INVALID_OBJECTIVE_LINEAR_TERM = 116
    INVALID_CONSTRAINT_MATRIX = 117
    INVALID_COMPLEMENTARITY_MATRIX = 118
    INVALID_ETA_VALUE = 119
    INVALID_LOWER_COMPLEMENTARITY_BOUND = 120
    INVALID_MAX_RHO_VALUE = 121
    MAX_ITERATIONS_REACHED = 200
    MAX_PENALTY_REACHED = 201
    INITIAL_SUBPROBLEM_FAILED = 202
    SUBPROBLEM_SOLVER_ERROR = 203
    FAILED_SYM_COMPLEMENTARITY_MATRIX = 204
    FAILED_SWITCH_TO_SPARSE = 205
    FAILED_SWITCH_TO_DENSE = 206
    LCQPOBJECT_NOT_SETUP = 300
    INDEX_OUT_OF_BOUNDS = 301
    UNABLE_TO_READ_FILE = 302
    DENSE_SPARSE_MISSMATCH = 402


class AlgorithmStatus(enum.IntEnum):
    """include/Utilities.hpp:103-109"""
    PROBLEM_NOT_SOLVED = 0
    W_STATIONARY_SOLUTION = 1
    C_STATIONARY_SOLUTION = 2
    M_STATIONARY_SOLUTION = 3
    S_STATIONARY_SOLUTION = 4


class PrintLevel(enum.IntEnum):
    """include/Utilities.hpp:115-119"""
    NONE = 0
    OUTER_LOOP_ITERATES = 1
    INNER_LOOP_ITERATES = 2


class QPSolver(enum.IntEnum):
    """include/Utilities.hpp:125-129 plus the backend this build adds.  Every value runs on the HIP subsolver (qpOASES
    and OSQP are not vendored); the three reference values keep their contracts from src/LCQProblem.cpp:888-963:
    QPOASES_DENSE needs dense problem data and QPOASES_SPARSE / OSQP_SPARSE sparse data (DENSE_SPARSE_MISSMATCH
    otherwise), OSQP_SPARSE refuses box constraints (INVALID_OSQP_BOX_CONSTRAINTS) and returns nC + 2 nComp duals
    without the box part; HIP_DENSE takes the problem in either mode."""
    QPOASES_DENSE = 0
    QPOASES_SPARSE = 1
    OSQP_SPARSE = 2
    HIP_DENSE = 3


def _export(enum_cls):
    # pybind11's export_values(): enum members are also module attributes (lcqpow.SUCCESSFUL_RETURN)
    globals().update(enum_cls.__members__)


for _e in (ReturnValue, AlgorithmStatus, PrintLevel, QPSolver):
    _export(_e)


class _Stats(C.Structure):
    _fields_ = [("iterTotal", C.c_int), ("iterOuter", C.c_int), ("subproblemIter", C.c_int), ("status", C.c_int),
                ("qpSolverExitFlag", C.c_int), ("nSteps", C.c_int), ("rhoOpt", C.c_double)]


class _CscArg(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("nnz", C.c_int), ("x", _dp), ("i", _ip), ("p", _ip)]


_lib = None


def _host():
    """liblcqpow_host.so (C ABI of include/lcqp_host.h); loads liblcqpow_hip.so first so its absence is reported as such"""
    global _lib
    if _lib is None:
        _capi.lib()
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.lcqp_host_options_create.restype = vp
        L.lcqp_host_options_copy.restype = vp
        L.lcqp_host_options_copy.argtypes = [vp]
        L.lcqp_host_options_destroy.argtypes = [vp]
        L.lcqp_host_options_set_to_default.argtypes = [vp]
        L.lcqp_host_options_set.argtypes = [vp, C.c_int, C.c_double]
        L.lcqp_host_options_get.restype = C.c_double
        L.lcqp_host_options_get.argtypes = [vp, C.c_int]
        L.lcqp_host_options_get_hip.argtypes = [vp, C.POINTER(_capi.Options)]
        L.lcqp_host_options_set_hip.argtypes = [vp, C.POINTER(_capi.Options)]
        L.lcqp_host_problem_create.restype = vp
        L.lcqp_host_problem_create.argtypes = [C.c_int] * 3
        L.lcqp_host_problem_destroy.argtypes = [vp]
        L.lcqp_host_problem_set_device.argtypes = [vp, C.c_int]
        L.lcqp_host_problem_set_host_loop.argtypes = [vp, C.c_int]
        L.lcqp_host_problem_last_engine.argtypes = [vp]
        L.lcqp_host_problem_set_options.argtypes = [vp, vp]
        L.lcqp_host_problem_load_dense.argtypes = [vp] + [_dp] * 15
        cp = C.POINTER(_CscArg)
        L.lcqp_host_problem_load_csc.argtypes = [vp, cp, _dp, cp, cp, _dp, _dp, _dp, _dp, cp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.lcqp_host_problem_load_files.argtypes = [vp, C.POINTER(C.c_char_p)]
        for f in ("switch_to_sparse", "switch_to_dense", "run", "number_of_primals", "number_of_duals"):
            getattr(L, "lcqp_host_problem_" + f).argtypes = [vp]
        L.lcqp_host_problem_get_primal.argtypes = [vp, _dp]
        L.lcqp_host_problem_get_dual.argtypes = [vp, _dp]
        L.lcqp_host_problem_get_stats.argtypes = [vp, C.POINTER(_Stats)]
        L.lcqp_host_problem_get_track.argtypes = [vp, C.c_int, _dp, C.c_int]
        _lib = L
    return _lib


def _rv(code):
    try:
        return ReturnValue(code)
    except ValueError:
        return code


# field ids of include/lcqp_host.h
(_STAT_TOL, _COMP_TOL, _RHO0, _BETA, _ZERO_FIRST, _PERTURB, _MAX_ITER, _MAX_RHO, _NDYN, _ETA, _PRINT, _STORE, _QPSOLVER,
 _SEED) = range(14)


class Options:
    """include/Options.hpp:30-221; setters validate like src/Options.cpp:80-259 and return the same codes."""

    def __init__(self, rhs=None):
        L = _host()
        self._h = L.lcqp_host_options_copy(rhs._h) if rhs is not None else L.lcqp_host_options_create()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lcqp_host_options_destroy(h)

    def _set(self, field, val):
        return _rv(_host().lcqp_host_options_set(self._h, field, float(val)))

    def _get(self, field):
        return _host().lcqp_host_options_get(self._h, field)

    def setToDefault(self):
        _host().lcqp_host_options_set_to_default(self._h)

    def getStationarityTolerance(self): return self._get(_STAT_TOL)
    def setStationarityTolerance(self, val): return self._set(_STAT_TOL, val)
    def getComplementarityTolerance(self): return self._get(_COMP_TOL)
    def setComplementarityTolerance(self, val): return self._set(_COMP_TOL, val)
    def getInitialPenaltyParameter(self): return self._get(_RHO0)
    def setInitialPenaltyParameter(self, val): return self._set(_RHO0, val)
    def getPenaltyUpdateFactor(self): return self._get(_BETA)
    def setPenaltyUpdateFactor(self, val): return self._set(_BETA, val)
    def getSolveZeroPenaltyFirst(self): return bool(self._get(_ZERO_FIRST))
    def setSolveZeroPenaltyFirst(self, val): return self._set(_ZERO_FIRST, bool(val))
    def getPerturbStep(self): return bool(self._get(_PERTURB))
    def setPerturbStep(self, val): return self._set(_PERTURB, bool(val))
    def getMaxIterations(self): return int(self._get(_MAX_ITER))
    def setMaxIterations(self, val): return self._set(_MAX_ITER, int(val))
    def getMaxPenaltyParameter(self): return self._get(_MAX_RHO)
    def setMaxPenaltyParameter(self, val): return self._set(_MAX_RHO, val)
    def getNDynamicPenalty(self): return int(self._get(_NDYN))
    def setNDynamicPenalty(self, val): return self._set(_NDYN, int(val))
    def getEtaDynamicPenalty(self): return self._get(_ETA)
    def setEtaDynamicPenalty(self, val): return self._set(_ETA, val)
    def getPrintLevel(self): return PrintLevel(int(self._get(_PRINT)))
    def setPrintLevel(self, val): return self._set(_PRINT, int(val))
    def getStoreSteps(self): return bool(self._get(_STORE))
    def setStoreSteps(self, val): return self._set(_STORE, bool(val))
    def getQPSolver(self): return QPSolver(int(self._get(_QPSOLVER)))
    def setQPSolver(self, val): return self._set(_QPSOLVER, int(val))

    # not in the reference: deterministic seed of perturbStep, and the subsolver knobs (lcqp_options_t tail)
    def setPerturbSeed(self, seed): return self._set(_SEED, int(seed))

    def getHIPOptions(self):
        o = _capi.Options()
        _host().lcqp_host_options_get_hip(self._h, C.byref(o))
        return o

    def setHIPOptions(self, o):
        _host().lcqp_host_options_set_hip(self._h, C.byref(o))


class OutputStatistics:
    """include/OutputStatistics.hpp:31-227: a value object filled by LCQProblem.getOutputStatistics(stats)."""
    (_INNER, _SUB, _ACCU, _ALPHA, _PNORM, _STAT, _OBJ, _PHI, _MERIT, _XSTEPS) = range(10)

    def __init__(self):
        self._s = _Stats()
        self._tracks = {}

    def getIterTotal(self): return self._s.iterTotal
    def getIterOuter(self): return self._s.iterOuter
    def getSubproblemIter(self): return self._s.subproblemIter
    def getRhoOpt(self): return self._s.rhoOpt
    def getSolutionStatus(self): return AlgorithmStatus(self._s.status)
    def getQPSolverExitFlag(self): return self._s.qpSolverExitFlag
    def getInnerIters(self): return [int(v) for v in self._tracks.get(self._INNER, [])]
    def getSubproblemIters(self): return [int(v) for v in self._tracks.get(self._SUB, [])]
    def getAccuSubproblemIters(self): return [int(v) for v in self._tracks.get(self._ACCU, [])]
    def getStepLength(self): return list(self._tracks.get(self._ALPHA, []))
    def getStepSize(self): return list(self._tracks.get(self._PNORM, []))
    def getStatVals(self): return list(self._tracks.get(self._STAT, []))
    def getObjVals(self): return list(self._tracks.get(self._OBJ, []))
    def getPhiVals(self): return list(self._tracks.get(self._PHI, []))
    def getMeritVals(self): return list(self._tracks.get(self._MERIT, []))

    def getxSteps(self):
        """rows = stored iterates (include/OutputStatistics.hpp:151; not bound by the reference's Python module)"""
        return self._tracks.get(self._XSTEPS, np.zeros((0, 0)))


class cscWrapper:
    """interfaces/python/lcqpow/LCQProblem.cpp:23-55: owns copies of a CSC triple (m, n, nnx, x, i, p)."""

    def __init__(self, m, n, nnx, x, i, p):
        self.m, self.n, self.nnx = int(m), int(n), int(nnx)
        self.x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        self.i = np.ascontiguousarray(np.asarray(i, dtype=np.int32))
        self.p = np.ascontiguousarray(np.asarray(p, dtype=np.int32))
        if self.p.size != self.n + 1 or self.x.size < self.nnx or self.i.size < self.nnx:
            raise ValueError("cscWrapper: need p[n+1], i[nnx], x[nnx]")

    def _arg(self):
        return _CscArg(self.m, self.n, self.nnx, self.x.ctypes.data_as(_dp), self.i.ctypes.data_as(_ip), self.p.ctypes.data_as(_ip))


def _vec(v):
    """Eigen::VectorXd argument -> contiguous doubles, None for absent or empty (getRawPtrFromEigenVectorXd, LCQProblem.cpp:58-61)"""
    if v is None:
        return None
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel())
    return a if a.size > 0 else None


def _mat(M, order):
    """Eigen::MatrixXd argument -> the element order .data() would have (LCQProblem.cpp:64-67 and the module docstring)"""
    if M is None:
        return None
    a = np.asarray(M, dtype=np.float64)
    if a.size == 0:
        return None
    if a.ndim == 1:
        a = a.reshape(-1, 1)       # Eigen turns a 1-d array into a column vector
    return np.ascontiguousarray(a.ravel(order="F" if order == "F" else "C"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


class LCQProblem:
    """include/LCQProblem.hpp:38-242 as bound by interfaces/python/lcqpow/LCQProblem.cpp:79-176."""

    def __init__(self, nV=None, nC=None, nComp=None, device=0):
        if nV is None:
            raise TypeError("LCQProblem(): the default-constructed object of the reference has no dimensions and cannot be "
                            "loaded (src/LCQProblem.cpp:40); pass nV, nC, nComp")
        self._h = _host().lcqp_host_problem_create(int(nV), int(nC), int(nComp))
        if not self._h:
            raise MemoryError("lcqp_host_problem_create")
        self._nV, self._nC, self._nComp = int(nV), int(nC), int(nComp)
        _host().lcqp_host_problem_set_device(self._h, int(device))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lcqp_host_problem_destroy(h)

    def setOptions(self, options):
        _host().lcqp_host_problem_set_options(self._h, options._h)

    def setHostLoop(self, host_loop=True):
        """QPSolver.HIP_DENSE runs the whole homotopy on the device; True keeps the reference's host loop over the SubsolverHIP plugin
        for it too (no reference analogue; the reference's three solver values always use the host loop)"""
        _host().lcqp_host_problem_set_host_loop(self._h, int(bool(host_loop)))

    def getLastEngine(self):
        """which engine the last runSolver used: 0 none yet, 1 host loop over the subsolver plugin, 2 whole homotopy on the device (dense
        kernels), 3 the sparse engine (no reference analogue)"""
        return int(_host().lcqp_host_problem_last_engine(self._h))

    def loadLCQP(self, Q=None, g=None, L=None, R=None, lbL=None, ubL=None, lbR=None, ubR=None, A=None, lbA=None, ubA=None,
                 lb=None, ub=None, x0=None, y0=None, *, order="F",
                 Q_file=None, g_file=None, L_file=None, R_file=None, lbL_file=None, ubL_file=None, lbR_file=None,
                 ubR_file=None, A_file=None, lbA_file=None, ubA_file=None, lb_file=None, ub_file=None, x0_file=None,
                 y0_file=None):
        """The three overloads of the reference, chosen by argument type: numpy arrays (dense), cscWrapper (sparse),
        or path strings / the *_file keywords (files)."""
        H = _host()
        if Q_file is not None or isinstance(Q, (str, bytes, os.PathLike)):
            names = [Q_file or Q, g_file or g, L_file or L, R_file or R, lbL_file or lbL, ubL_file or ubL, lbR_file or lbR,
                     ubR_file or ubR, A_file or A, lbA_file or lbA, ubA_file or ubA, lb_file or lb, ub_file or ub,
                     x0_file or x0, y0_file or y0]
            arr = (C.c_char_p * 15)(*[None if f is None else os.fsencode(f) for f in names])
            return _rv(H.lcqp_host_problem_load_files(self._h, arr))
        vecs = [_vec(v) for v in (lbL, ubL, lbR, ubR, lbA, ubA, lb, ub, x0, y0)]
        vlbL, vubL, vlbR, vubR, vlbA, vubA, vlb, vub, vx0, vy0 = vecs
        vg = _vec(g)
        # the C side reads nV, nC, nComp-sized blocks through raw pointers: a wrongly shaped array is an error here, not an
        # out-of-bounds read there
        n, nC, nK = self._nV, self._nC, self._nComp

        def sized(name, a, size):
            if a is not None and a.size != size:
                raise ValueError(f"loadLCQP: {name} has {a.size} values, expected {size}")
        for nm, a, sz in (("g", vg, n), ("lbL", vlbL, nK), ("ubL", vubL, nK), ("lbR", vlbR, nK), ("ubR", vubR, nK), ("lbA", vlbA, nC),
                          ("ubA", vubA, nC), ("lb", vlb, n), ("ub", vub, n), ("x0", vx0, n), ("y0", vy0, n + nC + 2 * nK)):
            sized(nm, a, sz)
        if isinstance(Q, cscWrapper):
            for M in (L, R):
                if not isinstance(M, cscWrapper):
                    raise TypeError("sparse loadLCQP: Q, L, R (and A) must all be cscWrapper")
            aQ, aL, aR = Q._arg(), L._arg(), R._arg()
            aA = A._arg() if isinstance(A, cscWrapper) else None
            return _rv(H.lcqp_host_problem_load_csc(self._h, C.byref(aQ), _ptr(vg), C.byref(aL), C.byref(aR), _ptr(vlbL),
                                                    _ptr(vubL), _ptr(vlbR), _ptr(vubR), C.byref(aA) if aA is not None else None,
                                                    _ptr(vlbA), _ptr(vubA), _ptr(vlb), _ptr(vub), _ptr(vx0), _ptr(vy0)))
        mQ, mL, mR, mA = (_mat(M, order) for M in (Q, L, R, A))
        for nm, a, sz in (("Q", mQ, n * n), ("L", mL, nK * n), ("R", mR, nK * n), ("A", mA, nC * n)):
            sized(nm, a, sz)
        return _rv(H.lcqp_host_problem_load_dense(self._h, _ptr(mQ), _ptr(vg), _ptr(mL), _ptr(mR), _ptr(vlbL), _ptr(vubL),
                                                  _ptr(vlbR), _ptr(vubR), _ptr(mA), _ptr(vlbA), _ptr(vubA), _ptr(vlb), _ptr(vub),
                                                  _ptr(vx0), _ptr(vy0)))

    def switchToSparseMode(self):
        return _rv(_host().lcqp_host_problem_switch_to_sparse(self._h))

    def switchToDenseMode(self):
        return _rv(_host().lcqp_host_problem_switch_to_dense(self._h))

    def runSolver(self):
        return _rv(_host().lcqp_host_problem_run(self._h))

    def getNumberOfPrimals(self):
        return _host().lcqp_host_problem_number_of_primals(self._h)

    def getNumberOfDuals(self):
        return _host().lcqp_host_problem_number_of_duals(self._h)

    def getPrimalSolution(self):
        x = np.zeros(self.getNumberOfPrimals())
        _host().lcqp_host_problem_get_primal(self._h, _ptr(x))
        return x

    def getDualSolution(self):
        y = np.zeros(self.getNumberOfDuals())
        if y.size:
            _host().lcqp_host_problem_get_dual(self._h, _ptr(y))
        return y

    def getOutputStatistics(self, stats):
        H = _host()
        H.lcqp_host_problem_get_stats(self._h, C.byref(stats._s))
        stats._tracks = {}
        n = stats._s.nSteps
        if n > 0:
            for which in range(9):
                buf = np.zeros(n)
                H.lcqp_host_problem_get_track(self._h, which, _ptr(buf), n)
                stats._tracks[which] = buf
            nx = H.lcqp_host_problem_get_track(self._h, 9, None, 0)
            xs = np.zeros(nx)
            H.lcqp_host_problem_get_track(self._h, 9, _ptr(xs), nx)
            stats._tracks[9] = xs.reshape(-1, self._nV) if nx else np.zeros((0, self._nV))
        return None
