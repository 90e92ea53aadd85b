"""Sparse arm on the GPU (lcqp_hip_sparse_*, BASELINE configs[4]) against the sparse CPU oracle through the C ABI."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def _run(hip, n, nC, nK, B, span=6, **okw):
    Qp, Ap = P.sparse_pattern(n, nC, nK, span=span)
    inst = [P.sparse_instance(i, n, nC, nK, span=span) for i in range(B)]
    sb = hip.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=hip.default_options(perturbStep=0, printLevel=0, **okw))
    assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                   lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst])) == 0
    sb.run()
    x, y, st = sb.solution()
    return sb, inst, x, y, st


@pytest.mark.parametrize("shape,B", [((64, 32, 8), 6), ((512, 256, 64), 4), ((4096, 2048, 512), 4)])
def test_sparse_hip_matches_oracle(hip, oracle, shape, B):
    """same algorithm, same inputs: solution to 1e-9 / duals 1e-7 (fp64 summation order only), same return code and status,
    iterate counts equal up to one inner cycle"""
    n, nC, nK = shape
    sb, inst, x, y, st = _run(hip, n, nC, nK, B)
    assert 1 <= sb.bandwidth() <= 63 and sb.algorithmic_bytes() > 0
    opt = oracle.default_options(perturbStep=0)
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert st[b]["returnValue"] == ro["ret"] == 0
        assert np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
        assert st[b]["status"] == ro["stats"]["status"] and abs(st[b]["iterTotal"] - ro["stats"]["iterTotal"]) <= 4
    # second run on the same handle reproduces the first bit for bit
    sb.run()
    x2, y2, _ = sb.solution()
    assert np.array_equal(x, x2) and np.array_equal(y, y2)
    sb.close()


def test_sparse_with_perturbed_steps(hip, oracle):
    """perturbStep on (the reference's default, src/LCQProblem.cpp:1353-1362): the LCQP level then takes C xk from the perturbed iterate
    (one more pass over E), so that the perturbation reaches the penalty gradient; Q xk is carried along the steps.  The oracle
    recomputes both from the perturbed iterate -- the two sides may differ in the last bits of an iterate, not in where they end."""
    n, nC, nK, B = 512, 256, 64, 4
    Qp, Ap = P.sparse_pattern(n, nC, nK)
    inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
    sb = hip.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=hip.default_options(perturbStep=1, perturbSeed=7, printLevel=0))
    assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                   lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst])) == 0
    sb.run()
    x, y, st = sb.solution()
    opt = oracle.default_options(perturbStep=1, perturbSeed=7)
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert st[b]["returnValue"] == ro["ret"] == 0
        assert np.abs(x[b] - ro["x"]).max() < 1e-7 and np.abs(y[b] - ro["y"]).max() < 1e-5
        L = d["E"].tocsr()[nC:nC + nK] @ x[b]; R = d["E"].tocsr()[nC + nK:] @ x[b]
        assert abs(L @ R) < 1e-10 and L.min() > -1e-9 and R.min() > -1e-9
    sb.close()


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_sparse_lane_group_widths(hip, oracle, monkeypatch, lanes):
    """the engine gives every instance G lanes of a wavefront, G the smallest of 8, 16, 32, 64 above the half bandwidth (here 7 -> 8);
    LCQP_SPARSE_LANES forces the wider groups (register window at 16, LDS window at 32 and 64) on the same problems; a batch
    that does not fill its last wavefront"""
    monkeypatch.setenv("LCQP_SPARSE_LANES", str(lanes))
    n, nC, nK, B = 512, 256, 64, 5
    sb, inst, x, y, st = _run(hip, n, nC, nK, B)
    assert sb.lanes() == lanes and sb.bandwidth() == 7
    opt = oracle.default_options(perturbStep=0)
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert st[b]["returnValue"] == ro["ret"] == 0
        assert np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
    sb.close()


@pytest.mark.parametrize("pool,B", [(4, 21), (8, 40)])
def test_sparse_phase_machine_pools(hip, oracle, monkeypatch, pool, B):
    """k_sparse_sched (round 4): instances move through per-phase queues, in POOLS of a power-of-two number of consecutive instances (the
    32-bit lane offsets); LCQP_SPARSE_POOL forces pools of 4 / 8 instances, so that a small batch spans several pools with a ragged last one and
    more wavefronts than instances.  Whatever lanes run an instance and in whatever order, the results are those of the instance alone:
    the same bits as a batch of one pool, and the oracle's solution."""
    n, nC, nK = 512, 256, 64
    sb0, inst, x0, y0, st0 = _run(hip, n, nC, nK, B)          # one pool
    sb0.close()
    monkeypatch.setenv("LCQP_SPARSE_POOL", str(pool))
    sb, _, x, y, st = _run(hip, n, nC, nK, B)
    assert np.array_equal(x, x0) and np.array_equal(y, y0)
    assert [s["iterTotal"] for s in st] == [s["iterTotal"] for s in st0] and all(s["returnValue"] == 0 for s in st)
    sb.run(); sb.synchronize()                               # a second run on the same handle: the queues are refilled
    x2, y2, _ = sb.solution()
    assert np.array_equal(x2, x)
    opt = oracle.default_options(perturbStep=0)
    for b in (0, B // 2, B - 1):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert ro["ret"] == 0 and np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
    sb.close()


def test_sparse_scheduler_under_contention_is_reproducible(hip, oracle):
    """A batch large enough that every pop and push of k_sparse_sched races with others (1200 instances, 2048 persistent wavefronts; streaming
    steps of up to 16 instances, band steps of 8, chained phases): three runs on one handle regroup the instances differently and must give
    the same bits; all solved; a sample equals the oracle.  (tools/micro/sparse_soak.py does the same at the benchmark's size.)"""
    n, nC, nK, B = 256, 128, 32, 1200
    sb, inst, x, y, st = _run(hip, n, nC, nK, B)
    assert all(s["returnValue"] == 0 for s in st)
    for _ in range(2):
        sb.run(); sb.synchronize()
        x2, y2, st2 = sb.solution()
        assert np.array_equal(x2, x) and np.array_equal(y2, y)
        assert [s["iterTotal"] for s in st2] == [s["iterTotal"] for s in st]
    opt = oracle.default_options(perturbStep=0)
    for b in (0, 399, 800, B - 1):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert ro["ret"] == 0 and np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
    sb.close()


@pytest.mark.parametrize("span", [6, 10])
def test_sparse_working_set_bits_in_lds_or_flags_in_memory(hip, monkeypatch, span):
    """sp_ph_factor (lane groups of 8 and 16) asks a bit set in LDS for the working set while it assembles a band row; problems with more
    rows of E than 16 KB of LDS per wavefront hold read the 4-byte flags from memory instead (LCQP_SPARSE_NOBITS=1 forces that path).  Same
    matrix either way: the same bits."""
    n, nC, nK, B = 512, 256, 64, 11
    sb0, _, x0, y0, st0 = _run(hip, n, nC, nK, B, span=span)
    assert sb0.lanes() in (8, 16, 32)
    sb0.close()
    monkeypatch.setenv("LCQP_SPARSE_NOBITS", "1")
    sb, _, x, y, st = _run(hip, n, nC, nK, B, span=span)
    assert np.array_equal(x, x0) and np.array_equal(y, y0)
    assert [s["iterTotal"] for s in st] == [s["iterTotal"] for s in st0] and all(s["returnValue"] == 0 for s in st)
    sb.close()


@pytest.mark.parametrize("span", [10, 18])
def test_sparse_wider_bands(hip, oracle, span):
    """constraint rows over 10 / 18 variables: half bandwidths beyond 7 select wider lane groups by themselves"""
    n, nC, nK, B = 512, 256, 64, 9
    sb, inst, x, y, st = _run(hip, n, nC, nK, B, span=span)
    assert sb.bandwidth() > 7 and sb.lanes() in (16, 32, 64) and sb.bandwidth() < sb.lanes() <= 2 * sb.bandwidth() + 2
    opt = oracle.default_options(perturbStep=0)
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert st[b]["returnValue"] == ro["ret"], (b, st[b], ro["stats"])
        if ro["ret"] == 0:
            assert np.abs(x[b] - ro["x"]).max() < 1e-8 and np.abs(y[b] - ro["y"]).max() < 1e-6
    sb.close()


def test_sparse_hip_batch_properties(hip):
    """a batch at BASELINE size: every instance solved, complementarity exact, stationarity of the returned duals"""
    n, nC, nK, B = 4096, 2048, 512, 64
    sb, inst, x, y, st = _run(hip, n, nC, nK, B)
    assert all(s["returnValue"] == 0 for s in st)
    for b in range(0, B, 7):
        d = inst[b]
        Qc, Ec = d["Q"].tocsr(), d["E"].tocsr()
        Lx, Rx = x[b][8 * np.arange(nK)], x[b][8 * np.arange(nK) + 4]
        assert (Lx * Rx).sum() < 2.2e-13
        assert np.abs(Qc @ x[b] + d["g"] - Ec.T @ y[b]).max() < 1e-8
    sb.close()


def test_sparse_default_options_and_admm_first(hip, oracle):
    """reference defaults (perturbStep on, seeded) and an ADMM-first cold start (the OSQP route: ADMM iterations, then polish)"""
    n, nC, nK, B = 512, 256, 64, 3
    for kw in (dict(perturbStep=1, perturbSeed=7), dict(admmFirst=50)):
        base = dict(perturbStep=0); base.update(kw)
        Qp, Ap = P.sparse_pattern(n, nC, nK)
        inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
        sb = hip.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=hip.default_options(printLevel=0, **base))
        assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                       lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst])) == 0
        sb.run()
        x, y, st = sb.solution()
        for b in range(B):
            d = inst[b]
            ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=oracle.default_options(**base))
            assert st[b]["returnValue"] == ro["ret"] == 0, (kw, st[b], ro["stats"])
            assert np.abs(x[b] - ro["x"]).max() < 1e-8
            if "admmFirst" in kw:
                assert st[b]["admmIter"] >= 50 and ro["stats"]["admmIter"] >= 50
        sb.close()


@pytest.mark.parametrize("case", ["shifted complementarity bounds", "x0 and y0 given"])
def test_sparse_bounds_and_initial_guess(hip, oracle, case):
    """lbL, lbR > 0 (phi_const and g_phi of src/LCQProblem.cpp:969-996), finite ubL; and a warm start (x0, y0 with the sign of
    src/SubsolverOSQP.cpp:196-199): the sparse loadLCQP overload's optional arguments (src/LCQProblem.cpp:390-441), HIP vs oracle"""
    n, nC, nK, B = 512, 256, 64, 5
    Qp, Ap = P.sparse_pattern(n, nC, nK)
    inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
    kw = []
    for b in range(B):
        rng = np.random.default_rng(b)
        if case.startswith("shifted"):
            kw.append(dict(lbL=np.full(nK, 0.01), lbR=np.full(nK, 0.02), ubL=np.full(nK, 5.0), ubR=np.full(nK, np.inf)))
        else:
            kw.append(dict(x0=rng.uniform(-0.5, 0.5, n), y0=rng.uniform(-0.1, 0.1, nC + 2 * nK)))
    sb = hip.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=hip.default_options(perturbStep=0, printLevel=0))
    stacked = {k: np.stack([q[k] for q in kw]) for k in kw[0]}
    assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                   lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]), **stacked) == 0
    sb.run()
    x, y, st = sb.solution()
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"],
                                      opt=oracle.default_options(perturbStep=0), **kw[b])
        assert st[b]["returnValue"] == ro["ret"] == 0, (case, b, st[b], ro["stats"])
        assert np.abs(x[b] - ro["x"]).max() < 1e-8 and np.abs(y[b] - ro["y"]).max() < 1e-6
        if case.startswith("shifted"):
            Lx, Rx = x[b][8 * np.arange(nK)], x[b][8 * np.arange(nK) + 4]
            assert ((Lx - 0.01) * (Rx - 0.02)).sum() < 2.2e-13 and Lx.min() > 0.01 - 1e-9 and Rx.min() > 0.02 - 1e-9
    sb.close()


def test_sparse_profile_entry_point_needs_a_profile_build(hip):
    """lcqp_hip_sparse_read_profile reports LCQP_HIP_UNSUPPORTED (901) on the product build: the phase stamps exist only in
    -DLCQP_PROFILE builds (tools/gpu.py sparse_profile)"""
    import ctypes as C
    sb, inst, x, y, st = _run(hip, 64, 32, 8, 2)
    out = np.zeros(8)
    hip.lib().lcqp_hip_sparse_read_profile.argtypes = [C.c_void_p, C.c_void_p]
    assert hip.lib().lcqp_hip_sparse_read_profile(sb.h, out.ctypes.data_as(C.c_void_p)) == 901
    sb.close()


def test_sparse_bordered_band_circle(hip, oracle):
    """examples/OptimizeOnCircle.cpp:44 (the reference's own OSQP_SPARSE example) has an arrow-shaped KKT matrix: the coupling row and the
    two shared variables touch a hundred nodes each.  The engine orders it as a narrow band plus three border nodes (bordered band LDL':
    band factor, W = U inv(B), Schur complement of the border) and matches the sparse oracle run with the same ordering, the dense oracle
    and the optimum the reference prints (examples/OptimizeOnCircle.cpp:144)."""
    import scipy.sparse as sp
    d = P.circle(100)
    n, nC, nK = d["nV"], d["nC"], d["nComp"]
    Q = sp.csc_matrix(d["Q"]); E = sp.csc_matrix(np.vstack([d["A"], d["L"], d["R"]]))
    Q.sort_indices(); E.sort_indices()
    B = 3
    sb = hip.SparseBatchLCQP(B, n, nC, nK, Q, E, opt=hip.default_options(perturbStep=0, printLevel=0))
    assert sb.border() == 3 and 1 <= sb.bandwidth() <= 63
    tile = lambda v: np.tile(np.asarray(v, dtype=float), (B, 1))
    assert sb.load(0, B, tile(Q.data), tile(d["g"]), tile(E.data), lbA=tile(d["lbA"]), ubA=tile(d["ubA"]), x0=tile(d["x0"])) == 0
    perm = sb.ordering()                                                   # (after load: the Hessians select between two orderings of the band)
    assert sorted(perm[-3:].tolist()) == [0, 1, n + nC - 1]                # x_0, x_1 and the row sum(theta) = 1
    sb.run()
    x, y, st = sb.solution()
    ro = oracle.sparse_lcqp_solve(n, nC, nK, Q.tocsr(), d["g"], E.tocsr(), lbA=d["lbA"], ubA=d["ubA"], x0=d["x0"], perm=perm, w=sb.bandwidth(), kb=sb.border(),
                                  opt=oracle.default_options(perturbStep=0))
    rd = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0))
    for b in range(B):
        assert st[b]["returnValue"] == ro["ret"] == 0 and st[b]["status"] == ro["stats"]["status"]
        assert np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
        assert abs(st[b]["iterTotal"] - ro["stats"]["iterTotal"]) <= 4
        assert np.abs(x[b] - rd["x"]).max() < 1e-7 and np.abs(x[b][:2] - [0.1811, -0.9835]).max() < 1e-4
    sb.close()


def _solve_general(hip, oracle, d, B=1, leaf=48):
    """one problem given as scipy matrices (Q, E = [A; L; R]) on the sparse engine and on the sparse oracle's general LDL' (w = -1) with an
    ordering computed in Python (tests/oracle_py.py::kkt_ordering_general): independent of the product's own analysis"""
    n, nC, nK = d["nV"], d["nC"], d["nComp"]
    Qc, Ec = d["Q"].tocsc(), d["E"].tocsc()
    Qc.sort_indices(); Ec.sort_indices()
    sb = hip.SparseBatchLCQP(B, n, nC, nK, Qc, Ec, opt=hip.default_options(perturbStep=0, printLevel=0))
    for b in range(B):
        assert sb.load(b, 1, Qc.data[None, :], d["g"][None, :], Ec.data[None, :], lbA=d["lbA"][None, :], ubA=d["ubA"][None, :]) == 0
    sb.run()
    x, y, st = sb.solution()
    Qr, Er = d["Q"].tocsr(), d["E"].tocsr()
    perm = oracle.kkt_ordering_general(n, Qr.indptr, Qr.indices, Er.indptr, Er.indices, leaf=leaf)
    ro = oracle.sparse_lcqp_solve(n, nC, nK, Qr, d["g"], Er, lbA=d["lbA"], ubA=d["ubA"], perm=perm, w=-1, kb=0, opt=oracle.default_options(perturbStep=0))
    return sb, x, y, st, ro


def test_sparse_pattern_with_dense_rows_runs_on_the_general_ldl(hip, oracle):
    """a pattern with more dense nodes than the border of the band engine takes (forty dense rows over two hundred variables: refused until
    round 6) is one dense front for the general sparse LDL' (lcqp_sparse_general.hpp): the engine takes it and matches the oracle"""
    import scipy.sparse as sp
    n, nC, nK = 200, 40, 8
    rng = np.random.default_rng(0)
    A = rng.standard_normal((nC, n)) / np.sqrt(n)
    L = np.zeros((nK, n)); R = np.zeros((nK, n))
    L[np.arange(nK), np.arange(nK)] = 1; R[np.arange(nK), nK + np.arange(nK)] = 1
    xs = rng.uniform(0.2, 1.0, n); xs[nK:2 * nK] = 0.0
    d = dict(nV=n, nC=nC, nComp=nK, Q=sp.csc_matrix(np.diag(rng.uniform(1, 2, n))), E=sp.csc_matrix(np.vstack([A, L, R])), g=rng.uniform(-1, 1, n),
             lbA=A @ xs - rng.uniform(0.1, 1, nC), ubA=A @ xs + rng.uniform(0.1, 1, nC))
    sb, x, y, st, ro = _solve_general(hip, oracle, d)
    assert sb.fronts() >= 1 and sb.lanes() == 64 and sb.border() == 0
    assert st[0]["returnValue"] == ro["ret"] == 0
    assert np.abs(x[0] - ro["x"]).max() < 1e-9 and np.abs(y[0] - ro["y"]).max() < 1e-7
    sb.close()


def test_sparse_pattern_too_dense_for_the_sparse_engine_is_refused(hip):
    """what the general LDL' cannot hold in a wavefront's LDS panel -- a front of more than 576 rows, here sixty dense rows over seven hundred
    variables, one clique -- is refused with a message (the host layer runs such a problem on the dense kernels behind the OSQP_SPARSE surface)"""
    import scipy.sparse as sp
    n, nC, nK = 700, 60, 8
    rng = np.random.default_rng(0)
    A = rng.standard_normal((nC, n))
    L = np.zeros((nK, n)); R = np.zeros((nK, n))
    L[np.arange(nK), np.arange(nK)] = 1; R[np.arange(nK), nK + np.arange(nK)] = 1
    Q = sp.csc_matrix(np.eye(n)); E = sp.csc_matrix(np.vstack([A, L, R]))
    with pytest.raises(RuntimeError, match="too dense for the sparse engine"):
        hip.SparseBatchLCQP(1, n, nC, nK, Q, E)


@pytest.mark.parametrize("g,nK,nC", [(44, 300, 200), (64, 300, 200)])
def test_sparse_grid_pattern_on_the_general_ldl(hip, oracle, g, nK, nC):
    """A KKT graph that is a 2-D grid (5-point stencil Hessian, complementarity and constraint rows between neighbouring cells): half bandwidth
    ~ 2 g after reverse Cuthill-McKee, no small border -- the pattern that is "neither banded nor bordered".  The sparse engine runs it on the
    general LDL' (nested dissection, dense fronts, one wavefront per instance): same solution as the oracle's general LDL' (up-looking, another
    ordering) to 1e-9 / 1e-7, same return code and status, iterate counts equal up to one inner cycle; a batch of three copies gives three
    times the same bits, and a second run reproduces the first."""
    d = P.grid_lcqp(g, nK, nC)
    sb, x, y, st, ro = _solve_general(hip, oracle, d, B=3)
    assert sb.fronts() > 4 and sb.lanes() == 64
    for b in range(3):
        assert st[b]["returnValue"] == ro["ret"] == 0
        assert np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
        assert st[b]["status"] == ro["stats"]["status"] and abs(st[b]["iterTotal"] - ro["stats"]["iterTotal"]) <= 4
        assert np.array_equal(x[b], x[0]) and np.array_equal(y[b], y[0])
    sb.run()
    x2, y2, _ = sb.solution()
    assert np.array_equal(x, x2) and np.array_equal(y, y2)
    sb.close()


def test_sparse_general_ldl_on_a_banded_pattern_matches_the_band_engine(hip, oracle, monkeypatch):
    """LCQP_SPARSE_GENERAL=1 (test hook) sends a pattern the band engine takes through the general LDL': the banded synthetic workload, four
    instances with different data -- each matches the oracle (band LDL', its own ordering) like the band engine does"""
    monkeypatch.setenv("LCQP_SPARSE_GENERAL", "1")
    n, nC, nK, B = 512, 256, 64, 4
    sb, inst, x, y, st = _run(hip, n, nC, nK, B)
    assert sb.fronts() > 4 and sb.lanes() == 64
    opt = oracle.default_options(perturbStep=0)
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=opt)
        assert st[b]["returnValue"] == ro["ret"] == 0
        assert np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
    sb.close()


@pytest.mark.parametrize("feature", ["perturbed steps", "admm first", "shifted complementarity bounds", "x0 and y0 given"])
def test_sparse_general_ldl_with_every_option_of_the_arm(hip, oracle, monkeypatch, feature):
    """The general LDL' sits behind the same two entry points as the band engines (factorise for a working set, solve), so everything else of
    the arm -- perturbStep (src/LCQProblem.cpp:1353-1362), an ADMM-first cold start (ADMM iterations on the factor of k_sparse_setup), shifted
    complementarity bounds with finite ubL, a warm start (x0, y0) -- must work through it unchanged: the banded workload sent through the general
    engine (LCQP_SPARSE_GENERAL=1) against the oracle"""
    monkeypatch.setenv("LCQP_SPARSE_GENERAL", "1")
    n, nC, nK, B = 512, 256, 64, 3
    Qp, Ap = P.sparse_pattern(n, nC, nK)
    inst = [P.sparse_instance(i, n, nC, nK) for i in range(B)]
    okw, lkw = dict(perturbStep=0), [dict() for _ in range(B)]
    if feature == "perturbed steps":
        okw = dict(perturbStep=1, perturbSeed=7)
    elif feature == "admm first":
        okw = dict(perturbStep=0, admmFirst=50)
    elif feature.startswith("shifted"):
        lkw = [dict(lbL=np.full(nK, 0.01), lbR=np.full(nK, 0.02), ubL=np.full(nK, 5.0), ubR=np.full(nK, np.inf)) for _ in range(B)]
    else:
        lkw = [dict(x0=np.random.default_rng(b).uniform(-0.5, 0.5, n), y0=np.random.default_rng(100 + b).uniform(-0.1, 0.1, nC + 2 * nK)) for b in range(B)]
    sb = hip.SparseBatchLCQP(B, n, nC, nK, Qp, Ap, opt=hip.default_options(printLevel=0, **okw))
    assert sb.fronts() > 4
    stacked = {k: np.stack([q[k] for q in lkw]) for k in lkw[0]}
    assert sb.load(0, B, np.stack([d["Q"].data for d in inst]), np.stack([d["g"] for d in inst]), np.stack([d["E"].data for d in inst]),
                   lbA=np.stack([d["lbA"] for d in inst]), ubA=np.stack([d["ubA"] for d in inst]), **stacked) == 0
    sb.run()
    x, y, st = sb.solution()
    tol = 1e-7 if feature == "perturbed steps" else 1e-8
    for b in range(B):
        d = inst[b]
        ro = oracle.sparse_lcqp_solve(n, nC, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], opt=oracle.default_options(**okw), **lkw[b])
        assert st[b]["returnValue"] == ro["ret"] == 0, (feature, b, st[b], ro["stats"])
        assert np.abs(x[b] - ro["x"]).max() < tol, (feature, b)
        if feature == "admm first":
            assert st[b]["admmIter"] >= 50
    sb.close()


def test_sparse_general_ldl_on_the_circle_example(hip, oracle, monkeypatch):
    """examples/OptimizeOnCircle.cpp (the reference's own OSQP_SPARSE example; a PSD Hessian with a 5e-12 diagonal, an arrow-shaped KKT matrix) through
    the general LDL' instead of the bordered band: the coupling row and the shared variables end up in the top separators; the optimum the
    reference prints (:144), the dense oracle's solution"""
    import scipy.sparse as sp
    monkeypatch.setenv("LCQP_SPARSE_GENERAL", "1")
    d = P.circle(100)
    n, nC, nK = d["nV"], d["nC"], d["nComp"]
    Q = sp.csc_matrix(d["Q"]); E = sp.csc_matrix(np.vstack([d["A"], d["L"], d["R"]]))
    Q.sort_indices(); E.sort_indices()
    sb = hip.SparseBatchLCQP(1, n, nC, nK, Q, E, opt=hip.default_options(perturbStep=0, printLevel=0))
    assert sb.fronts() >= 1 and sb.border() == 0
    one = lambda v: np.asarray(v, dtype=float)[None, :]
    assert sb.load(0, 1, one(Q.data), one(d["g"]), one(E.data), lbA=one(d["lbA"]), ubA=one(d["ubA"]), x0=one(d["x0"])) == 0
    sb.run()
    x, y, st = sb.solution()
    rd = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0))
    assert st[0]["returnValue"] == rd["ret"] == 0
    assert np.abs(x[0] - rd["x"]).max() < 1e-7 and np.abs(x[0][:2] - [0.1811, -0.9835]).max() < 1e-4
    sb.close()


@pytest.mark.parametrize("n,nC,nK,extra", [(512, 256, 64, 1), (512, 256, 64, 3)])
def test_sparse_banded_pattern_with_coupling_rows(hip, oracle, n, nC, nK, extra):
    """the banded synthetic pattern plus `extra` coupling rows that touch every variable (a budget constraint over an OCP horizon): band of
    half bandwidth 7 + border; the device matches the sparse oracle run with the device's ordering"""
    import scipy.sparse as sp
    B = 4
    rng = np.random.default_rng(5)
    insts, pats = [], None
    for b in range(B):
        d = P.sparse_instance(b, n, nC, nK)
        E = d["E"].tocsr()
        rows = rng.uniform(0.5, 1.5, (extra, n)) / n
        A2 = sp.vstack([E[:nC], sp.csr_matrix(rows), E[nC:]], format="csc")
        A2.sort_indices()
        xs = np.linalg.lstsq(E[:nC].toarray(), 0.5 * (d["lbA"] + d["ubA"]), rcond=None)[0]
        mid = rows @ xs
        insts.append(dict(Q=d["Q"], E=A2, g=d["g"], lbA=np.concatenate([d["lbA"], mid - 5.0]), ubA=np.concatenate([d["ubA"], mid + 0.05 * (b + 1)])))
    Qp, Ep = insts[0]["Q"], insts[0]["E"]
    sb = hip.SparseBatchLCQP(B, n, nC + extra, nK, Qp, Ep, opt=hip.default_options(perturbStep=0, printLevel=0))
    assert sb.border() == extra and sb.bandwidth() <= 15
    assert sb.load(0, B, np.stack([d["Q"].data for d in insts]), np.stack([d["g"] for d in insts]), np.stack([d["E"].data for d in insts]),
                   lbA=np.stack([d["lbA"] for d in insts]), ubA=np.stack([d["ubA"] for d in insts])) == 0
    sb.run()
    x, y, st = sb.solution()
    perm = sb.ordering()
    for b in range(B):
        d = insts[b]
        ro = oracle.sparse_lcqp_solve(n, nC + extra, nK, d["Q"].tocsr(), d["g"], d["E"].tocsr(), lbA=d["lbA"], ubA=d["ubA"], perm=perm, w=sb.bandwidth(), kb=extra,
                                      opt=oracle.default_options(perturbStep=0))
        assert st[b]["returnValue"] == ro["ret"] == 0
        assert np.abs(x[b] - ro["x"]).max() < 1e-9 and np.abs(y[b] - ro["y"]).max() < 1e-7
        assert abs(st[b]["iterTotal"] - ro["stats"]["iterTotal"]) <= 4
    sb.close()
