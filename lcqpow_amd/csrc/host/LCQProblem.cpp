#include "LCQProblem.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace LCQPow {

LCQProblem::LCQProblem()
    : nV(0), nC(0), nComp(0), nDuals(0), boxDualOffset(0), device(0), loaded(false), haveBox(false), sparseSolver(false), hostLoop(false),
      Q_sparse(0), A_sparse(0), L_sparse(0), R_sparse(0), C_sparse(0) {}

LCQProblem::LCQProblem(int _nV, int _nC, int _nComp)
    : nV(0), nC(0), nComp(0), nDuals(0), boxDualOffset(0), device(0), loaded(false), haveBox(false), sparseSolver(false), hostLoop(false),
      Q_sparse(0), A_sparse(0), L_sparse(0), R_sparse(0), C_sparse(0)
{
    // consistency checks of the reference constructor (src/LCQProblem.cpp:43-67)
    if (_nV <= 0 || _nComp <= 0 || _nC < 0) return;
    nV = _nV; nC = _nC; nComp = _nComp;
}

ReturnValue LCQProblem::loadLCQP(const double* const _Q, const double* const _g, const double* const _L, const double* const _R,
                                 const double* const _lbL, const double* const _ubL, const double* const _lbR,
                                 const double* const _ubR, const double* const _A, const double* const _lbA,
                                 const double* const _ubA, const double* const _lb, const double* const _ub,
                                 const double* const _x0, const double* const _y0)
{
    if (nV <= 0 || nComp <= 0) return LCQPOBJECT_NOT_SETUP;
    if (!_Q) return INVALID_ARGUMENT;
    if (!_g) return INVALID_OBJECTIVE_LINEAR_TERM;
    if (!_A && nC > 0) return INVALID_CONSTRAINT_MATRIX;
    if (!_L || !_R) return INVALID_COMPLEMENTARITY_MATRIX;
    const int m = nC + 2 * nComp;
    const double inf = INFINITY;
    clearSparse();
    sparseSolver = false;
    Q.assign(_Q, _Q + (size_t)nV * nV);
    g.assign(_g, _g + nV);
    L.assign(_L, _L + (size_t)nComp * nV);
    R.assign(_R, _R + (size_t)nComp * nV);
    // stacked constraint matrix [A; L; R] and its bounds (setConstraints / setComplementarityBounds)
    A.assign((size_t)m * nV, 0.0);
    if (nC > 0) std::copy(_A, _A + (size_t)nC * nV, A.begin());
    std::copy(L.begin(), L.end(), A.begin() + (size_t)nC * nV);
    std::copy(R.begin(), R.end(), A.begin() + (size_t)(nC + nComp) * nV);
    C.assign((size_t)nV * nV, 0.0);
    Utilities::MatrixSymmetrizationProduct(L.data(), R.data(), C.data(), nComp, nV);
    return loadVectors(_g, _lbL, _ubL, _lbR, _ubR, _lbA, _ubA, _lb, _ub, _x0, _y0);
}

// the vectors of a problem: bounds of the stacked rows (setConstraints / setComplementarityBounds, src/LCQProblem.cpp:585-608, 726-785), box
// bounds, initial guess -- shared by the dense and the CSC loader
ReturnValue LCQProblem::loadVectors(const double* _g, const double* _lbL, const double* _ubL, const double* _lbR, const double* _ubR,
                                    const double* _lbA, const double* _ubA, const double* _lb, const double* _ub, const double* _x0, const double* _y0)
{
    const int m = nC + 2 * nComp;
    const double inf = INFINITY;
    g.assign(_g, _g + nV);
    lbA.assign(m, -inf); ubA.assign(m, inf);
    for (int i = 0; i < nC; ++i) { if (_lbA) lbA[i] = _lbA[i]; if (_ubA) ubA[i] = _ubA[i]; }
    haveLbL = (_lbL != 0); haveLbR = (_lbR != 0);
    lbL.assign(nComp, 0.0); lbR.assign(nComp, 0.0);
    for (int i = 0; i < nComp; ++i) {
        if (_lbL) { if (_lbL[i] <= -inf) return INVALID_LOWER_COMPLEMENTARITY_BOUND; lbL[i] = _lbL[i]; }
        if (_lbR) { if (_lbR[i] <= -inf) return INVALID_LOWER_COMPLEMENTARITY_BOUND; lbR[i] = _lbR[i]; }
        lbA[nC + i] = lbL[i];
        ubA[nC + i] = _ubL ? _ubL[i] : inf;
        lbA[nC + nComp + i] = lbR[i];
        ubA[nC + nComp + i] = _ubR ? _ubR[i] : inf;
    }
    lb.assign(nV, -inf); ub.assign(nV, inf);
    for (int i = 0; i < nV; ++i) { if (_lb) lb[i] = _lb[i]; if (_ub) ub[i] = _ub[i]; }
    haveBox = (_lb != 0) || (_ub != 0);          // lb_tmp / ub_tmp of the reference (src/LCQProblem.cpp:113-121)
    xk.assign(nV, 0.0);
    if (_x0) std::copy(_x0, _x0 + nV, xk.begin());
    // the dual guess is kept in the layout it is passed in (nV + nC + 2 nComp, include/LCQProblem.ipp:144-155);
    // initializeSolver lays yk out for the chosen subsolver arm (src/LCQProblem.cpp:888-960)
    nDuals = nV + m; boxDualOffset = nV;
    y0Full.assign(nV + m, 0.0);
    haveYk = (_y0 != 0);
    if (_y0) std::copy(_y0, _y0 + nV + m, y0Full.begin());
    yk = y0Full;
    loaded = true;
    return SUCCESSFUL_RETURN;
}

static ReturnValue readOpt(std::vector<double>& v, int n, const char* file, bool& given)
{
    given = (file != 0);
    if (!given) return SUCCESSFUL_RETURN;
    v.assign(n, 0.0);
    return Utilities::readFromFile(v.data(), n, file);
}

ReturnValue LCQProblem::loadLCQP(const char* const Q_file, const char* const g_file, const char* const L_file,
                                 const char* const R_file, const char* const lbL_file, const char* const ubL_file,
                                 const char* const lbR_file, const char* const ubR_file, const char* const A_file,
                                 const char* const lbA_file, const char* const ubA_file, const char* const lb_file,
                                 const char* const ub_file, const char* const x0_file, const char* const y0_file)
{
    if (nV <= 0 || nComp <= 0) return LCQPOBJECT_NOT_SETUP;
    std::vector<double> q, gg, l, r, a, vlbL, vubL, vlbR, vubR, vlbA, vubA, vlb, vub, vx0, vy0;
    bool b[11];
    bool dummy;
    ReturnValue rc;
    if ((rc = readOpt(q, nV * nV, Q_file, dummy)) != SUCCESSFUL_RETURN || !dummy) return UNABLE_TO_READ_FILE;
    if ((rc = readOpt(gg, nV, g_file, dummy)) != SUCCESSFUL_RETURN || !dummy) return UNABLE_TO_READ_FILE;
    if ((rc = readOpt(l, nComp * nV, L_file, dummy)) != SUCCESSFUL_RETURN || !dummy) return UNABLE_TO_READ_FILE;
    if ((rc = readOpt(r, nComp * nV, R_file, dummy)) != SUCCESSFUL_RETURN || !dummy) return UNABLE_TO_READ_FILE;
    if ((rc = readOpt(vlbL, nComp, lbL_file, b[0])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vubL, nComp, ubL_file, b[1])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vlbR, nComp, lbR_file, b[2])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vubR, nComp, ubR_file, b[3])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(a, nC * nV, A_file, b[4])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vlbA, nC, lbA_file, b[5])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vubA, nC, ubA_file, b[6])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vlb, nV, lb_file, b[7])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vub, nV, ub_file, b[8])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vx0, nV, x0_file, b[9])) != SUCCESSFUL_RETURN) return rc;
    if ((rc = readOpt(vy0, nV + nC + 2 * nComp, y0_file, b[10])) != SUCCESSFUL_RETURN) return rc;
    auto p = [](std::vector<double>& v, bool given) -> const double* { return given ? v.data() : 0; };
    return loadLCQP(q.data(), gg.data(), l.data(), r.data(), p(vlbL, b[0]), p(vubL, b[1]), p(vlbR, b[2]), p(vubR, b[3]),
                    p(a, b[4]), p(vlbA, b[5]), p(vubA, b[6]), p(vlb, b[7]), p(vub, b[8]), p(vx0, b[9]), p(vy0, b[10]));
}

LCQProblem::~LCQProblem() { clearSparse(); }

void LCQProblem::clearSparse()
{
    Utilities::ClearSparseMat(&Q_sparse); Utilities::ClearSparseMat(&A_sparse); Utilities::ClearSparseMat(&L_sparse);
    Utilities::ClearSparseMat(&R_sparse); Utilities::ClearSparseMat(&C_sparse);
}

ReturnValue LCQProblem::loadLCQP(const csc* const _Q, const double* const _g, const csc* const _L, const csc* const _R,
                                 const double* const _lbL, const double* const _ubL, const double* const _lbR,
                                 const double* const _ubR, const csc* const _A, const double* const _lbA,
                                 const double* const _ubA, const double* const _lb, const double* const _ub,
                                 const double* const _x0, const double* const _y0)
{
    // going through the dense loader keeps one implementation of the bound handling (setConstraints /
    // setComplementarityBounds / setInitialGuess); the data are then held in CSC form as the reference does
    if (nV <= 0 || nComp <= 0) return LCQPOBJECT_NOT_SETUP;
    if (!_Q) return INVALID_ARGUMENT;
    if (!_L || !_R) return INVALID_COMPLEMENTARITY_MATRIX;
    if (!_A && nC > 0) return INVALID_CONSTRAINT_MATRIX;
    if (!_g) return INVALID_OBJECTIVE_LINEAR_TERM;
    if (_Q->m != nV || _Q->n != nV || _L->m != nComp || _L->n != nV || _R->m != nComp || _R->n != nV || (_A && nC > 0 && (_A->m != nC || _A->n != nV))) return INDEX_OUT_OF_BOUNDS;
    // Round 6: the data stay in compressed columns from the first line on, as in the reference (src/LCQProblem.cpp:390-441, 629-723).  Until then this
    // loader went through the dense one -- dense copies of Q, L, R, A and the dense C = L'R + R'L, nComp nV^2 operations: twenty minutes and
    // 2 GB for the 128 x 128 grid of tests/test_python_api.py before the sparse engine saw the problem.
    clearSparse();
    std::vector<double>().swap(Q); std::vector<double>().swap(A); std::vector<double>().swap(L); std::vector<double>().swap(R); std::vector<double>().swap(C);
    sparseSolver = false; loaded = false;
    const ReturnValue rv = loadVectors(_g, _lbL, _ubL, _lbR, _ubR, _lbA, _ubA, _lb, _ub, _x0, _y0);
    if (rv != SUCCESSFUL_RETURN) return rv;
    loaded = false;
    Q_sparse = Utilities::copyCSC(_Q); L_sparse = Utilities::copyCSC(_L); R_sparse = Utilities::copyCSC(_R);
    {   // A_sparse = [A; L; R] column by column (setConstraints, :629-723)
        const int m = nC + 2 * nComp;
        const bool hasA = (_A && nC > 0);
        const int nnz = (hasA ? _A->p[nV] : 0) + _L->p[nV] + _R->p[nV];
        std::vector<int> ai, ap(nV + 1, 0);
        std::vector<double> ax;
        ai.reserve(nnz); ax.reserve(nnz);
        for (int j = 0; j < nV; ++j) {
            if (hasA) for (int k = _A->p[j]; k < _A->p[j + 1]; ++k) { ai.push_back(_A->i[k]); ax.push_back(_A->x[k]); }
            for (int k = _L->p[j]; k < _L->p[j + 1]; ++k) { ai.push_back(nC + _L->i[k]); ax.push_back(_L->x[k]); }
            for (int k = _R->p[j]; k < _R->p[j + 1]; ++k) { ai.push_back(nC + nComp + _R->i[k]); ax.push_back(_R->x[k]); }
            ap[j + 1] = (int)ai.size();
        }
        A_sparse = Utilities::copyCSC(m, nV, (int)ai.size(), ax.data(), ai.data(), ap.data());
    }
    C_sparse = (L_sparse && R_sparse) ? Utilities::MatrixSymmetrizationProduct(L_sparse, R_sparse) : 0;
    if (!C_sparse) { int* p0 = new int[nV + 1](); C_sparse = Utilities::createCSC(nV, nV, 0, new double[1](), new int[1](), p0); }      // L'R + R'L = 0: an empty matrix, not a failure
    if (!Q_sparse || !A_sparse || !L_sparse || !R_sparse || !C_sparse) { clearSparse(); return FAILED_SWITCH_TO_SPARSE; }
    sparseSolver = true;
    loaded = true;
    return SUCCESSFUL_RETURN;
}

ReturnValue LCQProblem::switchToSparseMode()
{
    if (!loaded) return LCQPOBJECT_NOT_SETUP;
    if (sparseSolver) return SUCCESSFUL_RETURN;
    const int m = nC + 2 * nComp;
    Q_sparse = Utilities::dns_to_csc(Q.data(), nV, nV);
    A_sparse = Utilities::dns_to_csc(A.data(), m, nV);
    L_sparse = Utilities::dns_to_csc(L.data(), nComp, nV);
    R_sparse = Utilities::dns_to_csc(R.data(), nComp, nV);
    C_sparse = Utilities::dns_to_csc(C.data(), nV, nV);
    if (!Q_sparse || !A_sparse || !L_sparse || !R_sparse || !C_sparse) { clearSparse(); return FAILED_SWITCH_TO_SPARSE; }
    std::vector<double>().swap(Q); std::vector<double>().swap(A); std::vector<double>().swap(L);
    std::vector<double>().swap(R); std::vector<double>().swap(C);
    sparseSolver = true;
    return SUCCESSFUL_RETURN;
}

ReturnValue LCQProblem::switchToDenseMode()
{
    if (!loaded) return LCQPOBJECT_NOT_SETUP;
    if (!sparseSolver) return SUCCESSFUL_RETURN;
    double *q = Utilities::csc_to_dns(Q_sparse), *a = Utilities::csc_to_dns(A_sparse), *l = Utilities::csc_to_dns(L_sparse);
    double *r = Utilities::csc_to_dns(R_sparse), *c = Utilities::csc_to_dns(C_sparse);
    const bool ok = q && a && l && r && c;
    if (ok) {
        const int m = nC + 2 * nComp;
        Q.assign(q, q + (size_t)nV * nV); A.assign(a, a + (size_t)m * nV); L.assign(l, l + (size_t)nComp * nV);
        R.assign(r, r + (size_t)nComp * nV); C.assign(c, c + (size_t)nV * nV);
        clearSparse();
        sparseSolver = false;
    }
    delete[] q; delete[] a; delete[] l; delete[] r; delete[] c;
    return ok ? SUCCESSFUL_RETURN : FAILED_SWITCH_TO_DENSE;
}

void LCQProblem::mulQ(const double* v, double* out) const
{
    if (sparseSolver) { std::vector<double> z(nV, 0.0); Utilities::AffineLinearTransformation(1.0, Q_sparse, v, z.data(), out, nV); }
    else Utilities::MatrixMultiplication(Q.data(), v, out, nV, nV, 1);
}
void LCQProblem::mulC(const double* v, double* out) const
{
    if (sparseSolver) { std::vector<double> z(nV, 0.0); Utilities::AffineLinearTransformation(1.0, C_sparse, v, z.data(), out, nV); }
    else Utilities::MatrixMultiplication(C.data(), v, out, nV, nV, 1);
}
void LCQProblem::mulAT(const double* y, double* out) const
{
    if (sparseSolver) Utilities::TransponsedMatrixMultiplication(A_sparse, y, out);
    else Utilities::TransponsedMatrixMultiplication(A.data(), y, out, nC + 2 * nComp, nV, 1);
}
void LCQProblem::mulL(const double* v, double* out) const
{
    if (sparseSolver) Utilities::MatrixMultiplication(L_sparse, v, out);
    else Utilities::MatrixMultiplication(L.data(), v, out, nComp, nV, 1);
}
void LCQProblem::mulR(const double* v, double* out) const
{
    if (sparseSolver) Utilities::MatrixMultiplication(R_sparse, v, out);
    else Utilities::MatrixMultiplication(R.data(), v, out, nComp, nV, 1);
}
void LCQProblem::addLT(const double* y, double* out) const
{
    if (sparseSolver) Utilities::AddTransponsedMatrixMultiplication(L_sparse, y, out);
    else Utilities::AddTransponsedMatrixMultiplication(L.data(), y, out, nComp, nV, 1);
}
void LCQProblem::addRT(const double* y, double* out) const
{
    if (sparseSolver) Utilities::AddTransponsedMatrixMultiplication(R_sparse, y, out);
    else Utilities::AddTransponsedMatrixMultiplication(R.data(), y, out, nComp, nV, 1);
}

ReturnValue LCQProblem::initializeSolver(bool needSubsolver)
{
    if (!loaded) return LCQPOBJECT_NOT_SETUP;
    // The four arms of src/LCQProblem.cpp:888-963.  Every arm runs on the HIP subsolver (the reference's qpOASES and OSQP
    // libraries are not part of this build); what an arm keeps of the reference is its contract: the dense/sparse mode it
    // insists on, the dual layout, and for OSQP_SPARSE the refusal of box constraints and the missing box duals.
    const QPSolver arm = options.getQPSolver();
    const int m = nC + 2 * nComp;
    if (arm == QPOASES_DENSE) {
        if (sparseSolver) return DENSE_SPARSE_MISSMATCH;                               // :892-894
        nDuals = nV + m; boxDualOffset = nV;
    } else if (arm == QPOASES_SPARSE) {
        if (!sparseSolver) return DENSE_SPARSE_MISSMATCH;                              // :913-915
        nDuals = nV + m; boxDualOffset = nV;
    } else if (arm == OSQP_SPARSE) {
        if (haveBox) return INVALID_OSQP_BOX_CONSTRAINTS;                              // :930-932, :956-958
        if (!sparseSolver) return DENSE_SPARSE_MISSMATCH;                              // :952-954
        nDuals = m; boxDualOffset = 0;                                                 // :934-935
    } else if (arm == HIP_DENSE) {
        nDuals = nV + m; boxDualOffset = nV;     // new arm: dense kernels, problem held in either mode
    } else {
        return NOT_YET_IMPLEMENTED;
    }
    // dual guess in the arm's layout; the OSQP arm drops the box part (:938-949 -- the reference shifts by nV too, but its
    // memcpy counts bytes, not doubles; the intent is restated here)
    yk.assign(nDuals, 0.0);
    for (int i = 0; i < nDuals; ++i) yk[i] = y0Full[(nV - boxDualOffset) + i];
    ysub.assign(nV + m, 0.0);
    if (!needSubsolver) {
        // the loop runs on the device: no per-QP plugin object
    } else if (sparseSolver) {
        // CSC problem data: the loop below runs on the CSC Utilities; the device subsolver of this arm is dense,
        // so it receives dense copies of Q and [A;L;R] (the reference hands the CSC arrays to qpOASES / OSQP instead,
        // src/LCQProblem.cpp:908-927,960)
        double *q = Utilities::csc_to_dns(Q_sparse), *a = Utilities::csc_to_dns(A_sparse);
        if (!q || !a) { delete[] q; delete[] a; return FAILED_SWITCH_TO_DENSE; }
        Subsolver tmp(nV, m, q, a, HIP_DENSE, device);
        subsolver = tmp;
        delete[] q; delete[] a;
    } else {
        Subsolver tmp(nV, m, Q.data(), A.data(), HIP_DENSE, device);
        subsolver = tmp;
    }
    if (needSubsolver) subsolver.setOptions(options.getHIPOptions());
    gTilde = g;
    phiConst = 0.0;
    gPhi.clear();
    if (haveLbL || haveLbR) {
        phiConst = Utilities::DotProduct(lbL.data(), lbR.data(), nComp);
        gPhi.assign(nV, 0.0);
        if (haveLbL) addRT(lbL.data(), gPhi.data());
        if (haveLbR) addLT(lbR.data(), gPhi.data());
        for (int i = 0; i < nV; ++i) gPhi[i] = -gPhi[i];
    }
    Qx.assign(nV, 0.0); Cx.assign(nV, 0.0); Qp.assign(nV, 0.0); Cp.assign(nV, 0.0);
    gk.assign(nV, 0.0); xnew.assign(nV, 0.0); pk.assign(nV, 0.0); statk.assign(nV, 0.0);
    constrStatk.assign(nV, 0.0); lkTmp.assign(nV, 0.0); ykA.assign(m, 0.0);
    alphak = 1.0;
    rho = options.getInitialPenaltyParameter();
    outerIter = innerIter = totalIter = 0;
    qpIterk = qpSolverExitFlag = 0;
    perturbCounter = 0;
    algoStat = PROBLEM_NOT_SOLVED;
    complHistory.clear();
    stats.reset();
    if (options.getPrintLevel() > NONE) std::printf("\n");
    return SUCCESSFUL_RETURN;
}

// per-iterate outputs of a device run: the tracking vectors of OutputStatistics (src/OutputStatistics.cpp:131-164) and the
// iteration table of printIteration (src/LCQProblem.cpp:1528-1576), rebuilt from the device trace
// (scalars: |statk|inf, phi, rho, alphak, objective, merit, |pk|inf, QP iterations)
void LCQProblem::finishFromTrace(const std::vector<double>& sc, const std::vector<double>& xs, int len)
{
    const PrintLevel pl = options.getPrintLevel();
    int outer = 0, inner = 0;
    for (int k = 0; k < len; ++k) {
        const double* s = &sc[(size_t)k * 8];
        if (k > 0 && s[2] != sc[(size_t)(k - 1) * 8 + 2]) { outer++; inner = 0; }      // a penalty update in the previous pass
        if (options.getStoreSteps())
            stats.updateTrackingVectors(&xs[(size_t)k * nV], inner, (int)s[7], s[3], s[6], s[0], s[4], s[1], s[5], nV);
        if (pl != NONE && !(pl == OUTER_LOOP_ITERATES && inner > 0)) {
            const bool in = pl >= INNER_LOOP_ITERATES;
            if ((in && inner % 10 == 0) || (!in && outer % 10 == 0))
                std::printf(in ? " outer |  inner |   station  |   complem  |     rho    |   norm p   |    alpha   | sub it\n"
                               : " outer |   station  |   complem  |     rho    |   norm p\n");
            std::printf("%6d", outer);
            if (in) std::printf(" | %6d", inner);
            std::printf(" | %10.3g | %10.3g | %10.3g | %10.3g", s[0], s[1], s[2], s[6]);
            if (in) std::printf(" | %10.3g | %6d", s[3], (int)s[7]);
            std::printf(" \n");
        }
        inner++;
    }
    if (pl != NONE) std::fflush(stdout);
}

ReturnValue LCQProblem::runOnDevice()
{
    const int m = nC + 2 * nComp;
    double *q = 0, *a = 0, *l = 0, *r = 0;
    if (sparseSolver) {
        q = Utilities::csc_to_dns(Q_sparse); a = Utilities::csc_to_dns(A_sparse); l = Utilities::csc_to_dns(L_sparse); r = Utilities::csc_to_dns(R_sparse);
        if (!q || !a || !l || !r) { delete[] q; delete[] a; delete[] l; delete[] r; return FAILED_SWITCH_TO_DENSE; }
    }
    const double *Qd = sparseSolver ? q : Q.data(), *Ad = sparseSolver ? a : A.data(), *Ld = sparseSolver ? l : L.data(), *Rd = sparseSolver ? r : R.data();
    lcqp_hip_batch_t* bt = lcqp_hip_batch_create(1, nV, nC, nComp, haveBox ? 1 : 0, device);
    ReturnValue ret = SUBPROBLEM_SOLVER_ERROR;
    if (bt) {
        lcqp_options_t o = options.getHIPOptions();
        const bool wantTrace = o.storeSteps != 0 || o.printLevel != 0;
        o.storeSteps = wantTrace ? 1 : 0;
        int rc = lcqp_hip_batch_set_options(bt, &o);
        if (!rc) rc = lcqp_hip_batch_load(bt, 0, 1, Qd, g.data(), Ld, Rd, haveLbL ? lbL.data() : 0, &ubA[nC], haveLbR ? lbR.data() : 0, &ubA[nC + nComp],
                                          nC > 0 ? Ad : 0, lbA.data(), ubA.data(), haveBox ? lb.data() : 0, haveBox ? ub.data() : 0, xk.data(),
                                          haveYk ? y0Full.data() : 0);
        if (!rc) rc = lcqp_hip_batch_run(bt);
        lcqp_stats_t st;
        std::memset(&st, 0, sizeof(st));
        yk.assign(nV + m, 0.0);
        if (!rc) rc = lcqp_hip_batch_get_solution(bt, xk.data(), yk.data(), &st);
        if (!rc) {
            stats.updateIterTotal(st.iterTotal); stats.updateIterOuter(st.iterOuter); stats.updateSubproblemIter(st.subproblemIter);
            stats.updateRhoOpt(st.rhoOpt); stats.updateQPSolverExitFlag(st.qpSolverExitFlag);
            algoStat = (AlgorithmStatus)st.status;
            stats.updateSolutionStatus(algoStat);
            if (wantTrace) {
                const int cap = std::min(st.iterTotal + 1, 1024);
                std::vector<double> sc((size_t)cap * 8), xs((size_t)cap * nV);
                int len = 0;
                if (!lcqp_hip_batch_get_trace(bt, 0, cap, sc.data(), xs.data(), &len)) finishFromTrace(sc, xs, len);
            }
            ret = (ReturnValue)st.returnValue;
        }
        lcqp_hip_batch_destroy(bt);
    }
    delete[] q; delete[] a; delete[] l; delete[] r;
    return ret;
}

// OSQP_SPARSE arm with CSC data: the sparse engine (lcqp_hip_sparse_*, band LDL' of the KKT matrix) when the pattern is banded; the
// tracking vectors and the iteration table are rebuilt from the device trace as on the dense path.  Returns false when the engine does
// not take the problem (pattern not banded, an option it does not support): the caller then runs the host loop over the dense kernels.
bool LCQProblem::runSparseOnDevice(ReturnValue& ret)
{
    if (!sparseSolver) return false;
    lcqp_hip_sparse_t* sb = lcqp_hip_sparse_create(1, nV, nC, nComp, Q_sparse->p, Q_sparse->i, A_sparse->p, A_sparse->i, device);
    if (!sb) return false;                        // not a banded pattern (lcqp_hip_sparse_last_error says so)
    const int m = nC + 2 * nComp;
    lcqp_options_t o = options.getHIPOptions();
    const bool wantTrace = o.storeSteps != 0 || o.printLevel != 0;
    o.storeSteps = wantTrace ? 1 : 0;
    int rc = lcqp_hip_sparse_set_options(sb, &o);
    if (rc) { lcqp_hip_sparse_destroy(sb); return false; }      // e.g. nDynamicPenalty > 64: the host loop takes it
    rc = lcqp_hip_sparse_load(sb, 0, 1, Q_sparse->x, g.data(), A_sparse->x, lbA.data(), ubA.data(), haveLbL ? lbL.data() : 0, &ubA[nC],
                              haveLbR ? lbR.data() : 0, &ubA[nC + nComp], xk.data(), haveYk ? yk.data() : 0);
    if (rc > 0 && rc < LCQP_HIP_ERROR) { lcqp_hip_sparse_destroy(sb); ret = (ReturnValue)rc; lastEngine = ENGINE_SPARSE_DEVICE; return true; }      // the reference's own code for bad problem data
    if (!rc) rc = lcqp_hip_sparse_run(sb);
    lcqp_stats_t st;
    std::memset(&st, 0, sizeof(st));
    yk.assign(m, 0.0);
    if (!rc) rc = lcqp_hip_sparse_get_solution(sb, xk.data(), yk.data(), &st);
    if (!rc) {
        stats.updateIterTotal(st.iterTotal); stats.updateIterOuter(st.iterOuter); stats.updateSubproblemIter(st.subproblemIter);
        stats.updateRhoOpt(st.rhoOpt); stats.updateQPSolverExitFlag(st.qpSolverExitFlag);
        algoStat = (AlgorithmStatus)st.status;
        stats.updateSolutionStatus(algoStat);
        if (wantTrace) {
            const int cap = std::min(st.iterTotal + 1, 4096);
            std::vector<double> sc((size_t)cap * 8), xs((size_t)cap * nV);
            int len = 0;
            if (!lcqp_hip_sparse_get_trace(sb, 0, cap, sc.data(), xs.data(), &len)) finishFromTrace(sc, xs, len);
        }
        ret = (ReturnValue)st.returnValue;
    } else {
        ret = SUBPROBLEM_SOLVER_ERROR;
    }
    lcqp_hip_sparse_destroy(sb);
    lastEngine = ENGINE_SPARSE_DEVICE;
    return true;
}

ReturnValue LCQProblem::runSolver()
{
    const QPSolver arm = options.getQPSolver();
    const bool deviceLoop = (arm == HIP_DENSE && !hostLoop);
    lastEngine = ENGINE_NONE;
    // the sparse engine is tried before anything dense is built for the OSQP_SPARSE arm (no dense copies of a 4096-variable problem, no
    // dense plugin object that would go unused)
    const bool trySparse = (arm == OSQP_SPARSE && !hostLoop && sparseSolver);
    ReturnValue ret = initializeSolver(!(deviceLoop || trySparse));
    if (ret != SUCCESSFUL_RETURN) return ret;
    if (deviceLoop) { lastEngine = ENGINE_DENSE_DEVICE; return runOnDevice(); }
    if (trySparse) {
        if (runSparseOnDevice(ret)) return ret;
        ret = initializeSolver(true);      // not a pattern for the sparse engine: the host loop over the dense kernels
        if (ret != SUCCESSFUL_RETURN) return ret;
    }
    lastEngine = ENGINE_HOST_LOOP;
    if (options.getSolveZeroPenaltyFirst()) gk = g;
    else updateLinearization();
    ret = solveQPSubproblem(true);
    if (ret != SUCCESSFUL_RETURN) return ret;
    stats.updateRhoOpt(rho);   // Qk = Q + rho C is applied as Q v + rho C v (dense and CSC mode alike)
    for (;;) {
        Utilities::WeightedVectorAdd(1, xk.data(), alphak, pk.data(), xk.data(), nV);
        updateStationarity();
        printIteration();
        if (options.getStoreSteps()) storeSteps();
        totalIter++; stats.updateIterTotal(1);
        innerIter++;
        if (leyfferCheckPositive()) {
            updatePenalty();
            outerIter++; stats.updateIterOuter(1); innerIter = 0;
        }
        updateLinearization();
        if (Utilities::MaxAbs(statk.data(), nV) < options.getStationarityTolerance()) {
            if (getPhi() < options.getComplementarityTolerance()) {
                determineStationarityType();   // reads the untransformed duals copied into ykA
                transformDuals();
                stats.updateSolutionStatus(algoStat);
                return SUCCESSFUL_RETURN;
            }
            updatePenalty();
            outerIter++; stats.updateIterOuter(1); innerIter = 0;
        }
        if (totalIter > options.getMaxIterations()) return MAX_ITERATIONS_REACHED;
        if (rho > options.getMaxPenaltyParameter()) return MAX_PENALTY_REACHED;
        updateLinearization();
        ret = solveQPSubproblem(false);
        if (ret != SUCCESSFUL_RETURN) return ret;
        if (options.getPerturbStep()) perturbStep();
        getOptimalStepLength();
    }
}

void LCQProblem::updateLinearization()
{
    mulC(xk.data(), Cx.data());
    for (int i = 0; i < nV; ++i) gk[i] = rho * Cx[i] + gTilde[i];
}

ReturnValue LCQProblem::solveQPSubproblem(bool initialSolve)
{
    // the subsolver speaks the qpOASES layout (box duals first); the OSQP arm has no box part
    // (boxDualOffset = 0, src/SubsolverOSQP.cpp:186-200): its guess is placed behind nV zeros and its result read from there
    const bool osqpArm = (boxDualOffset == 0);
    const int m = nC + 2 * nComp;
    const double* y0 = 0;
    if (haveYk) {
        if (osqpArm) { std::fill(ysub.begin(), ysub.begin() + nV, 0.0); std::copy(yk.begin(), yk.end(), ysub.begin() + nV); y0 = ysub.data(); }
        else y0 = yk.data();
    }
    ReturnValue ret = subsolver.solve(initialSolve, qpIterk, qpSolverExitFlag, gk.data(), lbA.data(), ubA.data(), xk.data(),
                                      y0, osqpArm ? 0 : lb.data(), osqpArm ? 0 : ub.data());
    stats.updateSubproblemIter(qpIterk);
    stats.updateQPSolverExitFlag(qpSolverExitFlag);
    haveYk = true;
    if (ret != SUCCESSFUL_RETURN) return ret;
    if (osqpArm) {
        subsolver.getSolution(xnew.data(), ysub.data());
        for (int i = 0; i < m; ++i) yk[i] = ysub[nV + i];
    } else {
        subsolver.getSolution(xnew.data(), yk.data());
    }
    for (int i = 0; i < m; ++i) ykA[i] = yk[boxDualOffset + i];
    Utilities::WeightedVectorAdd(1, xnew.data(), -1, xk.data(), pk.data(), nV);
    return SUCCESSFUL_RETURN;
}

double LCQProblem::getPhi()
{
    double lin = gPhi.empty() ? 0.0 : Utilities::DotProduct(gPhi.data(), xk.data(), nV);
    mulC(xk.data(), Cx.data());
    return phiConst + lin + Utilities::DotProduct(xk.data(), Cx.data(), nV) / 2.0;
}
double LCQProblem::getObj()
{
    mulQ(xk.data(), Qx.data());
    return Utilities::DotProduct(g.data(), xk.data(), nV) + Utilities::DotProduct(xk.data(), Qx.data(), nV) / 2.0;
}
double LCQProblem::getMerit()
{
    mulQ(xk.data(), Qx.data()); mulC(xk.data(), Cx.data());
    double s = 0.0;
    for (int i = 0; i < nV; ++i) s += xk[i] * (Qx[i] + rho * Cx[i]);
    return Utilities::DotProduct(g.data(), xk.data(), nV) + s / 2.0;
}

void LCQProblem::updatePenalty()
{
    if (options.getNDynamicPenalty() > 0) complHistory.clear();
    rho *= options.getPenaltyUpdateFactor();
    stats.updateRhoOpt(rho);
    if (!gPhi.empty()) Utilities::WeightedVectorAdd(1.0, g.data(), rho, gPhi.data(), gTilde.data(), nV);
}

void LCQProblem::getOptimalStepLength()
{
    mulQ(pk.data(), Qp.data()); mulC(pk.data(), Cp.data());
    mulQ(xk.data(), Qx.data()); mulC(xk.data(), Cx.data());
    double qk = 0.0;
    for (int i = 0; i < nV; ++i) { qk += pk[i] * (Qp[i] + rho * Cp[i]); lkTmp[i] = (Qx[i] + rho * Cx[i]) + gTilde[i]; }
    const double lk = Utilities::DotProduct(pk.data(), lkTmp.data(), nV);
    alphak = 1.0;
    if (qk > 0 && lk < 0) alphak = std::min(-lk / qk, 1.0);
}

void LCQProblem::updateStationarity()
{
    mulQ(xk.data(), Qx.data()); mulC(xk.data(), Cx.data());
    mulAT(ykA.data(), constrStatk.data());
    // the box term only on the arms that carry box duals (src/LCQProblem.cpp:1262-1269: lb / ub are null on the OSQP arm)
    for (int i = 0; i < nV; ++i) statk[i] = ((Qx[i] + rho * Cx[i]) + gTilde[i]) - constrStatk[i] - (boxDualOffset > 0 ? yk[i] : 0.0);
}

bool LCQProblem::leyfferCheckPositive()
{
    const size_t n = (size_t)std::max(0, options.getNDynamicPenalty());
    if (n == 0) return false;
    const double cur = getPhi();
    if (complHistory.size() < n) { complHistory.push_back(cur); return false; }
    if (getPhi() < options.getComplementarityTolerance()) { complHistory.pop_front(); complHistory.push_back(cur); return false; }
    bool flag = true;
    for (size_t i = 0; i < n; ++i)
        if (cur < options.getEtaDynamicPenalty() * complHistory[i]) { flag = false; break; }
    complHistory.pop_front();
    complHistory.push_back(cur);
    return flag;
}

void LCQProblem::perturbStep()
{
    // same counter-based stream as the device path (lcqp_dev.hpp) and the oracle
    const unsigned long long seed = options.getHIPOptions().perturbSeed;
    for (int i = 0; i < nV; ++i) {
        unsigned long long z = seed + (perturbCounter + (unsigned long long)i + 1ULL) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        xk[i] += ((int)(z % 3ULL) - 1) * Utilities::EPS;
    }
    perturbCounter += (unsigned long long)nV;
}

void LCQProblem::transformDuals()
{
    std::vector<double> tmp(nComp);
    mulR(xk.data(), tmp.data());
    for (int i = 0; i < nComp; ++i) yk[boxDualOffset + nC + i] -= rho * tmp[i];
    mulL(xk.data(), tmp.data());
    for (int i = 0; i < nComp; ++i) yk[boxDualOffset + nC + nComp + i] -= rho * tmp[i];
}

void LCQProblem::determineStationarityType()
{
    std::vector<double> Lx(nComp), Rx(nComp);
    mulL(xk.data(), Lx.data());
    mulR(xk.data(), Rx.data());
    const double ctol = options.getComplementarityTolerance();
    bool s = true, m = true;
    for (int i = 0; i < nComp; ++i) {
        if (!(Lx[i] <= ctol && Rx[i] <= ctol)) continue;   // weak complementarity set
        const double a = ykA[nC + i], b = ykA[nC + nComp + i];
        const double prod = a * b, mn = std::min(a, b);
        if (mn < 0) s = false;
        if (std::abs(prod) >= ctol && mn <= 0) {
            if (prod <= ctol) { algoStat = W_STATIONARY_SOLUTION; return; }
            m = false;
        }
    }
    algoStat = s ? S_STATIONARY_SOLUTION : (m ? M_STATIONARY_SOLUTION : C_STATIONARY_SOLUTION);
}

void LCQProblem::storeSteps()
{
    stats.updateTrackingVectors(xk.data(), innerIter, qpIterk, alphak, Utilities::MaxAbs(pk.data(), nV),
                                Utilities::MaxAbs(statk.data(), nV), getObj(), getPhi(), getMerit(), nV);
}

void LCQProblem::printIteration()
{
    const PrintLevel pl = options.getPrintLevel();
    if (pl == NONE) return;
    if (pl == OUTER_LOOP_ITERATES && innerIter > 0) return;
    const bool inner = pl >= INNER_LOOP_ITERATES;
    if ((inner && innerIter % 10 == 0) || (!inner && outerIter % 10 == 0)) {
        std::printf(inner ? " outer |  inner |   station  |   complem  |     rho    |   norm p   |    alpha   | sub it\n"
                          : " outer |   station  |   complem  |     rho    |   norm p\n");
    }
    std::printf("%6d", outerIter);
    if (inner) std::printf(" | %6d", innerIter);
    std::printf(" | %10.3g | %10.3g | %10.3g | %10.3g", Utilities::MaxAbs(statk.data(), nV), getPhi(), rho, Utilities::MaxAbs(pk.data(), nV));
    if (inner) std::printf(" | %10.3g | %6d", alphak, qpIterk);
    std::printf(" \n");
}

AlgorithmStatus LCQProblem::getPrimalSolution(double* const xOpt) const
{
    if (xOpt && !xk.empty()) std::copy(xk.begin(), xk.end(), xOpt);
    return algoStat;
}

AlgorithmStatus LCQProblem::getDualSolution(double* const yOpt) const
{
    if (yOpt && !yk.empty()) std::copy(yk.begin(), yk.end(), yOpt);
    return algoStat;
}

}  // namespace LCQPow
