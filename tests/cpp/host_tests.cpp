// C++ tests of the host layer, modelled on the reference's test/RunUnitTests.cpp and test/examples/*.cpp.
//   host_tests cpu   Utilities known answers, Options validation, OutputStatistics (no GPU needed)
//   host_tests gpu   SolverTest.RunWarmUp, CheckQPReturnFlag, example programs, OptimizeOnCircle,
//                    BatchLCQProblem -- through LCQProblem / Subsolver / SubsolverHIP on GPU 0
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "BatchLCQProblem.hpp"
#include "LCQProblem.hpp"

using namespace LCQPow;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static void test_utilities()
{   // test/RunUnitTests.cpp:33-246
    { double A[6] = {1, 0, 2, 3, 1, 1}, B[12] = {2, 0, 0, 2, 1, 0, 0, 1, 0, -1, -1, 0}, C[8];
      Utilities::MatrixMultiplication(A, B, C, 2, 3, 4);
      double e[8] = {2, -2, -2, 2, 7, -1, -1, 7}; for (int i = 0; i < 8; i++) CHECK(C[i] == e[i]); }
    { double A[6] = {1, 0, 2, 3, 1, 1}, B[2] = {98, -10}, C[3];
      Utilities::TransponsedMatrixMultiplication(A, B, C, 2, 3, 1);
      CHECK(C[0] == 68 && C[1] == -10 && C[2] == 186); }
    { double A[6] = {1, 0, 2, 3, 1, 1}, B[6] = {2, 0, 1, 0, 0, -1}, C[9];
      Utilities::MatrixSymmetrizationProduct(A, B, C, 2, 3);
      double e[9] = {4, 0, 2, 0, 0, -1, 2, -1, 2}; for (int i = 0; i < 9; i++) CHECK(C[i] == e[i]); }
    { double A[6] = {1, 0, 2, 3, 1, 1}, b[3] = {2, 0, 1}, c[2] = {-3, -3}, d[2];
      Utilities::AffineLinearTransformation(2, A, b, c, d, 2, 3);
      CHECK(d[0] == 5 && d[1] == 11); }
    { double A[6] = {0, 1, 3, 1, 10, 1}, B[6] = {2, 0, 0, 4, 2, 2}, C[6];
      Utilities::WeightedMatrixAdd(-1, A, 0.5, B, C, 3, 2);
      double e[6] = {1, -1, -3, 1, -9, 0}; for (int i = 0; i < 6; i++) CHECK(C[i] == e[i]); }
    { double a[4] = {0, 1, 2, 3}, b[4] = {10, 2, 0, 3}, d[4];
      Utilities::WeightedVectorAdd(2, a, -1, b, d, 4);
      CHECK(d[0] == -10 && d[1] == 0 && d[2] == 4 && d[3] == 3); }
    { double p[3] = {1, 2, 3}, Q[9] = {0, 1, 0, 1, 2, 1, 0, 1, 0}; CHECK(Utilities::QuadraticFormProduct(Q, p, 3) == 24); }
    { double a[4] = {0, 1, 2, 3}, b[4] = {10, 2, 0, 3}; CHECK(Utilities::DotProduct(a, b, 4) == 11); }
    { double a[4] = {0, 1, 2, 3}, b[4] = {0, -1, 2, 0}, c[4] = {0, -4, 2, 0};
      CHECK(Utilities::MaxAbs(a, 4) == 3 && Utilities::MaxAbs(b, 4) == 2 && Utilities::MaxAbs(c, 4) == 4); }
}

static void test_options()
{   // src/Options.cpp:85-259,296-318; test/RunUnitTests.cpp:249-262
    Options o;
    CHECK(o.getComplementarityTolerance() == 1e3 * Utilities::EPS && o.getStationarityTolerance() == 1e6 * Utilities::EPS);
    CHECK(o.getInitialPenaltyParameter() == 0.01 && o.getPenaltyUpdateFactor() == 2.0 && o.getMaxPenaltyParameter() == 1e8);
    CHECK(o.getSolveZeroPenaltyFirst() && o.getPerturbStep() && o.getMaxIterations() == 1000);
    CHECK(o.getNDynamicPenalty() == 3 && o.getEtaDynamicPenalty() == 0.9 && o.getPrintLevel() == INNER_LOOP_ITERATES && !o.getStoreSteps());
    CHECK(o.setStationarityTolerance(0) == INVALID_STATIONARITY_TOLERANCE && o.setComplementarityTolerance(-1) == INVALID_COMPLEMENTARITY_TOLERANCE);
    CHECK(o.setInitialPenaltyParameter(0) == INVALID_INITIAL_PENALTY_VALUE && o.setPenaltyUpdateFactor(1.0) == INVALID_PENALTY_UPDATE_VALUE);
    CHECK(o.setMaxIterations(0) == INVALID_MAX_ITERATIONS_VALUE && o.setMaxPenaltyParameter(0) == INVALID_MAX_RHO_VALUE);
    CHECK(o.setEtaDynamicPenalty(1.0) == INVALID_ETA_VALUE && o.setPrintLevel(3) == INVALID_PRINT_LEVEL_VALUE);
    CHECK(o.setQPSolver(4) == INVALID_QPSOLVER && o.setQPSolver(3) == SUCCESSFUL_RETURN && o.getQPSolver() == HIP_DENSE);
    o.setStationarityTolerance(1e-3);
    Options c(o), d; d = o;
    CHECK(c.getStationarityTolerance() == 1e-3 && d.getStationarityTolerance() == 1e-3);
    OutputStatistics s;
    CHECK(s.updateIterTotal(0) == INVALID_TOTAL_ITER_COUNT && s.updateIterTotal(2) == SUCCESSFUL_RETURN && s.getIterTotal() == 2);
    CHECK(s.updateRhoOpt(-1) == INVALID_RHO_OPT && s.updateSubproblemIter(-1) == IVALID_SUBPROBLEM_ITER);
}

static void test_csc_utilities()
{   // test/RunUnitTests.cpp:265-410
    { double x[3] = {2.0, 1.0, 2.0}; int i[3] = {0, 0, 1}, p[4] = {0, 1, 3, 4};
      csc* Q = Utilities::copyCSC(2, 3, 3, x, i, p);
      double* F = Utilities::csc_to_dns(Q);
      double e[6] = {2, 1, 0, 0, 2, 0}; for (int k = 0; k < 6; k++) CHECK(F[k] == e[k]);
      delete[] F; Utilities::ClearSparseMat(&Q); CHECK(Q == 0); }
    { double x[3] = {2.0, 1.0, 10.0}; int i[3] = {0, 1, 1}, p[4] = {0, 2, 2, 4};
      csc* Q = Utilities::copyCSC(2, 3, 3, x, i, p);
      double* F = Utilities::csc_to_dns(Q);
      double e[6] = {2, 0, 0, 1, 0, 10}; for (int k = 0; k < 6; k++) CHECK(F[k] == e[k]);
      delete[] F; Utilities::ClearSparseMat(&Q); }
    { double x[3] = {2.0, 10.0, 1.0}; int i[3] = {0, 2, 0}, p[3] = {0, 2, 3};
      csc* T = Utilities::copyCSC(3, 2, 3, x, i, p);
      double* F = Utilities::csc_to_dns(T);
      double e[6] = {2, 1, 0, 0, 10, 0}; for (int k = 0; k < 6; k++) CHECK(F[k] == e[k]);
      delete[] F; Utilities::ClearSparseMat(&T); }
    {   // SparseDenseBackAndForth :335-375 (fixed seed)
        unsigned s = 12345u;
        for (int rep = 0; rep < 100; rep++) {
            double Q[10];
            for (int j = 0; j < 10; j++) { s = s * 1664525u + 1013904223u; unsigned rd = s >> 8; Q[j] = (rd % 4 == 0) ? (double)(rd % 9) : 0.0; }
            csc* S = Utilities::dns_to_csc(Q, 2, 5);
            double* F = Utilities::csc_to_dns(S);
            for (int j = 0; j < 10; j++) CHECK(F[j] == Q[j]);
            delete[] F; Utilities::ClearSparseMat(&S);
        }
    }
    { double x[4] = {2.0, 3.0, 3.0, 2.0}; int i[4] = {0, 1, 0, 1}, p[3] = {0, 2, 4};   // CSCtoTriangular :378-410
      csc* M = Utilities::copyCSC(2, 2, 4, x, i, p);
      csc* U = Utilities::copyCSC(M, true);
      CHECK(U && U->p[0] == 0 && U->p[1] == 1 && U->p[2] == 3 && U->i[0] == 0 && U->i[1] == 0 && U->i[2] == 1);
      CHECK(U->x[0] == 2 && U->x[1] == 3 && U->x[2] == 2 && U->m == 2 && U->n == 2 && U->nz == -1 && U->nzmax == 3);
      Utilities::ClearSparseMat(&M); Utilities::ClearSparseMat(&U); }
    {   // sparse products against the dense known answers (:33-104, :107-129, :190-204)
        double A[6] = {1, 0, 2, 3, 1, 1}, B[6] = {2, 0, 1, 0, 0, -1};
        csc *As = Utilities::dns_to_csc(A, 2, 3), *Bs = Utilities::dns_to_csc(B, 2, 3);
        double b[3] = {2, 0, 1}, c[2], bt[2] = {98, -10}, ct[3];
        Utilities::MatrixMultiplication(As, b, c); CHECK(c[0] == 4 && c[1] == 7);
        Utilities::TransponsedMatrixMultiplication(As, bt, ct); CHECK(ct[0] == 68 && ct[1] == -10 && ct[2] == 186);
        csc* Cs = Utilities::MatrixSymmetrizationProduct(As, Bs);
        double* Cd = Utilities::csc_to_dns(Cs);
        double e[9] = {4, 0, 2, 0, 0, -1, 2, -1, 2}; for (int k = 0; k < 9; k++) CHECK(Cd[k] == e[k]);
        double Qd[9] = {0, 1, 0, 1, 2, 1, 0, 1, 0}, p3[3] = {1, 2, 3}, z[3] = {-3, -3, -3}, d[3];
        csc* Qs = Utilities::dns_to_csc(Qd, 3, 3);
        CHECK(Utilities::QuadraticFormProduct(Qs, p3, 3) == 24);
        Utilities::AffineLinearTransformation(2, Qs, p3, z, d, 3); CHECK(d[0] == 1 && d[1] == 13 && d[2] == 1);
        delete[] Cd;
        Utilities::ClearSparseMat(&As); Utilities::ClearSparseMat(&Bs); Utilities::ClearSparseMat(&Cs); Utilities::ClearSparseMat(&Qs);
    }
}

static const double Qw[4] = {2, 0, 0, 2}, gw[2] = {-2, -2}, Lw[2] = {1, 0}, Rw[2] = {0, 1};

static void test_dense_to_sparse()
{   // LoadDataTest.DenseToSparse, test/RunUnitTests.cpp:413-460: dense -> sparse -> solve -> dense -> solve
    LCQProblem lcqp(2, 0, 1);
    Options options; options.setPrintLevel(NONE); lcqp.setOptions(options);
    CHECK(lcqp.loadLCQP(Qw, gw, Lw, Rw) == SUCCESSFUL_RETURN);
    CHECK(lcqp.switchToDenseMode() == SUCCESSFUL_RETURN);
    CHECK(lcqp.switchToSparseMode() == SUCCESSFUL_RETURN);
    CHECK(lcqp.switchToSparseMode() == SUCCESSFUL_RETURN);
    CHECK(lcqp.runSolver() == SUCCESSFUL_RETURN);
    double xs[2]; lcqp.getPrimalSolution(xs);
    CHECK(lcqp.switchToDenseMode() == SUCCESSFUL_RETURN);
    CHECK(lcqp.runSolver() == SUCCESSFUL_RETURN);
    // sparse overload of loadLCQP (src/LCQProblem.cpp:390-441) on the w_A example
    double A[2] = {1, -1}, lbA[1] = {-0.5}, ubA[1] = {INFINITY};
    csc *Qs = Utilities::dns_to_csc(Qw, 2, 2), *Ls = Utilities::dns_to_csc(Lw, 1, 2), *Rs = Utilities::dns_to_csc(Rw, 1, 2), *As = Utilities::dns_to_csc(A, 1, 2);
    LCQProblem p2(2, 1, 1); p2.setOptions(options);
    CHECK(p2.loadLCQP(Qs, gw, Ls, Rs, 0, 0, 0, 0, As, lbA, ubA) == SUCCESSFUL_RETURN);
    CHECK(p2.runSolver() == SUCCESSFUL_RETURN);
    double x2[2]; p2.getPrimalSolution(x2);
    const double tol = options.getStationarityTolerance();
    // the two strongly stationary points of the w_A example (test/examples/warm_up_w_A.cpp:32-41): (1, 0), and on the branch x1 = 0 the
    // constraint x1 - x2 >= -0.5 stops x2 at 0.5; which one a run reaches depends on the perturbation (perturbStep, src/LCQProblem.cpp:1353-1362)
    if (!((std::fabs(x2[0] - 1) <= tol && std::fabs(x2[1]) <= tol) || (std::fabs(x2[1] - 0.5) <= tol && std::fabs(x2[0]) <= tol))) printf("w_A (CSC) xOpt = [ %.17g, %.17g ]\n", x2[0], x2[1]);
    CHECK((std::fabs(x2[0] - 1) <= tol && std::fabs(x2[1]) <= tol) || (std::fabs(x2[1] - 0.5) <= tol && std::fabs(x2[0]) <= tol));
    Utilities::ClearSparseMat(&Qs); Utilities::ClearSparseMat(&Ls); Utilities::ClearSparseMat(&Rs); Utilities::ClearSparseMat(&As);
}

static void test_run_warm_up()
{   // SolverTest.RunWarmUp, test/RunUnitTests.cpp:505-551 (20 repetitions, seeds instead of time(NULL))
    LCQProblem lcqp(2, 0, 1);
    Options options; options.setPrintLevel(NONE);
    int found1 = 0, found2 = 0;
    for (int i = 0; i < 20; i++) {
        options.setPerturbSeed(1000 + i);
        lcqp.setOptions(options);
        CHECK(lcqp.loadLCQP(Qw, gw, Lw, Rw) == SUCCESSFUL_RETURN);
        CHECK(lcqp.runSolver() == SUCCESSFUL_RETURN);
        double x[2], y[4];
        lcqp.getPrimalSolution(x); lcqp.getDualSolution(y);
        const double tol = options.getStationarityTolerance();
        const bool s1 = std::fabs(x[0] - 1) <= tol && std::fabs(x[1]) <= tol, s2 = std::fabs(x[1] - 1) <= tol && std::fabs(x[0]) <= tol;
        CHECK(s1 || s2);
        found1 += s1; found2 += s2;
        CHECK(std::fabs(2 * x[0] - 2 - y[0] - y[2]) <= tol && std::fabs(2 * x[1] - 2 - y[1] - y[3]) <= tol);
    }
    CHECK(found1 > 0 && found2 > 0);
    CHECK(lcqp.getNumberOfPrimals() == 2 && lcqp.getNumberOfDuals() == 4);
}

static void test_qp_return_flag()
{   // OutputStatisticsTest.CheckQPReturnFlag, test/RunUnitTests.cpp:463-502
    double A[2] = {1, 0}, lbA[1] = {0}, ubA[1] = {-1};
    LCQProblem lcqp(2, 1, 1);
    Options options; options.setPrintLevel(NONE); lcqp.setOptions(options);
    CHECK(lcqp.loadLCQP(Qw, gw, Lw, Rw, 0, 0, 0, 0, A, lbA, ubA) == SUCCESSFUL_RETURN);
    CHECK(lcqp.runSolver() == SUBPROBLEM_SOLVER_ERROR);
    OutputStatistics stats; lcqp.getOutputStatistics(stats);
    CHECK(stats.getQPSolverExitFlag() != 0);
}

static void test_examples()
{   // test/examples/warm_up.cpp, warm_up_w_A.cpp, warm_up_binary.cpp, test_max_penalty.cpp, warm_up_store_steps.cpp
    Options options; options.setPrintLevel(NONE);
    { LCQProblem p(2, 0, 1); p.setOptions(options); double x0[2] = {1, 1}, y0[4] = {0, 0, 0, 0};
      CHECK(p.loadLCQP(Qw, gw, Lw, Rw, 0, 0, 0, 0, 0, 0, 0, 0, 0, x0, y0) == SUCCESSFUL_RETURN); CHECK(p.runSolver() == SUCCESSFUL_RETURN); }
    { LCQProblem p(2, 1, 1); p.setOptions(options); double A[2] = {1, -1}, lbA[1] = {-0.5}, ubA[1] = {INFINITY};
      CHECK(p.loadLCQP(Qw, gw, Lw, Rw, 0, 0, 0, 0, A, lbA, ubA) == SUCCESSFUL_RETURN); CHECK(p.runSolver() == SUCCESSFUL_RETURN); }
    { LCQProblem p(2, 0, 2); p.setOptions(options); double L[4] = {1, 0, 1, 0}, R[4] = {0, 1, -1, 0}, lbL[2] = {0, 0}, lbR[2] = {0, -0.5}, x0[2] = {0, 0};
      CHECK(p.loadLCQP(Qw, gw, L, R, lbL, 0, lbR, 0, 0, 0, 0, 0, 0, x0) == SUCCESSFUL_RETURN); CHECK(p.runSolver() == SUCCESSFUL_RETURN); }
    { Options o2 = options; o2.setMaxPenaltyParameter(1); LCQProblem p(2, 0, 1); p.setOptions(o2); double x0[2] = {1, 1}, y0[4] = {0, 0, 0, 0};
      CHECK(p.loadLCQP(Qw, gw, Lw, Rw, 0, 0, 0, 0, 0, 0, 0, 0, 0, x0, y0) == SUCCESSFUL_RETURN); CHECK(p.runSolver() == MAX_PENALTY_REACHED); }
    { Options o2 = options; o2.setStoreSteps(true); LCQProblem p(2, 0, 1); p.setOptions(o2);
      CHECK(p.loadLCQP(Qw, gw, Lw, Rw) == SUCCESSFUL_RETURN); CHECK(p.runSolver() == SUCCESSFUL_RETURN);
      OutputStatistics st; p.getOutputStatistics(st);
      CHECK((int)st.getxStepsStdVec().size() == st.getIterTotal() && (int)st.getPhiValsStdVec().size() == st.getIterTotal()); }
}

static void test_circle(bool print, bool hostLoop = false, double* xOut = 0, int* itOut = 0)
{   // examples/OptimizeOnCircle.cpp:32-99, dense path (BASELINE config C2)
    const int N = 100, nV = 2 + 2 * N, nC = N + 1, nComp = N;
    std::vector<double> Q(nV * nV, 0.0), g(nV, 0.0), L(nComp * nV, 0.0), R(nComp * nV, 0.0), A(nC * nV, 0.0), lbA(nC, 1.0), ubA(nC, 1.0), x0(nV, 0.0);
    const double xr[2] = {0.5, -0.6};
    x0[0] = xr[0]; x0[1] = xr[1];
    Q[0] = 17; Q[nV + 1] = 17; Q[1] = -15; Q[nV] = -15;
    for (int i = 2; i < nV; i++) Q[i * nV + i] = 5e-12;
    g[0] = -(17 * xr[0] - 15 * xr[1]); g[1] = -(-15 * xr[0] + 17 * xr[1]);
    for (int i = 0; i < N; i++) {
        A[i * nV + 0] = std::cos((2 * M_PI * i) / N); A[i * nV + 1] = std::sin((2 * M_PI * i) / N); A[i * nV + 2 + 2 * i] = 1;
        A[N * nV + 3 + 2 * i] = 1; L[i * nV + 2 + 2 * i] = 1; R[i * nV + 3 + 2 * i] = 1;
        x0[2 * i + 2] = 1; x0[2 * i + 3] = 1;
    }
    LCQProblem lcqp(nV, nC, nComp);
    Options options; options.setPrintLevel(print ? INNER_LOOP_ITERATES : NONE); options.setPerturbStep(false);
    lcqp.setOptions(options);
    lcqp.setHostLoop(hostLoop);
    CHECK(lcqp.loadLCQP(Q.data(), g.data(), L.data(), R.data(), 0, 0, 0, 0, A.data(), lbA.data(), ubA.data(), 0, 0, x0.data()) == SUCCESSFUL_RETURN);
    CHECK(lcqp.runSolver() == SUCCESSFUL_RETURN);
    std::vector<double> x(nV), y(nV + nC + 2 * nComp);
    lcqp.getPrimalSolution(x.data()); lcqp.getDualSolution(y.data());
    OutputStatistics st; lcqp.getOutputStatistics(st);
    if (xOut) { xOut[0] = x[0]; xOut[1] = x[1]; }
    if (itOut) { itOut[0] = st.getIterTotal(); itOut[1] = st.getIterOuter(); }
    std::printf("%scircle xOpt = [ %.10g, %.10g ]; i = %d; k = %d; rho = %g; WSR = %d; status = %d\n", hostLoop ? "host-loop " : "", x[0], x[1], st.getIterTotal(), st.getIterOuter(), st.getRhoOpt(), st.getSubproblemIter(), (int)st.getSolutionStatus());
    const bool glob = std::fabs(x[0] - 0.1811) < 1e-4 && std::fabs(x[1] + 0.9835) < 1e-4, loc = std::fabs(x[0] - 0.9764) < 1e-4 && std::fabs(x[1] + 0.2183) < 1e-4;
    CHECK(glob || loc);   // examples/OptimizeOnCircle.cpp:144-145
}

static void test_loops_agree()
{   // runSolver with the homotopy on the device (HIP_DENSE default: a batch of one through k_lcqp_run) and with the reference's host
    // loop over the SubsolverHIP plugin (setHostLoop) take the same path: same iterate counts, same minimiser
    double xd[2], xh[2]; int itd[2], ith[2];
    test_circle(false, false, xd, itd);
    test_circle(false, true, xh, ith);
    CHECK(itd[0] == ith[0] && itd[1] == ith[1]);
    CHECK(std::fabs(xd[0] - xh[0]) < 1e-8 && std::fabs(xd[1] - xh[1]) < 1e-8);
    Options options; options.setPrintLevel(NONE); options.setPerturbStep(false);
    double xw[2][2];
    for (int hl = 0; hl < 2; hl++) {
        LCQProblem p(2, 0, 1); p.setOptions(options); p.setHostLoop(hl != 0);
        double x0[2] = {1, 1}, y0[4] = {0, 0, 0, 0};
        CHECK(p.loadLCQP(Qw, gw, Lw, Rw, 0, 0, 0, 0, 0, 0, 0, 0, 0, x0, y0) == SUCCESSFUL_RETURN);
        CHECK(p.runSolver() == SUCCESSFUL_RETURN);
        p.getPrimalSolution(xw[hl]);
        CHECK(std::fabs(xw[hl][0] * xw[hl][1]) <= options.getComplementarityTolerance());
        OutputStatistics st; p.getOutputStatistics(st);
        CHECK(st.getSolutionStatus() >= W_STATIONARY_SOLUTION);
    }
    CHECK(std::fabs(xw[0][0] - xw[1][0]) < 1e-9 && std::fabs(xw[0][1] - xw[1][1]) < 1e-9);
}

static void test_batch()
{
    const int B = 8, n = 64, nC = 96, nComp = 16;
    BatchLCQProblem bt(B, n, nC, nComp);
    CHECK(bt.ok());
    Options options; options.setPrintLevel(NONE); options.setPerturbStep(false);
    CHECK(bt.setOptions(options) == SUCCESSFUL_RETURN);
    CHECK(bt.generateSynthetic(0x4C43515000000001ULL, 0) == SUCCESSFUL_RETURN);
    CHECK(bt.runSolver() == SUCCESSFUL_RETURN);
    std::vector<double> x(n);
    for (int i = 0; i < B; i++) {
        CHECK(bt.getReturnValue(i) == SUCCESSFUL_RETURN);
        AlgorithmStatus s = bt.getPrimalSolution(i, x.data());
        CHECK(s >= W_STATIONARY_SOLUTION);
        double phi = 0; for (int k = 0; k < nComp; k++) phi += x[k] * x[nComp + k];
        CHECK(std::fabs(phi) < 1e3 * Utilities::EPS);
    }
}

static void test_mixed_batch()
{   // LCQPow::MixedBatchLCQProblem: problems of three shapes, some with shifted complementarity bounds, in one call; every instance equals its solo
    // run through a batch of one (the bits: same kernels, same shape)
    const int shapes[3][3] = {{10, 4, 3}, {33, 12, 6}, {64, 0, 16}};
    std::vector<std::vector<double> > store;
    struct Ref { int nV, nC, nComp; const double *Q, *g, *L, *R, *lbL, *lbR, *A, *lbA, *ubA; };
    std::vector<Ref> refs;
    unsigned long long state = 0x1234567ULL;
    auto rnd = [&]() { state = state * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((state >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53); };
    for (int rep = 0; rep < 2; rep++)
        for (int s = 0; s < 3; s++) {
            const int n = shapes[s][0], nC = shapes[s][1], nK = shapes[s][2];
            std::vector<double> Q((size_t)n * n, 0.0), g(n), L((size_t)nK * n, 0.0), R((size_t)nK * n, 0.0), A((size_t)nC * n), lbA(nC), ubA(nC), lbL(nK), lbR(nK);
            for (int i = 0; i < n; i++) { Q[(size_t)i * n + i] = 1.0 + rnd(); g[i] = 2.0 * rnd() - 1.0; }
            for (int i = 0; i + 1 < n; i++) { const double v = 0.2 * rnd(); Q[(size_t)i * n + i + 1] = v; Q[(size_t)(i + 1) * n + i] = v; }
            for (int i = 0; i < nK; i++) { L[(size_t)i * n + i] = 1.0; R[(size_t)i * n + nK + i] = 1.0; lbL[i] = -0.1 * rnd(); lbR[i] = -0.1 * rnd(); }
            for (int r = 0; r < nC; r++) { for (int k = 0; k < n; k++) A[(size_t)r * n + k] = (2.0 * rnd() - 1.0) / 4.0; lbA[r] = -1.0 - rnd(); ubA[r] = 1.0 + rnd(); }
            const size_t b0 = store.size();
            store.push_back(Q); store.push_back(g); store.push_back(L); store.push_back(R); store.push_back(lbL); store.push_back(lbR); store.push_back(A); store.push_back(lbA); store.push_back(ubA);
            (void)b0;
        }
    for (size_t p = 0; p < store.size() / 9; p++) {
        const int s = (int)(p % 3);
        const std::vector<double>* v = &store[9 * p];
        const bool shifted = (p >= 3);      // the second repetition carries lbL / lbR, the first does not: mixed within every bucket
        Ref r = {shapes[s][0], shapes[s][1], shapes[s][2], v[0].data(), v[1].data(), v[2].data(), v[3].data(), shifted ? v[4].data() : 0, shifted ? v[5].data() : 0,
                 shapes[s][1] ? v[6].data() : 0, shapes[s][1] ? v[7].data() : 0, shapes[s][1] ? v[8].data() : 0};
        refs.push_back(r);
    }
    Options options; options.setPrintLevel(NONE); options.setPerturbStep(false);
    MixedBatchLCQProblem mixed;
    CHECK(mixed.setOptions(options) == SUCCESSFUL_RETURN);
    for (size_t p = 0; p < refs.size(); p++) {
        const Ref& r = refs[p];
        CHECK(mixed.addProblem(r.nV, r.nC, r.nComp, r.Q, r.g, r.L, r.R, r.lbL, 0, r.lbR, 0, r.A, r.lbA, r.ubA) == (int)p);
    }
    CHECK(mixed.runSolver() == SUCCESSFUL_RETURN);
    CHECK(mixed.numberOfBuckets() == 3 && mixed.size() == 6);
    for (size_t p = 0; p < refs.size(); p++) {
        const Ref& r = refs[p];
        BatchLCQProblem solo(1, r.nV, r.nC, r.nComp);
        CHECK(solo.ok() && solo.setOptions(options) == SUCCESSFUL_RETURN);
        CHECK(solo.loadLCQP(0, r.Q, r.g, r.L, r.R, r.lbL, 0, r.lbR, 0, r.A, r.lbA, r.ubA) == SUCCESSFUL_RETURN);
        CHECK(solo.runSolver() == SUCCESSFUL_RETURN);
        CHECK(mixed.getReturnValue((int)p) == SUCCESSFUL_RETURN && solo.getReturnValue(0) == SUCCESSFUL_RETURN);
        std::vector<double> xm(r.nV), xs(r.nV), ym(mixed.getNumberOfDuals((int)p)), ys(ym.size());
        CHECK(mixed.getPrimalSolution((int)p, xm.data()) == solo.getPrimalSolution(0, xs.data()));
        mixed.getDualSolution((int)p, ym.data()); solo.getDualSolution(0, ys.data());
        for (int k = 0; k < r.nV; k++) CHECK(xm[k] == xs[k]);
        for (size_t k = 0; k < ym.size(); k++) CHECK(ym[k] == ys[k]);
        CHECK(mixed.getStats((int)p).iterTotal == solo.getStats(0).iterTotal);
    }
}

static void test_pipeline()
{   // LCQPow::BatchPipeline: two batch objects in flight; the slots' bookkeeping (ADVICE, round 4): acquire() twice without a launch in between
    // must not dereference an empty launch order, a slot handed out with results and not launched again is free again, a drained pipeline
    // starts over with every slot free; every run of the same instances gives the same bits
    const int B = 6, n = 64, nC = 96, nComp = 16;
    Options options; options.setPrintLevel(NONE); options.setPerturbStep(false);
    BatchPipeline pipe(2, B, n, nC, nComp);
    CHECK(pipe.ok());
    for (int k = 0; k < pipe.depth(); k++) { CHECK(pipe.slot(k).setOptions(options) == SUCCESSFUL_RETURN); CHECK(pipe.slot(k).generateSynthetic(0x4C43515000000001ULL, 0) == SUCCESSFUL_RETURN); }
    BatchLCQProblem& a0 = pipe.acquire();
    BatchLCQProblem& a1 = pipe.acquire();          // nothing launched: the same free slot again, no crash
    CHECK(&a0 == &a1 && !pipe.hasResults());
    std::vector<double> xref(n), x(n);
    int launched = 0, collected = 0;
    for (int step = 0; step < 5; step++) {
        BatchLCQProblem& b = pipe.acquire();
        if (pipe.hasResults()) { collected++; CHECK(b.getReturnValue(0) == SUCCESSFUL_RETURN); }
        CHECK(pipe.launch(b) == SUCCESSFUL_RETURN); launched++;
    }
    while (BatchLCQProblem* b = pipe.drain()) {
        collected++;
        for (int i = 0; i < B; i++) CHECK(b->getReturnValue(i) == SUCCESSFUL_RETURN);
        b->getPrimalSolution(B - 1, x.data());
        if (collected == launched - 1) xref = x;
        if (collected == launched) for (int k = 0; k < n; k++) CHECK(x[k] == xref[k]);      // same instances, another slot: the same bits
    }
    CHECK(collected == launched);
    BatchLCQProblem& again = pipe.acquire();        // drained: every slot is free, nothing to wait for
    CHECK(!pipe.hasResults());
    CHECK(pipe.launch(again) == SUCCESSFUL_RETURN);
    CHECK(pipe.drain() == &again && pipe.drain() == 0);
}

static int run_from_files(const char* dir, int nV, int nC, int nComp)
{   // examples/solve_lcqp_from_file.cpp: loadLCQP(file names) + runSolver, then print the solution
    auto f = [&](const char* name) { static std::vector<std::string> keep; keep.push_back(std::string(dir) + "/" + name + ".txt"); return keep.back().c_str(); };
    LCQProblem lcqp(nV, nC, nComp);
    Options options; options.setPrintLevel(NONE); options.setPerturbStep(false);
    lcqp.setOptions(options);
    ReturnValue rc = lcqp.loadLCQP(f("Q"), f("g"), f("L"), f("R"), f("lbL"), f("ubL"), f("lbR"), f("ubR"), f("A"), f("lbA"), f("ubA"), f("lb"), f("ub"), f("x0"));
    if (rc != SUCCESSFUL_RETURN) { std::printf("load failed %d\n", (int)rc); return 1; }
    rc = lcqp.runSolver();
    std::vector<double> x(nV);
    lcqp.getPrimalSolution(x.data());
    OutputStatistics st; lcqp.getOutputStatistics(st);
    std::printf("files ret = %d; i = %d; k = %d; rho = %g; status = %d\nx =", (int)rc, st.getIterTotal(), st.getIterOuter(), st.getRhoOpt(), (int)st.getSolutionStatus());
    for (int i = 0; i < nV; i++) std::printf(" %.17g", x[i]);
    std::printf("\n");
    return rc == SUCCESSFUL_RETURN ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc > 5 && !std::strcmp(argv[1], "files")) return run_from_files(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]));
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
    test_utilities();
    test_csc_utilities();
    test_options();
    if (gpu) {
        if (lcqp_hip_device_count() < 1) { std::printf("FAIL no GPU visible\n"); return 2; }
        test_run_warm_up();
        test_qp_return_flag();
        test_examples();
        test_dense_to_sparse();
        test_circle(argc > 2);
        test_loops_agree();
        test_batch();
        test_pipeline();
        test_mixed_batch();
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED%.0d\n", failures);
    return failures ? 1 : 0;
}
