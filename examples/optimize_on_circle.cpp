// The problem of the reference's examples/OptimizeOnCircle.cpp on the HIP backend: the point of the unit circle -- approximated by the
// tangents at N angles, of which a complementarity selects one -- closest to x_ref in the norm of Q = [17 -15; -15 17].
//   min 1/2 (x - x_ref)' Q (x - x_ref)   s.t.  cos(t_i) x_1 + sin(t_i) x_2 + s_i = 1,  sum_i z_i = 1,  0 <= s_i _|_ z_i >= 0
// Given in compressed sparse columns and solved with QPSolver::OSQP_SPARSE, which this backend runs on its sparse engine (the KKT graph
// is a band plus three border nodes: x_1, x_2 and the coupling row); `optimize_on_circle [N] dense` takes the dense path instead.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "LCQProblem.hpp"

using namespace LCQPow;

int main(int argc, char** argv)
{
    const int N = (argc > 1) ? std::atoi(argv[1]) : 100;
    const bool dense = argc > 2 && !std::strcmp(argv[2], "dense");
    if (N < 3) { std::printf("usage: optimize_on_circle [N >= 3] [dense]\n"); return 1; }
    const int nV = 2 + 2 * N, nC = N + 1, nComp = N;
    const double pi = 3.14159265358979323846, xref[2] = {0.5, -0.6};
    std::vector<double> Q((size_t)nV * nV, 0.0), g(nV, 0.0), L((size_t)nComp * nV, 0.0), R((size_t)nComp * nV, 0.0), A((size_t)nC * nV, 0.0);
    std::vector<double> lbA(nC, 1.0), ubA(nC, 1.0), x0(nV, 1.0);
    Q[0] = Q[(size_t)nV + 1] = 17.0; Q[1] = Q[nV] = -15.0;
    for (int i = 2; i < nV; i++) Q[(size_t)i * nV + i] = 5e-12;      // the slack and selector variables enter the objective with a tiny weight only
    g[0] = -(17.0 * xref[0] - 15.0 * xref[1]);
    g[1] = -(-15.0 * xref[0] + 17.0 * xref[1]);
    x0[0] = xref[0]; x0[1] = xref[1];
    for (int i = 0; i < N; i++) {
        const double t = 2.0 * pi * i / N;
        A[(size_t)i * nV] = std::cos(t); A[(size_t)i * nV + 1] = std::sin(t); A[(size_t)i * nV + 2 + 2 * i] = 1.0;      // tangent i with its slack
        A[(size_t)N * nV + 3 + 2 * i] = 1.0;                                                                            // the selectors sum to one
        L[(size_t)i * nV + 2 + 2 * i] = 1.0;
        R[(size_t)i * nV + 3 + 2 * i] = 1.0;
    }

    LCQProblem lcqp(nV, nC, nComp);
    Options options;
    options.setPrintLevel(OUTER_LOOP_ITERATES);
    options.setQPSolver(dense ? HIP_DENSE : OSQP_SPARSE);
    lcqp.setOptions(options);
    ReturnValue rc;
    if (dense) {
        rc = lcqp.loadLCQP(Q.data(), g.data(), L.data(), R.data(), 0, 0, 0, 0, A.data(), lbA.data(), ubA.data(), 0, 0, x0.data());
    } else {
        csc *Qs = Utilities::dns_to_csc(Q.data(), nV, nV), *Ls = Utilities::dns_to_csc(L.data(), nComp, nV), *Rs = Utilities::dns_to_csc(R.data(), nComp, nV),
            *As = Utilities::dns_to_csc(A.data(), nC, nV);
        rc = lcqp.loadLCQP(Qs, g.data(), Ls, Rs, 0, 0, 0, 0, As, lbA.data(), ubA.data(), 0, 0, x0.data());
        Utilities::ClearSparseMat(&Qs); Utilities::ClearSparseMat(&Ls); Utilities::ClearSparseMat(&Rs); Utilities::ClearSparseMat(&As);
    }
    if (rc != SUCCESSFUL_RETURN) { std::printf("Failed to load LCQP (%d).\n", (int)rc); return 1; }
    rc = lcqp.runSolver();
    if (rc != SUCCESSFUL_RETURN) { std::printf("Failed to solve LCQP (%d).\n", (int)rc); return 1; }

    std::vector<double> x(nV);
    OutputStatistics stats;
    lcqp.getPrimalSolution(x.data());
    lcqp.getOutputStatistics(stats);
    std::printf("\nxOpt = [ %.6f, %.6f ]; |xOpt| = %.6f; i = %d; k = %d; rho = %g; WSR = %d; stationarity type = %d\n", x[0], x[1],
                std::sqrt(x[0] * x[0] + x[1] * x[1]), stats.getIterTotal(), stats.getIterOuter(), stats.getRhoOpt(), stats.getSubproblemIter(),
                (int)stats.getSolutionStatus());
    return 0;
}
