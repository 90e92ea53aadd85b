// The 2-variable toy LCQP of the reference's examples/warm_up.cpp (min (x1-1)^2 + (x2-1)^2, 0 <= x1 _|_ x2 >= 0)
// on the HIP backend: LCQProblem::loadLCQP / runSolver with QPSolver::HIP_DENSE.
#include <cstdio>

#include "LCQProblem.hpp"

using namespace LCQPow;

int main()
{
    double Q[2 * 2] = {2.0, 0.0, 0.0, 2.0};
    double g[2] = {-2.0, -2.0};
    double L[1 * 2] = {1.0, 0.0};
    double R[1 * 2] = {0.0, 1.0};
    double x0[2] = {1.0, 1.0};
    double y0[4] = {0.0, 0.0, 0.0, 0.0};
    const int nV = 2, nC = 0, nComp = 1;

    LCQProblem lcqp(nV, nC, nComp);
    Options options;
    options.setPrintLevel(INNER_LOOP_ITERATES);
    options.setQPSolver(HIP_DENSE);
    lcqp.setOptions(options);

    if (lcqp.loadLCQP(Q, g, L, R, 0, 0, 0, 0, 0, 0, 0, 0, 0, x0, y0) != SUCCESSFUL_RETURN) { std::printf("Failed to load LCQP.\n"); return 1; }
    if (lcqp.runSolver() != SUCCESSFUL_RETURN) { std::printf("Failed to solve LCQP.\n"); return 1; }

    double xOpt[2], yOpt[4];
    OutputStatistics stats;
    lcqp.getPrimalSolution(xOpt);
    lcqp.getDualSolution(yOpt);
    lcqp.getOutputStatistics(stats);
    std::printf("\nxOpt = [ %g, %g ];  yOpt = [ %g, %g, %g, %g ]; i = %d; k = %d; rho = %g; WSR = %d \n\n", xOpt[0], xOpt[1], yOpt[0],
                yOpt[1], yOpt[2], yOpt[3], stats.getIterTotal(), stats.getIterOuter(), stats.getRhoOpt(), stats.getSubproblemIter());
    return 0;
}
