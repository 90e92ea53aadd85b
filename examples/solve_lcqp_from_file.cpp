// An LCQP given as text files, as the reference's examples/solve_lcqp_from_file.cpp takes it: a directory with Q.txt, g.txt, L.txt, R.txt and,
// where the problem has them, lbL / ubL / lbR / ubR / A / lbA / ubA / lb / ub / x0 / y0 .txt (one number per line, matrices row by row;
// a file that does not exist is an absent argument).  Dimensions come from the files: nV = lines of g, nComp = lines of L / nV,
// nC = lines of A / nV.      usage: solve_lcqp_from_file <directory>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "LCQProblem.hpp"

using namespace LCQPow;

static long count_numbers(const std::string& path)
{
    std::ifstream f(path.c_str());
    if (!f) return -1;
    long n = 0;
    double v;
    while (f >> v) n++;
    return n;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::printf("usage: solve_lcqp_from_file <directory with Q.txt g.txt L.txt R.txt ...>\n"); return 1; }
    const std::string dir = argv[1];
    const char* names[15] = {"Q", "g", "L", "R", "lbL", "ubL", "lbR", "ubR", "A", "lbA", "ubA", "lb", "ub", "x0", "y0"};
    std::string path[15];
    const char* arg[15];
    for (int k = 0; k < 15; k++) {
        path[k] = dir + "/" + names[k] + ".txt";
        arg[k] = (count_numbers(path[k]) >= 0) ? path[k].c_str() : 0;      // absent file: absent argument
    }
    const long nV = count_numbers(path[1]), nL = count_numbers(path[2]), nA = count_numbers(path[8]);
    if (nV <= 0 || nL <= 0 || nL % nV != 0 || (nA > 0 && nA % nV != 0) || !arg[0] || !arg[3]) {
        std::printf("%s: need Q.txt, g.txt, L.txt, R.txt with matching sizes\n", dir.c_str());
        return 1;
    }
    const int nComp = (int)(nL / nV), nC = nA > 0 ? (int)(nA / nV) : 0;
    std::printf("LCQP from %s: nV = %ld, nC = %d, nComp = %d\n", dir.c_str(), nV, nC, nComp);

    LCQProblem lcqp((int)nV, nC, nComp);
    Options options;
    options.setPrintLevel(OUTER_LOOP_ITERATES);
    lcqp.setOptions(options);
    ReturnValue rc = lcqp.loadLCQP(arg[0], arg[1], arg[2], arg[3], arg[4], arg[5], arg[6], arg[7], arg[8], arg[9], arg[10], arg[11], arg[12], arg[13], arg[14]);
    if (rc != SUCCESSFUL_RETURN) { std::printf("Failed to load LCQP (%d).\n", (int)rc); return 1; }
    rc = lcqp.runSolver();
    if (rc != SUCCESSFUL_RETURN) { std::printf("Failed to solve LCQP (%d).\n", (int)rc); return 1; }

    std::vector<double> x(nV);
    OutputStatistics stats;
    lcqp.getPrimalSolution(x.data());
    lcqp.getOutputStatistics(stats);
    std::printf("\ni = %d; k = %d; rho = %g; WSR = %d; stationarity type = %d\nxOpt =", stats.getIterTotal(), stats.getIterOuter(), stats.getRhoOpt(),
                stats.getSubproblemIter(), (int)stats.getSolutionStatus());
    for (long i = 0; i < nV; i++) std::printf(" %.10g", x[i]);
    std::printf("\n");
    return 0;
}
