// Subsolver: the by-value dispatcher LCQProblem owns (include/Subsolver.hpp, src/Subsolver.cpp:36-136),
// here with the HIP_DENSE arm.  The qpOASES / OSQP arms of the reference need third-party code that is
// not vendored; selecting them returns INVALID_QPSOLVER.
#ifndef LCQPOW_AMD_SUBSOLVER_HPP
#define LCQPOW_AMD_SUBSOLVER_HPP

#include "SubsolverHIP.hpp"

namespace LCQPow {

class Subsolver {
  public:
    Subsolver();
    Subsolver(int nV, int nC, const double* Q, const double* A, QPSolver qpSolver = HIP_DENSE, int device = 0);
    Subsolver(const Subsolver& rhs);
    Subsolver& operator=(const Subsolver& rhs);

    void getSolution(double* x, double* y);
    ReturnValue solve(bool initialSolve, int& iterations, int& exit_flag, const double* g, const double* lbA,
                      const double* ubA, const double* x0 = 0, const double* y0 = 0, const double* lb = 0,
                      const double* ub = 0);
    void setOptions(const lcqp_options_t& options);

  private:
    void copy(const Subsolver& rhs);
    QPSolver qpSolver;
    SubsolverHIP solverHIP;
};

}  // namespace LCQPow
#endif
