// The plugin interface, identical in shape to the reference's include/SubsolverBase.hpp:28-58.
#ifndef LCQPOW_AMD_SUBSOLVERBASE_HPP
#define LCQPOW_AMD_SUBSOLVERBASE_HPP

#include "Utilities.hpp"

namespace LCQPow {

class SubsolverBase {
  public:
    virtual ~SubsolverBase() {}
    virtual void getSolution(double* x, double* y) = 0;
    virtual ReturnValue solve(bool initialSolve, int& iterations, int& exit_flag, const double* const _g,
                              const double* const _lbA, const double* const _ubA, const double* const x0 = 0,
                              const double* const y0 = 0, const double* const _lb = 0, const double* const _ub = 0) = 0;
};

}  // namespace LCQPow
#endif
