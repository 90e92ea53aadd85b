// SubsolverHIP: the MI355X backend behind SubsolverBase.  Same constructor / copy / assignment /
// setOptions / solve / getSolution surface as SubsolverQPOASES (include/SubsolverQPOASES.hpp:36-110),
// implemented on the C ABI of include/lcqp_hip.h.
#ifndef LCQPOW_AMD_SUBSOLVERHIP_HPP
#define LCQPOW_AMD_SUBSOLVERHIP_HPP

#include "SubsolverBase.hpp"
#include "lcqp_hip.h"

namespace LCQPow {

class SubsolverHIP : public SubsolverBase {
  public:
    SubsolverHIP();
    // nC = number of stacked rows nC + 2*nComp, Q (nV x nV) and A (nC x nV) row-major; deep copies
    SubsolverHIP(int nV, int nC, const double* Q, const double* A, int device = 0);
    SubsolverHIP(const SubsolverHIP& rhs);
    virtual ~SubsolverHIP();
    SubsolverHIP& operator=(const SubsolverHIP& rhs);

    void setOptions(const lcqp_options_t& options);
    ReturnValue solve(bool initialSolve, int& iterations, int& exit_flag, const double* const g,
                      const double* const lbA, const double* const ubA, const double* const x0 = 0,
                      const double* const y0 = 0, const double* const lb = 0, const double* const ub = 0) override;
    void getSolution(double* x, double* y) override;

  private:
    void clear();
    lcqp_hip_qp_t* qp;
};

}  // namespace LCQPow
#endif
