// BatchLCQProblem: B independent dense LCQPs of one shape solved entirely on one MI355X (one persistent
// workgroup per instance).  An extension behind the reference surface -- the reference has no batched
// API -- whose per-instance semantics are those of LCQProblem::loadLCQP / runSolver.
#ifndef LCQPOW_AMD_BATCHLCQPROBLEM_HPP
#define LCQPOW_AMD_BATCHLCQPROBLEM_HPP

#include <vector>

#include "Options.hpp"
#include "OutputStatistics.hpp"

namespace LCQPow {

class BatchLCQProblem {
  public:
    BatchLCQProblem(int batch, int nV, int nC, int nComp, bool withBoxBounds = false, int device = 0)
        : B(batch), nV_(nV), nC_(nC), nComp_(nComp), h(lcqp_hip_batch_create(batch, nV, nC, nComp, withBoxBounds ? 1 : 0, device)) {}
    ~BatchLCQProblem() { if (h) lcqp_hip_batch_destroy(h); }
    BatchLCQProblem(const BatchLCQProblem&) = delete;
    BatchLCQProblem& operator=(const BatchLCQProblem&) = delete;

    bool ok() const { return h != nullptr; }
    ReturnValue setOptions(const Options& o) { return (ReturnValue)lcqp_hip_batch_set_options(h, &o.getHIPOptions()); }
    // instance-by-instance load with the argument list of LCQProblem::loadLCQP
    ReturnValue loadLCQP(int instance, const double* Q, const double* g, const double* L, const double* R,
                         const double* lbL = 0, const double* ubL = 0, const double* lbR = 0, const double* ubR = 0,
                         const double* A = 0, const double* lbA = 0, const double* ubA = 0, const double* lb = 0,
                         const double* ub = 0, const double* x0 = 0, const double* y0 = 0)
    {
        return (ReturnValue)lcqp_hip_batch_load(h, instance, 1, Q, g, L, R, lbL, ubL, lbR, ubR, A, lbA, ubA, lb, ub, x0, y0);
    }
    ReturnValue generateSynthetic(unsigned long long seed0, unsigned long long firstInstance)
    {
        return (ReturnValue)lcqp_hip_batch_generate_synthetic(h, seed0, firstInstance);
    }
    // runSolver for every instance; per-instance return values are in getReturnValue(i)
    ReturnValue runSolver()
    {
        int rc = lcqp_hip_batch_run(h);
        if (rc) return (ReturnValue)rc;
        x.assign((size_t)B * nV_, 0.0);
        y.assign((size_t)B * (nV_ + nC_ + 2 * nComp_), 0.0);
        st.assign(B, lcqp_stats_t());
        return (ReturnValue)lcqp_hip_batch_get_solution(h, x.data(), y.data(), st.data());
    }
    ReturnValue getReturnValue(int i) const { return (ReturnValue)st[i].returnValue; }
    AlgorithmStatus getPrimalSolution(int i, double* xOpt) const
    {
        for (int k = 0; k < nV_; ++k) xOpt[k] = x[(size_t)i * nV_ + k];
        return (AlgorithmStatus)st[i].status;
    }
    AlgorithmStatus getDualSolution(int i, double* yOpt) const
    {
        const int nd = nV_ + nC_ + 2 * nComp_;
        for (int k = 0; k < nd; ++k) yOpt[k] = y[(size_t)i * nd + k];
        return (AlgorithmStatus)st[i].status;
    }
    const lcqp_stats_t& getStats(int i) const { return st[i]; }
    int getNumberOfPrimals() const { return nV_; }
    int getNumberOfDuals() const { return nV_ + nC_ + 2 * nComp_; }
    lcqp_hip_batch_t* handle() { return h; }

  private:
    int B, nV_, nC_, nComp_;
    lcqp_hip_batch_t* h;
    std::vector<double> x, y;
    std::vector<lcqp_stats_t> st;
};

}  // namespace LCQPow
#endif
