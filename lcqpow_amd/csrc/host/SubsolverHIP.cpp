#include "SubsolverHIP.hpp"

namespace LCQPow {

SubsolverHIP::SubsolverHIP() : qp(nullptr) {}

SubsolverHIP::SubsolverHIP(int nV, int nC, const double* Q, const double* A, int device)
{
    qp = lcqp_hip_qp_create(nV, nC, Q, A, nullptr, device);
}

SubsolverHIP::SubsolverHIP(const SubsolverHIP& rhs) : qp(rhs.qp ? lcqp_hip_qp_clone(rhs.qp) : nullptr) {}

SubsolverHIP::~SubsolverHIP() { clear(); }

SubsolverHIP& SubsolverHIP::operator=(const SubsolverHIP& rhs)
{
    if (this != &rhs) {
        clear();
        qp = rhs.qp ? lcqp_hip_qp_clone(rhs.qp) : nullptr;
    }
    return *this;
}

void SubsolverHIP::clear()
{
    if (qp) lcqp_hip_qp_destroy(qp);
    qp = nullptr;
}

void SubsolverHIP::setOptions(const lcqp_options_t& options)
{
    if (qp) lcqp_hip_qp_set_options(qp, &options);
}

ReturnValue SubsolverHIP::solve(bool initialSolve, int& iterations, int& exit_flag, const double* const g,
                                const double* const lbA, const double* const ubA, const double* const x0,
                                const double* const y0, const double* const lb, const double* const ub)
{
    if (!qp) { iterations = 0; exit_flag = -1; return SUBPROBLEM_SOLVER_ERROR; }
    const int rc = lcqp_hip_qp_solve(qp, initialSolve ? 1 : 0, &iterations, &exit_flag, g, lbA, ubA, x0, y0, lb, ub);
    return rc == LCQP_SUCCESSFUL_RETURN ? SUCCESSFUL_RETURN : SUBPROBLEM_SOLVER_ERROR;
}

void SubsolverHIP::getSolution(double* x, double* y)
{
    if (qp) lcqp_hip_qp_get_solution(qp, x, y);
}

}  // namespace LCQPow
