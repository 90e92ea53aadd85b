#include "Subsolver.hpp"

namespace LCQPow {

Subsolver::Subsolver() : qpSolver(HIP_DENSE) {}

Subsolver::Subsolver(int nV, int nC, const double* Q, const double* A, QPSolver _qpSolver, int device) : qpSolver(_qpSolver)
{
    if (qpSolver == HIP_DENSE) {
        SubsolverHIP tmp(nV, nC, Q, A, device);
        solverHIP = tmp;
    }
}

Subsolver::Subsolver(const Subsolver& rhs) { copy(rhs); }

Subsolver& Subsolver::operator=(const Subsolver& rhs)
{
    if (this != &rhs) copy(rhs);
    return *this;
}

void Subsolver::copy(const Subsolver& rhs)
{
    qpSolver = rhs.qpSolver;
    if (qpSolver == HIP_DENSE) solverHIP = rhs.solverHIP;
}

void Subsolver::getSolution(double* x, double* y)
{
    if (qpSolver == HIP_DENSE) solverHIP.getSolution(x, y);
}

ReturnValue Subsolver::solve(bool initialSolve, int& iterations, int& exit_flag, const double* g, const double* lbA,
                             const double* ubA, const double* x0, const double* y0, const double* lb, const double* ub)
{
    if (qpSolver == HIP_DENSE) return solverHIP.solve(initialSolve, iterations, exit_flag, g, lbA, ubA, x0, y0, lb, ub);
    return INVALID_QPSOLVER;
}

void Subsolver::setOptions(const lcqp_options_t& options)
{
    if (qpSolver == HIP_DENSE) solverHIP.setOptions(options);
}

}  // namespace LCQPow
