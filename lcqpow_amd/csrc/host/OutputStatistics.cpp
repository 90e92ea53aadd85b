#include "OutputStatistics.hpp"

namespace LCQPow {

void OutputStatistics::reset()
{
    iterTotal = iterOuter = subproblemIter = qpSolverExitFlag = 0;
    rhoOpt = 0.0;
    status = PROBLEM_NOT_SOLVED;
    xSteps.clear(); innerIters.clear(); subproblemIters.clear(); accuSubproblemIters.clear();
    stepLength.clear(); stepSize.clear(); statVals.clear(); objVals.clear(); phiVals.clear(); meritVals.clear();
}

ReturnValue OutputStatistics::updateIterTotal(int delta)
{
    if (delta <= 0) return INVALID_TOTAL_ITER_COUNT;
    iterTotal += delta;
    return SUCCESSFUL_RETURN;
}
ReturnValue OutputStatistics::updateIterOuter(int delta)
{
    if (delta <= 0) return INVALID_TOTAL_OUTER_ITER;
    iterOuter += delta;
    return SUCCESSFUL_RETURN;
}
ReturnValue OutputStatistics::updateSubproblemIter(int delta)
{
    if (delta < 0) return IVALID_SUBPROBLEM_ITER;
    subproblemIter += delta;
    return SUCCESSFUL_RETURN;
}
ReturnValue OutputStatistics::updateRhoOpt(double rho)
{
    if (rho <= 0) return INVALID_RHO_OPT;
    rhoOpt = rho;
    return SUCCESSFUL_RETURN;
}
ReturnValue OutputStatistics::updateTrackingVectors(const double* xStep, int inner, int subIters, double sLen, double sSize,
                                                    double statVal, double objVal, double phiVal, double meritVal, int nV)
{
    xSteps.emplace_back(xStep, xStep + nV);
    innerIters.push_back(inner);
    subproblemIters.push_back(subIters);
    accuSubproblemIters.push_back(subIters + (accuSubproblemIters.empty() ? 0 : accuSubproblemIters.back()));
    stepLength.push_back(sLen);
    stepSize.push_back(sSize);
    statVals.push_back(statVal);
    objVals.push_back(objVal);
    phiVals.push_back(phiVal);
    meritVals.push_back(meritVal);
    return SUCCESSFUL_RETURN;
}

}  // namespace LCQPow
