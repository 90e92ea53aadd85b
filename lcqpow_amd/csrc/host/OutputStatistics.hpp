// OutputStatistics: counters and optional per-iterate tracking vectors with the reference's accessor
// names (include/OutputStatistics.hpp, src/OutputStatistics.cpp:81-164).
#ifndef LCQPOW_AMD_OUTPUTSTATISTICS_HPP
#define LCQPOW_AMD_OUTPUTSTATISTICS_HPP

#include <vector>

#include "Utilities.hpp"

namespace LCQPow {

class OutputStatistics {
  public:
    OutputStatistics() { reset(); }
    void reset();
    ReturnValue updateIterTotal(int delta);
    ReturnValue updateIterOuter(int delta);
    ReturnValue updateSubproblemIter(int delta);
    ReturnValue updateRhoOpt(double rho);
    ReturnValue updateSolutionStatus(AlgorithmStatus s) { status = s; return SUCCESSFUL_RETURN; }
    ReturnValue updateQPSolverExitFlag(int flag) { qpSolverExitFlag = flag; return SUCCESSFUL_RETURN; }
    ReturnValue updateTrackingVectors(const double* xStep, int innerIters, int subproblemIters, double stepLength,
                                      double stepSize, double statVal, double objVal, double phiVal, double meritVal, int nV);

    int getIterTotal() const { return iterTotal; }
    int getIterOuter() const { return iterOuter; }
    int getSubproblemIter() const { return subproblemIter; }
    double getRhoOpt() const { return rhoOpt; }
    AlgorithmStatus getSolutionStatus() const { return status; }
    int getQPSolverExitFlag() const { return qpSolverExitFlag; }
    const std::vector<std::vector<double>>& getxStepsStdVec() const { return xSteps; }
    const std::vector<int>& getInnerItersStdVec() const { return innerIters; }
    const std::vector<int>& getSubproblemItersStdVec() const { return subproblemIters; }
    const std::vector<int>& getAccuSubproblemItersStdVec() const { return accuSubproblemIters; }
    const std::vector<double>& getStepLengthStdVec() const { return stepLength; }
    const std::vector<double>& getStepSizeStdVec() const { return stepSize; }
    const std::vector<double>& getStatValsStdVec() const { return statVals; }
    const std::vector<double>& getObjValsStdVec() const { return objVals; }
    const std::vector<double>& getPhiValsStdVec() const { return phiVals; }
    const std::vector<double>& getMeritValsStdVec() const { return meritVals; }

  private:
    int iterTotal, iterOuter, subproblemIter, qpSolverExitFlag;
    double rhoOpt;
    AlgorithmStatus status;
    std::vector<std::vector<double>> xSteps;
    std::vector<int> innerIters, subproblemIters, accuSubproblemIters;
    std::vector<double> stepLength, stepSize, statVals, objVals, phiVals, meritVals;
};

}  // namespace LCQPow
#endif
