// Options with the 13 algorithm options, names, defaults and setter validation of the reference
// (include/Options.hpp, src/Options.cpp:85-259,296-333).  The embedded qpOASES::Options / OSQPSettings of
// the reference are replaced by the knobs of the HIP subsolver (lcqp_options_t tail fields).
#ifndef LCQPOW_AMD_OPTIONS_HPP
#define LCQPOW_AMD_OPTIONS_HPP

#include "Utilities.hpp"
#include "lcqp_hip.h"

namespace LCQPow {

class Options {
  public:
    Options() { setToDefault(); }
    void setToDefault();

    double getStationarityTolerance() const { return o.stationarityTolerance; }
    ReturnValue setStationarityTolerance(double val);
    double getComplementarityTolerance() const { return o.complementarityTolerance; }
    ReturnValue setComplementarityTolerance(double val);
    double getInitialPenaltyParameter() const { return o.initialPenaltyParameter; }
    ReturnValue setInitialPenaltyParameter(double val);
    double getPenaltyUpdateFactor() const { return o.penaltyUpdateFactor; }
    ReturnValue setPenaltyUpdateFactor(double val);
    bool getSolveZeroPenaltyFirst() const { return o.solveZeroPenaltyFirst != 0; }
    ReturnValue setSolveZeroPenaltyFirst(bool val) { o.solveZeroPenaltyFirst = val; return SUCCESSFUL_RETURN; }
    bool getPerturbStep() const { return o.perturbStep != 0; }
    ReturnValue setPerturbStep(bool val) { o.perturbStep = val; return SUCCESSFUL_RETURN; }
    int getMaxIterations() const { return o.maxIterations; }
    ReturnValue setMaxIterations(int val);
    double getMaxPenaltyParameter() const { return o.maxPenaltyParameter; }
    ReturnValue setMaxPenaltyParameter(double val);
    int getNDynamicPenalty() const { return o.nDynamicPenalty; }
    ReturnValue setNDynamicPenalty(int val) { o.nDynamicPenalty = val; return SUCCESSFUL_RETURN; }
    double getEtaDynamicPenalty() const { return o.etaDynamicPenalty; }
    ReturnValue setEtaDynamicPenalty(double val);
    PrintLevel getPrintLevel() const { return (PrintLevel)o.printLevel; }
    ReturnValue setPrintLevel(PrintLevel val) { o.printLevel = (int)val; return SUCCESSFUL_RETURN; }
    ReturnValue setPrintLevel(int val);
    bool getStoreSteps() const { return o.storeSteps != 0; }
    ReturnValue setStoreSteps(bool val) { o.storeSteps = val; return SUCCESSFUL_RETURN; }
    QPSolver getQPSolver() const { return qpSolver; }
    ReturnValue setQPSolver(QPSolver val) { qpSolver = val; return SUCCESSFUL_RETURN; }
    ReturnValue setQPSolver(int val);
    // deterministic stand-in for srand(time(NULL)) (src/LCQProblem.cpp:1016)
    void setPerturbSeed(unsigned long long seed) { o.perturbSeed = seed; }

    // subsolver options (the role qpOASES::Options plays in the reference, src/Options.cpp:262-271)
    lcqp_options_t& getHIPOptions() { return o; }
    const lcqp_options_t& getHIPOptions() const { return o; }

  private:
    lcqp_options_t o;
    QPSolver qpSolver;
};

}  // namespace LCQPow
#endif
