#include "Options.hpp"

namespace LCQPow {

void Options::setToDefault()
{
    lcqp_hip_options_default(&o);   // src/Options.cpp:296-333 defaults live behind the C ABI
    qpSolver = HIP_DENSE;           // the only backend this build carries
}

ReturnValue Options::setStationarityTolerance(double val)
{
    if (val <= Utilities::EPS) return INVALID_STATIONARITY_TOLERANCE;
    o.stationarityTolerance = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setComplementarityTolerance(double val)
{
    if (val <= Utilities::EPS) return INVALID_COMPLEMENTARITY_TOLERANCE;
    o.complementarityTolerance = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setInitialPenaltyParameter(double val)
{
    if (val <= Utilities::ZERO) return INVALID_INITIAL_PENALTY_VALUE;
    o.initialPenaltyParameter = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setPenaltyUpdateFactor(double val)
{
    if (val <= 1) return INVALID_PENALTY_UPDATE_VALUE;
    o.penaltyUpdateFactor = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setMaxIterations(int val)
{
    if (val <= 0) return INVALID_MAX_ITERATIONS_VALUE;
    o.maxIterations = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setMaxPenaltyParameter(double val)
{
    if (val <= 0) return INVALID_MAX_RHO_VALUE;
    o.maxPenaltyParameter = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setEtaDynamicPenalty(double val)
{
    if (val <= Utilities::EPS || val >= 1) return INVALID_ETA_VALUE;
    o.etaDynamicPenalty = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setPrintLevel(int val)
{
    if (val < NONE || val > INNER_LOOP_ITERATES) return INVALID_PRINT_LEVEL_VALUE;
    o.printLevel = val;
    return SUCCESSFUL_RETURN;
}
ReturnValue Options::setQPSolver(int val)
{
    if (val < QPOASES_DENSE || val > HIP_DENSE) return INVALID_QPSOLVER;   // range check extended by one
    qpSolver = (QPSolver)val;
    return SUCCESSFUL_RETURN;
}

}  // namespace LCQPow
