// C ABI over the host classes (include/lcqp_host.h): the binding layer of SURVEY.md §8(f-4).  Thin by design:
// every entry forwards to LCQProblem / Options / OutputStatistics, which in turn run the QP subproblems through
// liblcqpow_hip.so; nothing here computes.
#include <cstring>
#include <new>
#include <vector>

#include "LCQProblem.hpp"
#include "lcqp_host.h"

using namespace LCQPow;

struct lcqp_host_options { Options opt; };
struct lcqp_host_problem {
    LCQProblem lcqp;
    lcqp_host_problem(int nV, int nC, int nComp) : lcqp(nV, nC, nComp) {}
};

template <typename T>
static int copy_out(const std::vector<T>& v, double* out, int cap)
{
    int n = (int)v.size();
    if (out) for (int k = 0; k < n && k < cap; k++) out[k] = (double)v[k];
    return n;
}

extern "C" {

lcqp_host_options_t* lcqp_host_options_create(void) { return new (std::nothrow) lcqp_host_options(); }

lcqp_host_options_t* lcqp_host_options_copy(const lcqp_host_options_t* rhs)
try {
    if (!rhs) return nullptr;
    lcqp_host_options_t* o = new (std::nothrow) lcqp_host_options();
    if (o) o->opt = rhs->opt;
    return o;
}
catch (...) { return nullptr; }

void lcqp_host_options_destroy(lcqp_host_options_t* o) { delete o; }

void lcqp_host_options_set_to_default(lcqp_host_options_t* o) { if (o) o->opt.setToDefault(); }

int lcqp_host_options_set(lcqp_host_options_t* o, int field, double v)
{
    if (!o) return INVALID_ARGUMENT;
    Options& s = o->opt;
    switch (field) {
        case LCQP_OPT_STATIONARITY_TOLERANCE:    return s.setStationarityTolerance(v);
        case LCQP_OPT_COMPLEMENTARITY_TOLERANCE: return s.setComplementarityTolerance(v);
        case LCQP_OPT_INITIAL_PENALTY_PARAMETER: return s.setInitialPenaltyParameter(v);
        case LCQP_OPT_PENALTY_UPDATE_FACTOR:     return s.setPenaltyUpdateFactor(v);
        case LCQP_OPT_SOLVE_ZERO_PENALTY_FIRST:  return s.setSolveZeroPenaltyFirst(v != 0);
        case LCQP_OPT_PERTURB_STEP:              return s.setPerturbStep(v != 0);
        case LCQP_OPT_MAX_ITERATIONS:            return s.setMaxIterations((int)v);
        case LCQP_OPT_MAX_PENALTY_PARAMETER:     return s.setMaxPenaltyParameter(v);
        case LCQP_OPT_N_DYNAMIC_PENALTY:         return s.setNDynamicPenalty((int)v);
        case LCQP_OPT_ETA_DYNAMIC_PENALTY:       return s.setEtaDynamicPenalty(v);
        case LCQP_OPT_PRINT_LEVEL:               return s.setPrintLevel((int)v);
        case LCQP_OPT_STORE_STEPS:               return s.setStoreSteps(v != 0);
        case LCQP_OPT_QP_SOLVER:                 return s.setQPSolver((int)v);
        case LCQP_OPT_PERTURB_SEED:              s.setPerturbSeed((unsigned long long)v); return SUCCESSFUL_RETURN;
    }
    return INVALID_ARGUMENT;
}

double lcqp_host_options_get(const lcqp_host_options_t* o, int field)
{
    if (!o) return 0.0;
    const Options& s = o->opt;
    switch (field) {
        case LCQP_OPT_STATIONARITY_TOLERANCE:    return s.getStationarityTolerance();
        case LCQP_OPT_COMPLEMENTARITY_TOLERANCE: return s.getComplementarityTolerance();
        case LCQP_OPT_INITIAL_PENALTY_PARAMETER: return s.getInitialPenaltyParameter();
        case LCQP_OPT_PENALTY_UPDATE_FACTOR:     return s.getPenaltyUpdateFactor();
        case LCQP_OPT_SOLVE_ZERO_PENALTY_FIRST:  return s.getSolveZeroPenaltyFirst();
        case LCQP_OPT_PERTURB_STEP:              return s.getPerturbStep();
        case LCQP_OPT_MAX_ITERATIONS:            return s.getMaxIterations();
        case LCQP_OPT_MAX_PENALTY_PARAMETER:     return s.getMaxPenaltyParameter();
        case LCQP_OPT_N_DYNAMIC_PENALTY:         return s.getNDynamicPenalty();
        case LCQP_OPT_ETA_DYNAMIC_PENALTY:       return s.getEtaDynamicPenalty();
        case LCQP_OPT_PRINT_LEVEL:               return (int)s.getPrintLevel();
        case LCQP_OPT_STORE_STEPS:               return s.getStoreSteps();
        case LCQP_OPT_QP_SOLVER:                 return (int)s.getQPSolver();
        case LCQP_OPT_PERTURB_SEED:              return (double)s.getHIPOptions().perturbSeed;
    }
    return 0.0;
}

void lcqp_host_options_get_hip(const lcqp_host_options_t* o, lcqp_options_t* out)
{
    if (o && out) *out = o->opt.getHIPOptions();
}

void lcqp_host_options_set_hip(lcqp_host_options_t* o, const lcqp_options_t* in)
{
    if (o && in) o->opt.getHIPOptions() = *in;
}

lcqp_host_problem_t* lcqp_host_problem_create(int nV, int nC, int nComp)
try {
    return new (std::nothrow) lcqp_host_problem(nV, nC, nComp);
}
catch (...) { return nullptr; }   // nothing throws across the C boundary

void lcqp_host_problem_destroy(lcqp_host_problem_t* p) { delete p; }

void lcqp_host_problem_set_device(lcqp_host_problem_t* p, int device) { if (p) p->lcqp.setDevice(device); }
void lcqp_host_problem_set_host_loop(lcqp_host_problem_t* p, int hostLoop) { if (p) p->lcqp.setHostLoop(hostLoop != 0); }
int lcqp_host_problem_last_engine(const lcqp_host_problem_t* p) { return p ? p->lcqp.getLastEngine() : 0; }

void lcqp_host_problem_set_options(lcqp_host_problem_t* p, const lcqp_host_options_t* o)
{
    if (p && o) p->lcqp.setOptions(o->opt);
}

int lcqp_host_problem_load_dense(lcqp_host_problem_t* p, const double* Q, const double* g, const double* L, const double* R,
                                 const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                                 const double* A, const double* lbA, const double* ubA,
                                 const double* lb, const double* ub, const double* x0, const double* y0)
try {
    if (!p) return INVALID_ARGUMENT;
    return p->lcqp.loadLCQP(Q, g, L, R, lbL, ubL, lbR, ubR, A, lbA, ubA, lb, ub, x0, y0);
}
catch (...) { return INVALID_ARGUMENT; }

// borrowed view of a caller's CSC triple; loadLCQP(csc) deep-copies (src/LCQProblem.cpp:409-414)
static const csc* view(const lcqp_csc_arg_t* a, csc& store)
{
    if (!a || (!a->x && a->nnz > 0)) return nullptr;
    if (!a->p) return nullptr;
    store.m = a->m; store.n = a->n; store.nzmax = a->nnz; store.nz = -1;
    store.p = const_cast<int*>(a->p);
    store.i = const_cast<int*>(a->i);
    store.x = const_cast<double*>(a->x);
    return &store;
}

int lcqp_host_problem_load_csc(lcqp_host_problem_t* p, const lcqp_csc_arg_t* Q, const double* g,
                               const lcqp_csc_arg_t* L, const lcqp_csc_arg_t* R,
                               const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                               const lcqp_csc_arg_t* A, const double* lbA, const double* ubA,
                               const double* lb, const double* ub, const double* x0, const double* y0)
try {
    if (!p) return INVALID_ARGUMENT;
    csc sQ, sL, sR, sA;
    return p->lcqp.loadLCQP(view(Q, sQ), g, view(L, sL), view(R, sR), lbL, ubL, lbR, ubR, view(A, sA), lbA, ubA, lb, ub, x0, y0);
}
catch (...) { return INVALID_ARGUMENT; }

int lcqp_host_problem_load_files(lcqp_host_problem_t* p, const char* const f[15])
try {
    if (!p || !f) return INVALID_ARGUMENT;
    return p->lcqp.loadLCQP(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9], f[10], f[11], f[12], f[13], f[14]);
}
catch (...) { return UNABLE_TO_READ_FILE; }

int lcqp_host_problem_switch_to_sparse(lcqp_host_problem_t* p)
try { return p ? p->lcqp.switchToSparseMode() : INVALID_ARGUMENT; }
catch (...) { return FAILED_SWITCH_TO_SPARSE; }
int lcqp_host_problem_switch_to_dense(lcqp_host_problem_t* p)
try { return p ? p->lcqp.switchToDenseMode() : INVALID_ARGUMENT; }
catch (...) { return FAILED_SWITCH_TO_DENSE; }

int lcqp_host_problem_run(lcqp_host_problem_t* p)
try { return p ? p->lcqp.runSolver() : INVALID_ARGUMENT; }
catch (...) { return SUBPROBLEM_SOLVER_ERROR; }   // host memory exhausted inside the solver
int lcqp_host_problem_number_of_primals(const lcqp_host_problem_t* p) { return p ? p->lcqp.getNumberOfPrimals() : 0; }
int lcqp_host_problem_number_of_duals(const lcqp_host_problem_t* p) { return p ? p->lcqp.getNumberOfDuals() : 0; }
int lcqp_host_problem_get_primal(const lcqp_host_problem_t* p, double* xOpt) { return p ? p->lcqp.getPrimalSolution(xOpt) : PROBLEM_NOT_SOLVED; }
int lcqp_host_problem_get_dual(const lcqp_host_problem_t* p, double* yOpt) { return p ? p->lcqp.getDualSolution(yOpt) : PROBLEM_NOT_SOLVED; }

void lcqp_host_problem_get_stats(const lcqp_host_problem_t* p, lcqp_host_stats_t* out)
try {
    if (!p || !out) return;
    OutputStatistics s;
    p->lcqp.getOutputStatistics(s);
    out->iterTotal = s.getIterTotal();
    out->iterOuter = s.getIterOuter();
    out->subproblemIter = s.getSubproblemIter();
    out->status = (int)s.getSolutionStatus();
    out->qpSolverExitFlag = s.getQPSolverExitFlag();
    out->nSteps = (int)s.getInnerItersStdVec().size();
    out->rhoOpt = s.getRhoOpt();
}
catch (...) { }

int lcqp_host_problem_get_track(const lcqp_host_problem_t* p, int which, double* out, int cap)
try {
    if (!p) return 0;
    OutputStatistics s;
    p->lcqp.getOutputStatistics(s);
    switch (which) {
        case LCQP_TRACK_INNER_ITERS:           return copy_out(s.getInnerItersStdVec(), out, cap);
        case LCQP_TRACK_SUBPROBLEM_ITERS:      return copy_out(s.getSubproblemItersStdVec(), out, cap);
        case LCQP_TRACK_ACCU_SUBPROBLEM_ITERS: return copy_out(s.getAccuSubproblemItersStdVec(), out, cap);
        case LCQP_TRACK_STEP_LENGTH:           return copy_out(s.getStepLengthStdVec(), out, cap);
        case LCQP_TRACK_STEP_SIZE:             return copy_out(s.getStepSizeStdVec(), out, cap);
        case LCQP_TRACK_STAT_VALS:             return copy_out(s.getStatValsStdVec(), out, cap);
        case LCQP_TRACK_OBJ_VALS:              return copy_out(s.getObjValsStdVec(), out, cap);
        case LCQP_TRACK_PHI_VALS:              return copy_out(s.getPhiValsStdVec(), out, cap);
        case LCQP_TRACK_MERIT_VALS:            return copy_out(s.getMeritValsStdVec(), out, cap);
        case LCQP_TRACK_X_STEPS: {
            const std::vector<std::vector<double>>& xs = s.getxStepsStdVec();
            int total = 0;
            for (const std::vector<double>& row : xs)
                for (double v : row) { if (out && total < cap) out[total] = v; total++; }
            return total;
        }
    }
    return 0;
}
catch (...) { return 0; }

}  // extern "C"
