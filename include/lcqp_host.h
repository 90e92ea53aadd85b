/*
 * lcqp_host.h -- C ABI over the single-problem host layer (LCQPow::LCQProblem / Options / OutputStatistics of
 * lcqpow_amd/csrc/host, library liblcqpow_host.so).  SURVEY.md §8(f-4): this is what a language binding binds.
 * The reference exposes the same three classes to Python through pybind11
 * (interfaces/python/lcqpow/LCQProblem.cpp:70-176, Options.cpp:12-43, OutputStatistics.cpp:14-31); here the
 * binding layer is plain C so ctypes / cgo / JNI can all use it, and lcqpow_amd/lcqpow.py is the Python mirror.
 *
 * Every call runs the QP subproblems on the GPU through liblcqpow_hip.so (include/lcqp_hip.h); there is no
 * CPU path behind this interface.  Matrices are dense row-major doubles (src/Utilities.cpp:43) or CSC triples
 * (m, n, nnz, x, i, p) with the field meaning of the reference's csc struct (src/Utilities.cpp:469-484).
 * NULL is allowed wherever the reference allows it.  Return values are LCQPow::ReturnValue codes
 * (include/Utilities.hpp:37-87) unless stated otherwise; nothing throws across this boundary.
 */
#ifndef LCQP_HOST_H
#define LCQP_HOST_H

#include "lcqp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Options (include/Options.hpp:30-221, setter validation src/Options.cpp:80-259) ---- */
typedef struct lcqp_host_options lcqp_host_options_t;

enum {                                        /* field ids for lcqp_host_options_set / _get */
    LCQP_OPT_STATIONARITY_TOLERANCE = 0,      /* src/Options.cpp:80-91 */
    LCQP_OPT_COMPLEMENTARITY_TOLERANCE = 1,   /* :94-105 */
    LCQP_OPT_INITIAL_PENALTY_PARAMETER = 2,   /* :108-119 */
    LCQP_OPT_PENALTY_UPDATE_FACTOR = 3,       /* :122-133 */
    LCQP_OPT_SOLVE_ZERO_PENALTY_FIRST = 4,    /* :136-144 */
    LCQP_OPT_PERTURB_STEP = 5,                /* :147-155 */
    LCQP_OPT_MAX_ITERATIONS = 6,              /* :158-169 */
    LCQP_OPT_MAX_PENALTY_PARAMETER = 7,       /* :172-182 */
    LCQP_OPT_N_DYNAMIC_PENALTY = 8,           /* :185-193 */
    LCQP_OPT_ETA_DYNAMIC_PENALTY = 9,         /* :196-207 */
    LCQP_OPT_PRINT_LEVEL = 10,                /* :210-228 */
    LCQP_OPT_STORE_STEPS = 11,                /* :231-239 */
    LCQP_OPT_QP_SOLVER = 12,                  /* :242-259 (range extended by HIP_DENSE = 3) */
    LCQP_OPT_PERTURB_SEED = 13                /* stand-in for srand(time(NULL)), src/LCQProblem.cpp:1016 */
};

lcqp_host_options_t* lcqp_host_options_create(void);                             /* Options::Options() = setToDefault */
lcqp_host_options_t* lcqp_host_options_copy(const lcqp_host_options_t* rhs);     /* Options(const Options&) */
void   lcqp_host_options_destroy(lcqp_host_options_t* o);
void   lcqp_host_options_set_to_default(lcqp_host_options_t* o);                 /* src/Options.cpp:296-333 */
int    lcqp_host_options_set(lcqp_host_options_t* o, int field, double value);   /* the setter's ReturnValue */
double lcqp_host_options_get(const lcqp_host_options_t* o, int field);
/* subsolver knobs (the role of the embedded qpOASES::Options, src/Options.cpp:262-271): whole struct in / out */
void   lcqp_host_options_get_hip(const lcqp_host_options_t* o, lcqp_options_t* out);
void   lcqp_host_options_set_hip(lcqp_host_options_t* o, const lcqp_options_t* in);

/* ---- LCQProblem (include/LCQProblem.hpp:38-242) ---- */
typedef struct lcqp_host_problem lcqp_host_problem_t;

lcqp_host_problem_t* lcqp_host_problem_create(int nV, int nC, int nComp);        /* LCQProblem(int,int,int) src/LCQProblem.cpp:43-84 */
void lcqp_host_problem_destroy(lcqp_host_problem_t* p);
void lcqp_host_problem_set_device(lcqp_host_problem_t* p, int device);           /* which GPU runs the subsolver (no reference analogue) */
/* HIP_DENSE runs the whole homotopy on the device (batch of one); hostLoop != 0 keeps the reference's host loop (src/LCQProblem.cpp:444-560)
 * over the SubsolverHIP plugin for it as well -- the reference's three solver values always use the host loop */
void lcqp_host_problem_set_host_loop(lcqp_host_problem_t* p, int hostLoop);
/* which engine the last runSolver used: 0 none yet, 1 host loop over the subsolver plugin, 2 whole homotopy on the device (dense kernels),
 * 3 the sparse engine (OSQP_SPARSE arm with a banded KKT pattern) */
int lcqp_host_problem_last_engine(const lcqp_host_problem_t* p);
void lcqp_host_problem_set_options(lcqp_host_problem_t* p, const lcqp_host_options_t* o);   /* include/LCQProblem.hpp:242 */

/* dense loadLCQP, src/LCQProblem.cpp:87-144 */
int lcqp_host_problem_load_dense(lcqp_host_problem_t* p, const double* Q, const double* g, const double* L, const double* R,
                                 const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                                 const double* A, const double* lbA, const double* ubA,
                                 const double* lb, const double* ub, const double* x0, const double* y0);

/* CSC matrix argument of the sparse loadLCQP; x == NULL means "matrix not given" (only allowed for A) */
typedef struct { int m, n, nnz; const double* x; const int* i; const int* p; } lcqp_csc_arg_t;

/* sparse loadLCQP, src/LCQProblem.cpp:390-441 (matrices are copied) */
int lcqp_host_problem_load_csc(lcqp_host_problem_t* p, const lcqp_csc_arg_t* Q, const double* g,
                               const lcqp_csc_arg_t* L, const lcqp_csc_arg_t* R,
                               const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                               const lcqp_csc_arg_t* A, const double* lbA, const double* ubA,
                               const double* lb, const double* ub, const double* x0, const double* y0);

/* file loadLCQP, src/LCQProblem.cpp:147-387: files[15] in the argument order Q g L R lbL ubL lbR ubR A lbA ubA lb ub x0 y0,
 * NULL entries for files that are not given */
int lcqp_host_problem_load_files(lcqp_host_problem_t* p, const char* const files[15]);

int lcqp_host_problem_switch_to_sparse(lcqp_host_problem_t* p);                  /* src/LCQProblem.cpp:1037-1068 */
int lcqp_host_problem_switch_to_dense(lcqp_host_problem_t* p);                   /* :1071-1102 */

int lcqp_host_problem_run(lcqp_host_problem_t* p);                               /* runSolver, src/LCQProblem.cpp:444-560 */
int lcqp_host_problem_number_of_primals(const lcqp_host_problem_t* p);           /* include/LCQProblem.hpp:221 */
int lcqp_host_problem_number_of_duals(const lcqp_host_problem_t* p);             /* :228 */
int lcqp_host_problem_get_primal(const lcqp_host_problem_t* p, double* xOpt);    /* returns AlgorithmStatus, src/LCQProblem.cpp:1485-1493 */
int lcqp_host_problem_get_dual(const lcqp_host_problem_t* p, double* yOpt);      /* returns AlgorithmStatus, :1496-1504 */

/* ---- OutputStatistics (include/OutputStatistics.hpp:31-227) ---- */
typedef struct {
    int    iterTotal, iterOuter, subproblemIter, status, qpSolverExitFlag, nSteps;   /* nSteps: length of the tracking vectors */
    double rhoOpt;
} lcqp_host_stats_t;

enum {                                        /* tracking vectors, filled when storeSteps is set (src/OutputStatistics.cpp:131-164) */
    LCQP_TRACK_INNER_ITERS = 0, LCQP_TRACK_SUBPROBLEM_ITERS = 1, LCQP_TRACK_ACCU_SUBPROBLEM_ITERS = 2,
    LCQP_TRACK_STEP_LENGTH = 3, LCQP_TRACK_STEP_SIZE = 4, LCQP_TRACK_STAT_VALS = 5, LCQP_TRACK_OBJ_VALS = 6,
    LCQP_TRACK_PHI_VALS = 7, LCQP_TRACK_MERIT_VALS = 8,
    LCQP_TRACK_X_STEPS = 9                    /* nSteps x nV, row per stored iterate */
};

void lcqp_host_problem_get_stats(const lcqp_host_problem_t* p, lcqp_host_stats_t* out);     /* getOutputStatistics, include/LCQProblem.hpp:235 */
/* copies min(cap, length) doubles of tracking vector `which` (integers are converted) and returns its full length */
int  lcqp_host_problem_get_track(const lcqp_host_problem_t* p, int which, double* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
