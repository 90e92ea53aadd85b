/*
 * lcqp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the LCQPow penalty-homotopy hot path used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The shipped path
 * (lcqpow_amd/csrc, liblcqpow_hip.so) never links or calls anything in this directory.
 *
 * What is pinned and what is not:
 *   - orc_util_* follow /root/reference/src/Utilities.cpp loop-for-loop and are pinned against the
 *     reference's own known-answer tests (test/RunUnitTests.cpp:33-246) in tests/test_oracle_kat.py.
 *   - orc_lcqp_* follows src/LCQProblem.cpp:444-560,563-626,726-785,880,885-1034,1105-1326,
 *     1353-1362,1381-1482 and is pinned by the reference's solver-level tests
 *     (test/RunUnitTests.cpp:463-551, test/examples/ *.cpp) and the printed optima of
 *     examples/OptimizeOnCircle.cpp:144-145.
 *   - orc_qp_* stands where the reference calls qpOASES (src/SubsolverQPOASES.cpp:134-181).
 *     qpOASES is an un-vendored submodule (external/qpOASES is empty, pinned commit unknown), so
 *     its pivoting cannot be restated: PARITY UNPINNED at that boundary.  orc_qp_* instead solves
 *     the same convex QP to a KKT-verified active-set solution (unique x for strictly convex QPs)
 *     with the qpOASES dual layout/sign, using the algorithm the HIP backend implements.
 *   - The reference itself cannot be compiled here: every translation unit includes <qpOASES.hpp>
 *     and <osqp.h> (src/Utilities.cpp:29-33), which this image lacks.  No oracle/_ref is built.
 */
#ifndef LCQP_ORACLE_H
#define LCQP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return values: include/Utilities.hpp:37-87 ---- */
enum {
    ORC_SUCCESSFUL_RETURN = 0,
    ORC_INVALID_OBJECTIVE_LINEAR_TERM = 116,
    ORC_INVALID_CONSTRAINT_MATRIX = 117,
    ORC_INVALID_COMPLEMENTARITY_MATRIX = 118,
    ORC_INVALID_LOWER_COMPLEMENTARITY_BOUND = 120,
    ORC_MAX_ITERATIONS_REACHED = 200,
    ORC_MAX_PENALTY_REACHED = 201,
    ORC_SUBPROBLEM_SOLVER_ERROR = 203,
    ORC_LCQPOBJECT_NOT_SETUP = 300
};

/* AlgorithmStatus: include/Utilities.hpp:103-109 */
enum { ORC_PROBLEM_NOT_SOLVED = 0, ORC_W_STATIONARY = 1, ORC_C_STATIONARY = 2, ORC_M_STATIONARY = 3, ORC_S_STATIONARY = 4 };

/* Options: defaults of src/Options.cpp:296-333 plus the subsolver knobs that replace qpOASES options.
 * Field order is identical to lcqp_options_t in include/lcqp_hip.h (Python maps both with one class). */
typedef struct {
    double complementarityTolerance; /* 1e3*EPS */
    double stationarityTolerance;    /* 1e6*EPS */
    double initialPenaltyParameter;  /* 0.01 */
    double penaltyUpdateFactor;      /* 2 */
    double maxPenaltyParameter;      /* 1e8 */
    double etaDynamicPenalty;        /* 0.9 */
    int    solveZeroPenaltyFirst;    /* 1 */
    int    perturbStep;              /* 1 */
    int    maxIterations;            /* 1000 */
    int    nDynamicPenalty;          /* 3 */
    int    printLevel;               /* 2 (ignored: the oracle never prints) */
    int    storeSteps;               /* 0 */
    uint64_t perturbSeed;            /* reference seeds rand() with time(NULL); here deterministic */
    /* --- QP subsolver (factor-once ADMM + active-set polish) --- */
    double admmRho;                  /* 0.1  (times max|diag Q|) */
    double admmSigma;                /* 1e-6 (times max|diag Q|) */
    double admmAlpha;                /* 1.6 */
    double rhoEqMult;                /* 1e3 */
    double proxSmall;                /* 1e-12 (times max|diag Q|): SPD Hessians */
    double proxBig;                  /* 1e-8  (times max|diag Q|): PSD Hessians */
    double pivotThreshold;           /* 1e-7: min Cholesky pivot / scale below which proxBig is used */
    double depTau;                   /* 1e-12: relative pivot below which an active row is dependent */
    double feasTol;                  /* 1e-9 */
    double resTol;                   /* 1e-12 */
    int    admmFirst;                /* 0: ADMM iterations before the first polish of an initial solve (ADMM is the fallback) */
    int    admmHot;                  /* 0: ADMM iterations before the first polish of a hot-started solve */
    int    maxTrials;                /* 12: active-set trials per polish */
    int    maxRounds;                /* 40: ADMM/polish rounds per QP */
} orc_options_t;

/* OutputStatistics counters: src/OutputStatistics.cpp:81-128, plus subsolver work counters used for
 * the algorithmic-byte accounting of bench.py. */
typedef struct {
    int    iterTotal, iterOuter, subproblemIter, status, qpSolverExitFlag, returnValue;
    double rhoOpt;
    int    admmIter, trials, factorizations, corrections, qpSolves, reserved;
} orc_stats_t;

void orc_options_default(orc_options_t* o);

/* ---- Utilities restated (src/Utilities.cpp) ---- */
void   orc_util_matmul(const double* A, const double* B, double* C, int m, int n, int p);           /* :38-47 */
void   orc_util_matmul_t(const double* A, const double* B, double* C, int m, int n, int p);         /* :62-72 */
void   orc_util_add_matmul_t(const double* A, const double* B, double* C, int m, int n, int p);     /* :85-93 */
void   orc_util_symm_product(const double* A, const double* B, double* C, int m, int n);            /* :104-116 */
void   orc_util_affine(double alpha, const double* A, const double* b, const double* c, double* d, int m, int n); /* :176-186 */
void   orc_util_weighted_matadd(double alpha, const double* A, double beta, const double* B, double* C, int m, int n); /* :202-206 */
void   orc_util_weighted_vecadd(double alpha, const double* a, double beta, const double* b, double* c, int m);        /* :209-211 */
double orc_util_quadform(const double* Q, const double* p, int m);                                  /* :214-225 */
double orc_util_dot(const double* a, const double* b, int m);                                       /* :244-250 */
double orc_util_maxabs(const double* a, int m);                                                     /* :253-265 */

/* ---- QP subsolver with the SubsolverBase semantics (include/SubsolverBase.hpp:37,52-56) ---- */
typedef struct orc_qp orc_qp_t;
orc_qp_t* orc_qp_create(int nV, int nC, const double* Q, const double* A, const orc_options_t* opt); /* SubsolverQPOASES.cpp:32-46 */
void      orc_qp_destroy(orc_qp_t* q);
int       orc_qp_solve(orc_qp_t* q, int initialSolve, int* iterations, int* exit_flag, const double* g,
                       const double* lbA, const double* ubA, const double* x0, const double* y0,
                       const double* lb, const double* ub);                                          /* :134-169 */
void      orc_qp_get_solution(orc_qp_t* q, double* x, double* y);                                    /* :172-181 */
void      orc_qp_get_counters(orc_qp_t* q, int* admm, int* trials, int* facts, int* corrections);
int       orc_qp_get_sweeps(orc_qp_t* q);   /* trials that swept Q and E (stats.reserved) */

/* ---- LCQP solve (LCQProblem::loadLCQP dense + runSolver) ----
 * NULL is allowed wherever the reference allows it (lbL,ubL,lbR,ubR,A (nC==0),lbA,ubA,lb,ub,x0,y0).
 * trace (optional, may be NULL): per pass of the loop, rows of [statk_inf, phi, rho, alphak, obj, merit, |pk|_inf, QP iterations]
 * and xk. */
int orc_lcqp_solve(int nV, int nC, int nComp,
                   const double* Q, const double* g, const double* L, const double* R,
                   const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                   const double* A, const double* lbA, const double* ubA,
                   const double* lb, const double* ub, const double* x0, const double* y0,
                   const orc_options_t* opt, double* xOpt, double* yOpt, orc_stats_t* stats,
                   int traceCap, double* traceScalars /* traceCap x 8 */, double* traceX /* traceCap x nV */, int* traceLen);

/* the dependent-row rules of orc_qp_* inside orc_lcqp_solve: on (1, the default) since every device kernel carries them (k_lcqp_run,
 * k_qp_solve); 0 switches them off for A/B runs of the oracle against itself */
void orc_lcqp_set_robust(int on);
/* 1 (default): the QP solver sums E x in the device's order (64 lanes + butterfly); 0: left to right.  See dot_lanes in lcqp_oracle.c. */
void orc_qp_set_sum_order(int device_order);
void orc_qp_set_trace(int on);          /* diagnostic: one stderr line per round / trial of orc_qp_solve (the device prints the same in a -DLCQP_TRACE_QP build) */
void orc_qp_set_enter_cap(int div);      /* cap on entering rows of a cold polish: max(n / div, 16) per trial; 0: off (test hook) */

/* ---- synthetic instances (include/lcqp_synth.h) and a threaded batch driver for the CPU baseline ---- */
void orc_synth_generate(uint64_t seed0, uint64_t instance, int n, int nC, int nComp,
                        double* Q, double* g, double* L, double* R, double* A, double* lbA, double* ubA);
/* solves instances [first, first+count) with `threads` worker threads (one LCQP per thread at a time);
 * outputs may be NULL. Returns the number of instances that returned SUCCESSFUL_RETURN. */
int orc_synth_batch_solve(uint64_t seed0, int first, int count, int n, int nC, int nComp, const orc_options_t* opt,
                          int threads, double* xOut, double* yOut, orc_stats_t* statsOut);
/* steady-state timing of the same solver: `threads` pinned workers (cpus[t], NULL = unpinned) with `perThread` pre-generated instances
 * each (instance ids first + t*perThread + k), buffers warm, common start; *secondsOut = start to last finish.  Returns the solved count. */
int orc_synth_bench(uint64_t seed0, int first, int threads, int perThread, const int* cpus, int n, int nC, int nComp, const orc_options_t* opt,
                    double* xOut, double* yOut, orc_stats_t* statsOut, double* secondsOut);
/* read bandwidth (GB/s) of the host memory under `threads` pinned streaming workers: the figure bench.py prints beside the cpu_baseline */
double orc_host_stream_gbps(int threads, const int* cpus, size_t bytesPerThread, int reps);


/* ---- sparse arm (lcqp_oracle_sparse.c): LCQProblem::runSolver with the OSQP_SPARSE conventions of src/LCQProblem.cpp:929-960
 * (no box constraints, nC + 2 nComp duals) over an OSQP-style subsolver (ADMM on the quasi-definite KKT matrix + active-set
 * polish).  Q: CSR of the full symmetric matrix; E = [A; L; R]: CSR, nC + 2 nComp rows; y0 / yOpt: nC + 2 nComp entries.
 * perm[nV + m]: ordering of the KKT matrix [Q E'; E .] (position -> node; node < nV: variable, else row node - nV); its first
 * nV + m - kb positions form a band of half bandwidth w, the last kb are border nodes (rows / variables too dense for a band) -- the
 * factorisations are bordered band LDL' (kb <= 64; kb = 0: plain band). */
int orc_sparse_lcqp_solve(int nV, int nC, int nComp,
                          const int* Qp, const int* Qi, const double* Qx, const double* g,
                          const int* Ep, const int* Ei, const double* Ex,
                          const double* lbA, const double* ubA, const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                          const double* x0, const double* y0, const int* perm, int w, int kb,
                          const orc_options_t* opt, double* xOpt, double* yOpt, orc_stats_t* stats);

#ifdef __cplusplus
}
#endif
#endif

/* ------------------------------------------------------------------------------------------------
 * CSC (compressed sparse column) half of Utilities, SURVEY.md §8(f-1).  orc_csc_t has the fields of the
 * OSQP `csc` struct the reference uses (src/Utilities.cpp:469-484): nzmax, m, n, p[n+1], i[nzmax], x[nzmax].
 * ---------------------------------------------------------------------------------------------- */
#ifndef ORC_CSC_DEFINED
#define ORC_CSC_DEFINED
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int nzmax, m, n; int* p; int* i; double* x; int nz; } orc_csc_t;
orc_csc_t* orc_csc_create(int m, int n, int nnz, const double* x, const int* i, const int* p);   /* copyCSC :487-513 (deep copy) */
void       orc_csc_free(orc_csc_t* M);                                                           /* ClearSparseMat :268-287 */
orc_csc_t* orc_csc_upper(const orc_csc_t* M);                                                    /* copyCSC(M, true) :516-560 */
double*    orc_csc_to_dns(const orc_csc_t* M);                                                   /* :593-617, caller frees */
orc_csc_t* orc_dns_to_csc(const double* full, int m, int n);                                     /* :620-650 */
void       orc_csc_matmul(const orc_csc_t* A, const double* b, double* c);                       /* MatrixMultiplication :49-59 */
void       orc_csc_matmul_t(const orc_csc_t* A, const double* b, double* c);                     /* TransponsedMatrixMultiplication :75-82 */
void       orc_csc_add_matmul_t(const orc_csc_t* A, const double* b, double* c);                 /* :96-102 */
orc_csc_t* orc_csc_symm_product(const orc_csc_t* L, const orc_csc_t* R);                         /* MatrixSymmetrizationProduct :118-173 */
void       orc_csc_affine(double alpha, const orc_csc_t* S, const double* b, const double* c, double* d, int m);  /* :189-199 (S'b) */
double     orc_csc_quadform(const orc_csc_t* S, const double* p, int m);                         /* :228-241 */
#ifdef __cplusplus
}
#endif
#endif
