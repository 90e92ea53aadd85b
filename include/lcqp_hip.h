/*
 * lcqp_hip.h -- C ABI of the MI355X-native LCQPow hot path (liblcqpow_hip.so).
 *
 * This is the drop-in boundary of SURVEY.md §8(b): plain C types, caller-owned buffers, integer
 * return codes, no exceptions.  Each entry point cites the reference interface it replaces
 * (file:line under the LCQPow tree).  A reference-side binding (a SubsolverHIP class derived from
 * SubsolverBase) is shown in INTEGRATION.md.
 *
 * Two groups:
 *   lcqp_hip_qp_*     one convex QP object with the SubsolverBase semantics
 *                     (include/SubsolverBase.hpp:37,52-56; src/SubsolverQPOASES.cpp:32-46,134-181).
 *   lcqp_hip_batch_*  B independent LCQPs solved by the penalty homotopy entirely on the device
 *                     (LCQProblem::loadLCQP + runSolver, src/LCQProblem.cpp:87-144,444-560), one
 *                     persistent workgroup per instance.  The reference has no batched API; this is
 *                     the throughput path that BASELINE.json's metric is measured on.
 *
 * All matrices are dense row-major doubles exactly as the reference takes them
 * (src/Utilities.cpp:43).  Infinite bounds are IEEE +-INFINITY (src/LCQProblem.cpp:596,607).
 */
#ifndef LCQP_HIP_H
#define LCQP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ReturnValue subset used across the boundary: include/Utilities.hpp:37-87 */
#define LCQP_SUCCESSFUL_RETURN 0
#define LCQP_INVALID_ARGUMENT 100
#define LCQP_INVALID_OBJECTIVE_LINEAR_TERM 116
#define LCQP_INVALID_CONSTRAINT_MATRIX 117
#define LCQP_INVALID_COMPLEMENTARITY_MATRIX 118
#define LCQP_INVALID_LOWER_COMPLEMENTARITY_BOUND 120
#define LCQP_MAX_ITERATIONS_REACHED 200
#define LCQP_MAX_PENALTY_REACHED 201
#define LCQP_SUBPROBLEM_SOLVER_ERROR 203
#define LCQP_LCQPOBJECT_NOT_SETUP 300
/* backend errors (new) */
#define LCQP_HIP_ERROR 900          /* a HIP runtime call failed; see lcqp_hip_last_error() */
#define LCQP_HIP_UNSUPPORTED 901    /* dimensions outside what the kernels are built for */

/* Options: the 13 algorithm options of include/Options.hpp with the defaults of
 * src/Options.cpp:296-333, followed by the subsolver knobs that take the place of
 * qpOASES::Options (src/Options.cpp:320-321). */
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
    int    printLevel;               /* 2; the device path never prints */
    int    storeSteps;               /* 0 */
    uint64_t perturbSeed;            /* deterministic stand-in for srand(time(NULL)), src/LCQProblem.cpp:1016 */
    double admmRho, admmSigma, admmAlpha, rhoEqMult;
    double proxSmall, proxBig, pivotThreshold, depTau, feasTol, resTol;
    int    admmFirst, admmHot, maxTrials, maxRounds;
} lcqp_options_t;

/* OutputStatistics counters (src/OutputStatistics.cpp:81-128) + work counters of the subsolver. */
typedef struct {
    int    iterTotal, iterOuter, subproblemIter, status, qpSolverExitFlag, returnValue;
    double rhoOpt;
    int    admmIter, trials, factorizations, corrections, qpSolves, reserved;
} lcqp_stats_t;

void lcqp_hip_options_default(lcqp_options_t* opt);               /* Options::setToDefault, src/Options.cpp:296 */
const char* lcqp_hip_last_error(void);
int  lcqp_hip_device_count(void);
/* Ask the HIP runtime for n hardware queues (sets GPU_MAX_HW_QUEUES unless the environment already holds a value).  Opt-in, for the
 * PROGRAM to call before the first HIP call of the process -- the runtime reads the variable once; the library never changes the
 * environment by itself.  A BatchPipeline with more batch objects alive than its own wants this.
 * Returns 0 (set), 1 (already set by the caller's environment: left alone) or LCQP_HIP_ERROR (n outside 1..64). */
int  lcqp_hip_request_hw_queues(int n);

/* ------------------------------------------------------------------------------------------------
 * QP object == SubsolverBase implementation state.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lcqp_hip_qp lcqp_hip_qp_t;

/* SubsolverQPOASES(int nV,int nC,double* Q,double* A): src/SubsolverQPOASES.cpp:32-46 (deep copy of Q, A).
 * nC is the number of stacked rows (nC + 2*nComp).  Host pointers.  Returns NULL on failure. */
lcqp_hip_qp_t* lcqp_hip_qp_create(int nV, int nC, const double* Q, const double* A,
                                  const lcqp_options_t* opt, int device);
/* copy-ctor / operator= of the reference (src/SubsolverQPOASES.cpp:184-230): the clone carries the host copies
 * of Q, A and the options; its device state is built by its own first (initial) solve -- the reference only
 * copies subsolvers before their first use (src/Subsolver.cpp:125-136, src/LCQProblem.cpp:906-907). */
lcqp_hip_qp_t* lcqp_hip_qp_clone(const lcqp_hip_qp_t* src);
void lcqp_hip_qp_destroy(lcqp_hip_qp_t* qp);
/* SubsolverQPOASES::setOptions, src/SubsolverQPOASES.cpp:120-131 (takes effect at the next initial solve) */
int  lcqp_hip_qp_set_options(lcqp_hip_qp_t* qp, const lcqp_options_t* opt);
/* SubsolverBase::solve, include/SubsolverBase.hpp:52-56 / src/SubsolverQPOASES.cpp:134-169.
 * Returns LCQP_SUCCESSFUL_RETURN or LCQP_SUBPROBLEM_SOLVER_ERROR; *exit_flag != 0 on failure (the reference stores
 * qpOASES' raw status there, src/LCQProblem.cpp:1122, and its tests only require "non-zero on failure"):
 *   1 the subsolver gave up after maxRounds rounds, 2 lbA > ubA or lb > ub, 3 the setup factorisations failed,
 *   4 certified primal infeasible, 5 certified unbounded (OSQP's certificates on the ADMM iterates), -1 HIP error.
 * lbA/ubA/lb/ub/x0/y0 may be NULL as in the reference. */
int  lcqp_hip_qp_solve(lcqp_hip_qp_t* qp, int initialSolve, int* iterations, int* exit_flag,
                       const double* g, const double* lbA, const double* ubA,
                       const double* x0, const double* y0, const double* lb, const double* ub);
/* SubsolverBase::getSolution, include/SubsolverBase.hpp:37 / src/SubsolverQPOASES.cpp:172-181:
 * x[nV], y[nV + nC]: box duals first, then one dual per stacked row; Qx + g - A'y_A - y_box = 0. */
void lcqp_hip_qp_get_solution(lcqp_hip_qp_t* qp, double* x, double* y);
void lcqp_hip_qp_get_counters(lcqp_hip_qp_t* qp, int* admm, int* trials, int* factorizations, int* corrections);

/* ------------------------------------------------------------------------------------------------
 * Batch of B independent dense LCQPs of one shape (nV, nC, nComp).
 * ---------------------------------------------------------------------------------------------- */
typedef struct lcqp_hip_batch lcqp_hip_batch_t;

/* LCQProblem(nV,nC,nComp) x B: src/LCQProblem.cpp:43-79.  withBox != 0 reserves rows for lb/ub. */
lcqp_hip_batch_t* lcqp_hip_batch_create(int batch, int nV, int nC, int nComp, int withBox, int device);
void lcqp_hip_batch_destroy(lcqp_hip_batch_t* b);
/* LCQProblem::setOptions, include/LCQProblem.ipp:160-163 */
int  lcqp_hip_batch_set_options(lcqp_hip_batch_t* b, const lcqp_options_t* opt);
/* LCQProblem::loadLCQP (dense), src/LCQProblem.cpp:87-144, for instances [first, first+count):
 * arrays are host pointers, packed instance after instance ([count][...]); NULL as in the reference. */
int  lcqp_hip_batch_load(lcqp_hip_batch_t* b, int first, int count,
                         const double* Q, const double* g, const double* L, const double* R,
                         const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                         const double* A, const double* lbA, const double* ubA,
                         const double* lb, const double* ub, const double* x0, const double* y0);
/* Fill instances [0,B) with the synthetic generator of include/lcqp_synth.h directly in HBM
 * (instance id = firstInstance + b). */
int  lcqp_hip_batch_generate_synthetic(lcqp_hip_batch_t* b, uint64_t seed0, uint64_t firstInstance);
/* read one instance's problem data back (any pointer may be NULL) -- used by parity tests / cpu_baseline */
int  lcqp_hip_batch_read_problem(lcqp_hip_batch_t* b, int instance, double* Q, double* g, double* L, double* R,
                                 double* A, double* lbA, double* ubA);
/* Constant-matrix setup: C = L'R + R'L (src/LCQProblem.cpp:622-623), phi expressions (:969-996) and the
 * two factorisations the subsolver reuses across every iterate (replaces qp.init's setup,
 * src/SubsolverQPOASES.cpp:152).  Asynchronous on the batch stream. */
int  lcqp_hip_batch_setup(lcqp_hip_batch_t* b);
/* LCQProblem::runSolver for all instances, src/LCQProblem.cpp:444-560.  Asynchronous on the batch
 * stream; includes setup if it has not run since the last load. */
int  lcqp_hip_batch_run(lcqp_hip_batch_t* b);
/* Hint of a caller that keeps several batch objects in flight (BatchPipeline): the setup of this object will run beside the homotopy
 * kernel of another one.  The library then launches the setup kernels that fit into the registers and LDS ONE finished instance frees on
 * a compute unit (the streamed form of Et = E L1^-T instead of the register-resident one), so that the setup starts in the gaps of the
 * other launch instead of behind it (+10 % through a two-deep pipeline at B = 1024, profiles/round6/setup/pipelined_variants.log).  The
 * results are the same bits either way; the default (0) is the faster setup of a batch that runs alone. */
int  lcqp_hip_batch_set_overlapped(lcqp_hip_batch_t* b, int overlapped);
int  lcqp_hip_batch_synchronize(lcqp_hip_batch_t* b);
/* time of the last run measured with HIP events on the batch stream, ms: setup_ms = the setup kernels, solve_ms = the homotopy launch */
int  lcqp_hip_batch_last_timing(lcqp_hip_batch_t* b, float* setup_ms, float* solve_ms);
/* getPrimalSolution / getDualSolution / getOutputStatistics, src/LCQProblem.cpp:1485-1504,1519:
 * x[B][nV], y[B][nV+nC+2nComp], stats[B]; returnValue of runSolver is stats[i].returnValue. */
int  lcqp_hip_batch_get_solution(lcqp_hip_batch_t* b, double* x, double* y, lcqp_stats_t* stats);
/* per-iterate tracking of one instance when options.storeSteps != 0 (LCQProblem::storeSteps,
 * src/LCQProblem.cpp:1365-1378; OutputStatistics tracking vectors, src/OutputStatistics.cpp:131-164): scalars[len][8] =
 * (|statk|_inf, phi, rho, alphak, objective, merit, |pk|_inf, iterations of the last QP), x[len][nV] = xk at the top of
 * every pass of the loop; at most cap rows are copied. */
int  lcqp_hip_batch_get_trace(lcqp_hip_batch_t* b, int instance, int cap, double* scalars, double* x, int* len);
/* per-instance cycle counters of the homotopy kernel's phases, out[B][16]; all zero unless the library was
 * built with -DLCQP_PROFILE (tools/gpu.py phase_profile) */
int  lcqp_hip_batch_read_profile(lcqp_hip_batch_t* b, unsigned long long* out);
/* raw HIP stream (hipStream_t) the batch launches on, for event timing by the caller */
void* lcqp_hip_batch_stream(lcqp_hip_batch_t* b);
/* algorithmic HBM bytes of the last run, from the work counters the kernels keep */
double lcqp_hip_batch_algorithmic_bytes(lcqp_hip_batch_t* b);
/* the work sums that enter it, summed over the batch (counted by the kernel): out[0] = rows of Et read by the corrections (twice the
 * rows of the factor for a full correction, once plus the rows that left for a predicted one), out[1] = entries of the inverse factor read by
 * the corrections (its triangle, n_T (n_T + 1) / 2 per pass; full rows beyond 256 slots), out[2] = bytes moved by the working-set updates (and the entries of M a predicted correction
 * reads), out[3] = number of working-set updates, out[4] = rows of E read by the residual sweeps (stage 1: unscreened inactive rows;
 * stage 2: active rows), out[5] = triangular solves with L1 (two per full correction, one per predicted correction) */
int    lcqp_hip_batch_work_sums(lcqp_hip_batch_t* b, double out[6]);

/* ------------------------------------------------------------------------------------------------
 * Building blocks exposed for parity tests and micro-benchmarks (each is one kernel launch over a
 * batch of independent instances; host pointers, synchronous).
 * ---------------------------------------------------------------------------------------------- */
/* Utilities::AffineLinearTransformation for symmetric A, src/Utilities.cpp:176-186: d = alpha*A*b + c */
int lcqp_hip_util_symv(int batch, int n, double alpha, const double* A, const double* b, const double* c, double* d);
/* Utilities::TransponsedMatrixMultiplication with p = 1, src/Utilities.cpp:62-72: c = A' * b (A is m x n) */
int lcqp_hip_util_gemv_t(int batch, int m, int n, const double* A, const double* b, double* c);
/* Utilities::MatrixMultiplication with p = 1, src/Utilities.cpp:38-47: c = A * b */
int lcqp_hip_util_gemv(int batch, int m, int n, const double* A, const double* b, double* c);
/* the row sweep of the subsolver's trials on its own (wg_rows through a row list, scalars indexed by row): for the nlist rows r = list[b][k]
 * of every instance, dots[b][r] = A_r . x (other entries of dots stay) and outT[b] = sum_k coef[b][r] A_r; x, coef, dots, outT may be NULL */
int lcqp_hip_util_rows_list(int batch, int m, int n, const double* A, const int* list, int nlist, const double* x, const double* coef,
                            double* dots, double* outT);
/* Utilities::MatrixSymmetrizationProduct, src/Utilities.cpp:104-116: C = A'B + B'A (A, B are m x n) */
int lcqp_hip_util_symm_product(int batch, int m, int n, const double* A, const double* B, double* C);
/* ---- CSC utilities on the device (host pointers, synchronous) ----
 * Handle on one CSC matrix (fields of the reference's `csc`, src/Utilities.cpp:469-484) and its transpose. */
typedef struct lcqp_hip_csc lcqp_hip_csc_t;
lcqp_hip_csc_t* lcqp_hip_csc_create(int m, int n, int nnz, const int* p, const int* i, const double* x, int device);
void lcqp_hip_csc_destroy(lcqp_hip_csc_t* M);
/* d = alpha * op(A) b + c (c may be NULL).  transposed == 0: Utilities::MatrixMultiplication(csc) src/Utilities.cpp:49-59;
 * transposed != 0: TransponsedMatrixMultiplication(csc) :75-82 and, for symmetric S, AffineLinearTransformation(csc)
 * :189-199 (QuadraticFormProduct :228-241 is b'd).  repeat > 1 re-launches for timing, *ms = time per launch. */
int lcqp_hip_csc_apply(lcqp_hip_csc_t* M, int transposed, double alpha, const double* b, const double* c, double* d,
                       int repeat, float* ms);
/* micro-benchmark of the row sweep on device-resident random data (batch matrices of m x n):
 * mode 1 = A x (dots), 2 = A'y (axpy), 3 = both in one sweep; *ms = time per launch */
int lcqp_hip_bench_rows(int batch, int m, int n, int mode, int repeat, float* ms);
/* Cholesky factorisation + nrhs back-solves of an SPD n x n matrix: x = K^-1 b (the factor-once /
 * back-solve-many kernel pair).  repeat > 1 re-runs the back-solve for timing; *ms gets the
 * per-back-solve kernel time. */
int lcqp_hip_chol_solve(int batch, int n, const double* K, const double* b, double* x, int repeat, float* ms);


/* ------------------------------------------------------------------------------------------------
 * Batch of B independent SPARSE LCQPs that share one sparsity pattern: the OSQP_SPARSE arm of the reference
 * (src/LCQProblem.cpp:929-960: no box constraints, nC + 2 nComp duals) on the device -- ADMM on the quasi-definite KKT
 * matrix + active-set polish (the role of SubsolverOSQP, src/SubsolverOSQP.cpp:124-200), CSR/CSC products, band LDL' in a
 * reverse Cuthill-McKee ordering computed here once per pattern.  Pattern arrays are the CSC arrays the reference holds
 * (Q_sparse and the stacked A_sparse = [A; L; R], src/LCQProblem.cpp:629-723; Q full symmetric).  Three factorisation engines, chosen
 * here per pattern: a band of half width <= 63; such a band plus at most 16 dense border nodes (rows or variables that touch many
 * others: the arrow of examples/OptimizeOnCircle.cpp); and, for any other pattern -- as the reference's OSQP arm takes any
 * (src/SubsolverOSQP.cpp:136-152) --, a general sparse LDL' (nested dissection with dense fronts, one wavefront per instance:
 * lcqp_sparse_general.hpp; a 2-D grid with 16 384 variables is one).  Returns NULL only when a front of that factorisation would exceed
 * 576 rows or the factor 2^28 entries (lcqp_hip_sparse_last_error() says so): the host layer runs such a problem on the dense kernels,
 * which take nV <= 4096.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lcqp_hip_sparse lcqp_hip_sparse_t;
lcqp_hip_sparse_t* lcqp_hip_sparse_create(int batch, int nV, int nC, int nComp, const int* Qp, const int* Qi,
                                          const int* Ap, const int* Ai, int device);
void lcqp_hip_sparse_destroy(lcqp_hip_sparse_t* s);
const char* lcqp_hip_sparse_last_error(void);
/* diagnostic (-DLCQP_SCHED_PROFILE builds, zeros otherwise): per phase of the scheduler (start, round, trial, factor, correct, QP end, idle polls)
 * clock ticks, wavefront steps, instances served: 21 values */
int  lcqp_hip_sparse_sched_profile(lcqp_hip_sparse_t* s, unsigned long long* out21);
int  lcqp_hip_sparse_bandwidth(const lcqp_hip_sparse_t* s);              /* half bandwidth of the KKT band */
int  lcqp_hip_sparse_lanes(const lcqp_hip_sparse_t* s);                  /* lanes of a wavefront per instance: 8, 16, 32 or 64 */
int  lcqp_hip_sparse_fronts(const lcqp_hip_sparse_t* s);                 /* fronts of the general sparse LDL' (a pattern that is neither banded nor bordered: nested dissection, dense fronts, a wavefront per instance); 0: a band engine */
int  lcqp_hip_sparse_border(const lcqp_hip_sparse_t* s);                 /* border nodes of the bordered band: the last positions of the ordering (0: plain band) */
/* perm[nV + nC + 2 nComp]: position -> node.  Two orderings of the band are prepared (the second, for Hessians that are safely definite by
 * their diagonals, puts every row behind one of its variables); this is the one the instances loaded so far select (before any load: the first) */
int  lcqp_hip_sparse_get_ordering(const lcqp_hip_sparse_t* s, int* perm);
int  lcqp_hip_sparse_set_options(lcqp_hip_sparse_t* s, const lcqp_options_t* opt);
/* loadLCQP, sparse overload (src/LCQProblem.cpp:390-441), values only: Qx [count][nnzQ], Ax [count][nnzA] in the CSC order of
 * the pattern; y0 [count][nC + 2 nComp]; NULL as in the reference */
int  lcqp_hip_sparse_load(lcqp_hip_sparse_t* s, int first, int count, const double* Qx, const double* g, const double* Ax,
                          const double* lbA, const double* ubA, const double* lbL, const double* ubL, const double* lbR,
                          const double* ubR, const double* x0, const double* y0);
int  lcqp_hip_sparse_run(lcqp_hip_sparse_t* s);                          /* runSolver for every instance (asynchronous) */
int  lcqp_hip_sparse_synchronize(lcqp_hip_sparse_t* s);
int  lcqp_hip_sparse_last_timing(lcqp_hip_sparse_t* s, float* setup_ms, float* solve_ms);
int  lcqp_hip_sparse_get_solution(lcqp_hip_sparse_t* s, double* x, double* y, lcqp_stats_t* stats);   /* y: [B][nC + 2 nComp] */
double lcqp_hip_sparse_algorithmic_bytes(lcqp_hip_sparse_t* s);
/* per-iterate trace of one instance of the last run (options.storeSteps), same layout as lcqp_hip_batch_get_trace
 * (src/LCQProblem.cpp:1365-1378, 1528-1576: the host LCQProblem rebuilds the tracking vectors and the iteration table from it) */
int  lcqp_hip_sparse_get_trace(lcqp_hip_sparse_t* s, int instance, int cap, double* scalars, double* x, int* len);
/* mean clock ticks per instance in 8 phases of the last run; LCQP_HIP_UNSUPPORTED unless the library was built with -DLCQP_PROFILE */
int  lcqp_hip_sparse_read_profile(lcqp_hip_sparse_t* s, double* out);

#ifdef __cplusplus
}
#endif
#endif /* LCQP_HIP_H */
