// LCQProblem: the reference's public surface (include/LCQProblem.hpp:56-242) on the HIP backend.
//   LCQProblem(nV,nC,nComp) / loadLCQP (dense arrays or files) / runSolver / getPrimalSolution /
//   getDualSolution / getNumberOfPrimals / getNumberOfDuals / getOutputStatistics / setOptions.
// runSolver drives the penalty homotopy on the host exactly like the reference does and calls the
// subsolver through the Subsolver dispatcher (one solve site, one getSolution site:
// src/LCQProblem.cpp:1118,1138).  For throughput over many LCQPs use BatchLCQProblem, which runs the whole
// loop on the device.
#ifndef LCQPOW_AMD_LCQPROBLEM_HPP
#define LCQPOW_AMD_LCQPROBLEM_HPP

#include <deque>
#include <vector>

#include "Options.hpp"
#include "OutputStatistics.hpp"
#include "Subsolver.hpp"

namespace LCQPow {

class LCQProblem {
  public:
    LCQProblem();
    LCQProblem(int nV, int nC, int nComp);

    ReturnValue loadLCQP(const double* const Q, const double* const g, const double* const L, const double* const R,
                         const double* const lbL = 0, const double* const ubL = 0, const double* const lbR = 0,
                         const double* const ubR = 0, const double* const A = 0, const double* const lbA = 0,
                         const double* const ubA = 0, const double* const lb = 0, const double* const ub = 0,
                         const double* const x0 = 0, const double* const y0 = 0);
    ReturnValue loadLCQP(const char* const Q_file, const char* const g_file, const char* const L_file,
                         const char* const R_file, const char* const lbL_file = 0, const char* const ubL_file = 0,
                         const char* const lbR_file = 0, const char* const ubR_file = 0, const char* const A_file = 0,
                         const char* const lbA_file = 0, const char* const ubA_file = 0, const char* const lb_file = 0,
                         const char* const ub_file = 0, const char* const x0_file = 0, const char* const y0_file = 0);
    // sparse overload of loadLCQP (src/LCQProblem.cpp:390-441): CSC matrices, copied
    ReturnValue loadLCQP(const csc* const Q, const double* const g, const csc* const L, const csc* const R,
                         const double* const lbL = 0, const double* const ubL = 0, const double* const lbR = 0,
                         const double* const ubR = 0, const csc* const A = 0, const double* const lbA = 0,
                         const double* const ubA = 0, const double* const lb = 0, const double* const ub = 0,
                         const double* const x0 = 0, const double* const y0 = 0);
    // src/LCQProblem.cpp:1037-1102: convert the stored problem between dense arrays and CSC
    ReturnValue switchToSparseMode();
    ReturnValue switchToDenseMode();
    ~LCQProblem();
    LCQProblem(const LCQProblem&) = delete;
    LCQProblem& operator=(const LCQProblem&) = delete;
    ReturnValue runSolver();
    AlgorithmStatus getPrimalSolution(double* const xOpt) const;
    AlgorithmStatus getDualSolution(double* const yOpt) const;
    int getNumberOfPrimals() const { return nV; }
    int getNumberOfDuals() const { return nDuals; }
    void getOutputStatistics(OutputStatistics& stats_) const { stats_ = stats; }
    void setOptions(const Options& options_) { options = options_; }
    void setDevice(int device_) { device = device_; }
    // HIP_DENSE runs the whole homotopy on the device (a batch of one through lcqp_hip_batch_*); the reference's three solver values keep
    // the reference's host loop over the subsolver plugin.  hostLoop = true forces the host loop for HIP_DENSE too.
    void setHostLoop(bool hostLoop_) { hostLoop = hostLoop_; }
    // which engine the last runSolver used: 0 none yet, 1 the reference's host loop over the subsolver plugin, 2 the whole homotopy on the
    // device (dense kernels, k_lcqp_run), 3 the sparse engine (k_sparse_sched)
    enum Engine { ENGINE_NONE = 0, ENGINE_HOST_LOOP = 1, ENGINE_DENSE_DEVICE = 2, ENGINE_SPARSE_DEVICE = 3 };
    int getLastEngine() const { return lastEngine; }

  private:
    ReturnValue initializeSolver(bool needSubsolver = true);
    ReturnValue loadVectors(const double* g, const double* lbL, const double* ubL, const double* lbR, const double* ubR, const double* lbA,
                            const double* ubA, const double* lb, const double* ub, const double* x0, const double* y0);      // bounds, guess: shared by the dense and the CSC loader
    ReturnValue runOnDevice();                 // HIP_DENSE: LCQProblem::runSolver as one batch-of-one launch of k_lcqp_run
    bool runSparseOnDevice(ReturnValue& ret);  // OSQP_SPARSE with a banded or bordered pattern: k_sparse_sched; false when the pattern is not banded
    void finishFromTrace(const std::vector<double>& sc, const std::vector<double>& xs, int len);
    ReturnValue solveQPSubproblem(bool initialSolve);
    void updateLinearization();
    void updateStationarity();
    void updatePenalty();
    void getOptimalStepLength();
    bool leyfferCheckPositive();
    void perturbStep();
    void transformDuals();
    void determineStationarityType();
    void storeSteps();
    void printIteration();
    double getPhi();
    double getObj();
    double getMerit();

    // products with the stored matrices in whichever mode the problem is held (dense arrays or CSC)
    void mulQ(const double* v, double* out) const;      // out = Q v
    void mulC(const double* v, double* out) const;      // out = C v
    void mulAT(const double* y, double* out) const;     // out = [A;L;R]' y
    void mulL(const double* v, double* out) const;      // out = L v
    void mulR(const double* v, double* out) const;      // out = R v
    void addLT(const double* y, double* out) const;     // out += L' y
    void addRT(const double* y, double* out) const;     // out += R' y
    void clearSparse();

    int nV, nC, nComp, nDuals, boxDualOffset, device;
    bool loaded, haveYk, haveLbL, haveLbR, haveBox, sparseSolver, hostLoop;
    int lastEngine = ENGINE_NONE;
    std::vector<double> y0Full, ysub;   // y0 as loaded (reference layout nV + nC + 2 nComp); subsolver dual vector (box duals first)
    csc *Q_sparse, *A_sparse, *L_sparse, *R_sparse, *C_sparse;
    std::vector<double> Q, g, L, R, A, lbA, ubA, lb, ub, lbL, lbR, C, Qk;
    std::vector<double> gTilde, gPhi, xk, yk, ykA, gk, xnew, pk, statk, constrStatk, lkTmp, Qx, Cx, Qp, Cp;
    double phiConst, alphak, rho;
    int outerIter, innerIter, totalIter, qpIterk, qpSolverExitFlag;
    unsigned long long perturbCounter;
    AlgorithmStatus algoStat;
    std::deque<double> complHistory;
    Options options;
    OutputStatistics stats;
    Subsolver subsolver;
};

}  // namespace LCQPow
#endif
