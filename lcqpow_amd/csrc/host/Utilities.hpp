// Host-side enums, constants and dense helpers with the names and argument meaning of the reference's
// Utilities (include/Utilities.hpp:37-129,140-328,350-362; src/Utilities.cpp:38-265).  Written from
// scratch for this backend: std::vector-free, row-major, no qpOASES/OSQP types.  The sparse (csc) half of the
// reference's Utilities (src/Utilities.cpp:49-59,75-82,118-173,189-199,228-241,593-650) is declared below as well.
#ifndef LCQPOW_AMD_UTILITIES_HPP
#define LCQPOW_AMD_UTILITIES_HPP

#include <cmath>
#include <cstddef>

namespace LCQPow {

enum ReturnValue {
    NOT_YET_IMPLEMENTED = -1,
    SUCCESSFUL_RETURN = 0,
    INVALID_ARGUMENT = 100,
    INVALID_PENALTY_UPDATE_VALUE = 101,
    INVALID_COMPLEMENTARITY_TOLERANCE = 102,
    INVALID_INITIAL_PENALTY_VALUE = 103,
    INVALID_MAX_ITERATIONS_VALUE = 104,
    INVALID_STATIONARITY_TOLERANCE = 105,
    INVALID_NUMBER_OF_OPTIM_VARS = 106,
    INVALID_NUMBER_OF_COMP_VARS = 107,
    INVALID_NUMBER_OF_CONSTRAINT_VARS = 108,
    INVALID_QPSOLVER = 109,
    INVALID_OSQP_BOX_CONSTRAINTS = 110,
    INVALID_TOTAL_ITER_COUNT = 111,
    INVALID_TOTAL_OUTER_ITER = 112,
    IVALID_SUBPROBLEM_ITER = 113,
    INVALID_RHO_OPT = 114,
    INVALID_PRINT_LEVEL_VALUE = 115,
    INVALID_OBJECTIVE_LINEAR_TERM = 116,
    INVALID_CONSTRAINT_MATRIX = 117,
    INVALID_COMPLEMENTARITY_MATRIX = 118,
    INVALID_ETA_VALUE = 119,
    INVALID_LOWER_COMPLEMENTARITY_BOUND = 120,
    INVALID_MAX_RHO_VALUE = 121,
    MAX_ITERATIONS_REACHED = 200,
    MAX_PENALTY_REACHED = 201,
    INITIAL_SUBPROBLEM_FAILED = 202,
    SUBPROBLEM_SOLVER_ERROR = 203,
    FAILED_SYM_COMPLEMENTARITY_MATRIX = 204,
    FAILED_SWITCH_TO_SPARSE = 205,
    FAILED_SWITCH_TO_DENSE = 206,
    LCQPOBJECT_NOT_SETUP = 300,
    INDEX_OUT_OF_BOUNDS = 301,
    UNABLE_TO_READ_FILE = 302,
    DENSE_SPARSE_MISSMATCH = 402
};

enum AlgorithmStatus {
    PROBLEM_NOT_SOLVED = 0,
    W_STATIONARY_SOLUTION = 1,
    C_STATIONARY_SOLUTION = 2,
    M_STATIONARY_SOLUTION = 3,
    S_STATIONARY_SOLUTION = 4
};

enum PrintLevel { NONE = 0, OUTER_LOOP_ITERATES = 1, INNER_LOOP_ITERATES = 2 };

// include/Utilities.hpp:125-129 plus the new backend (SURVEY.md §8b touch-point 1)
enum QPSolver { QPOASES_DENSE = 0, QPOASES_SPARSE = 1, OSQP_SPARSE = 2, HIP_DENSE = 3 };

// Compressed sparse column matrix with the fields of the `csc` struct the reference takes from <osqp.h>
// (used at src/Utilities.cpp:469-484): column pointers p[n+1], row indices i[nzmax], values x[nzmax].
struct csc {
    int nzmax, m, n;
    int* p;
    int* i;
    double* x;
    int nz;   // -1: compressed-column form
};

class Utilities {
  public:
    static constexpr double EPS = 2.221e-16;
    static constexpr double ZERO = 1.0e-25;
    static constexpr double INFTY = 1.0e20;

    template <typename P> static bool isNullPtr(P p) { return p == nullptr; }
    template <typename P> static bool isNotNullPtr(P p) { return p != nullptr; }

    // C(m x p) = A(m x n) * B(n x p)
    static void MatrixMultiplication(const double* A, const double* B, double* C, int m, int n, int p);
    // C(n x p) = A(m x n)' * B(m x p)
    static void TransponsedMatrixMultiplication(const double* A, const double* B, double* C, int m, int n, int p);
    // C(n x p) += A(m x n)' * B(m x p)
    static void AddTransponsedMatrixMultiplication(const double* A, const double* B, double* C, int m, int n, int p);
    // C(n x n) = A'B + B'A, A and B are m x n
    static void MatrixSymmetrizationProduct(const double* A, const double* B, double* C, int m, int n);
    // d = alpha*A*b + c, A is m x n
    static void AffineLinearTransformation(double alpha, const double* A, const double* b, const double* c, double* d, int m, int n);
    static void WeightedMatrixAdd(double alpha, const double* A, double beta, const double* B, double* C, int m, int n);
    static void WeightedVectorAdd(double alpha, const double* a, double beta, const double* b, double* c, int m);
    static double QuadraticFormProduct(const double* Q, const double* p, int m);
    static double DotProduct(const double* a, const double* b, int m);
    static double MaxAbs(const double* a, int m);
    static ReturnValue readFromFile(double* data, int n, const char* datafilename);   // src/Utilities.cpp:341-366

    // ---- CSC half (src/Utilities.cpp:49-59,75-82,96-102,118-173,189-199,228-241,268-309,469-650) ----
    static csc* createCSC(int m, int n, int nnx, double* x, int* i, int* p);          // takes ownership of x, i, p
    static csc* copyCSC(int m, int n, int nnx, const double* x, const int* i, const int* p);
    static csc* copyCSC(const csc* M, bool toUpperTriangular = false);
    static void ClearSparseMat(csc** M);
    static double* csc_to_dns(const csc* sparse);                                      // new[]; caller delete[]s
    static csc* dns_to_csc(const double* full, int m, int n);
    static void MatrixMultiplication(const csc* A, const double* b, double* c);       // c = A b
    static void TransponsedMatrixMultiplication(const csc* A, const double* b, double* c);      // c = A' b
    static void AddTransponsedMatrixMultiplication(const csc* A, const double* b, double* c);   // c += A' b
    static csc* MatrixSymmetrizationProduct(const csc* L, const csc* R);               // L'R + R'L, 0 when empty
    static void AffineLinearTransformation(double alpha, const csc* S, const double* b, const double* c, double* d, int m);
    static double QuadraticFormProduct(const csc* S, const double* p, int m);
};

}  // namespace LCQPow
#endif
