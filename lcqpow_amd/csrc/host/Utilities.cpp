#include "Utilities.hpp"

#include <algorithm>
#include <cstdio>

namespace LCQPow {

void Utilities::MatrixMultiplication(const double* A, const double* B, double* C, int m, int n, int p)
{
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < p; ++c) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[(size_t)r * n + k] * B[(size_t)k * p + c];
            C[(size_t)r * p + c] = s;
        }
}

void Utilities::TransponsedMatrixMultiplication(const double* A, const double* B, double* C, int m, int n, int p)
{
    for (size_t e = 0; e < (size_t)n * p; ++e) C[e] = 0.0;
    AddTransponsedMatrixMultiplication(A, B, C, m, n, p);
}

void Utilities::AddTransponsedMatrixMultiplication(const double* A, const double* B, double* C, int m, int n, int p)
{
    // row sweep over A (unit stride) instead of the column-strided walk of the reference
    for (int k = 0; k < m; ++k)
        for (int r = 0; r < n; ++r) {
            const double a = A[(size_t)k * n + r];
            if (a == 0.0) continue;
            for (int c = 0; c < p; ++c) C[(size_t)r * p + c] += a * B[(size_t)k * p + c];
        }
}

void Utilities::MatrixSymmetrizationProduct(const double* A, const double* B, double* C, int m, int n)
{
    for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += A[(size_t)k * n + r] * B[(size_t)k * n + c] + B[(size_t)k * n + r] * A[(size_t)k * n + c];
            C[(size_t)r * n + c] = s;
            C[(size_t)c * n + r] = s;
        }
}

void Utilities::AffineLinearTransformation(double alpha, const double* A, const double* b, const double* c, double* d, int m, int n)
{
    for (int r = 0; r < m; ++r) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[(size_t)r * n + k] * b[k];
        d[r] = alpha * s + c[r];
    }
}

void Utilities::WeightedMatrixAdd(double alpha, const double* A, double beta, const double* B, double* C, int m, int n)
{
    for (size_t e = 0; e < (size_t)m * n; ++e) C[e] = alpha * A[e] + beta * B[e];
}

void Utilities::WeightedVectorAdd(double alpha, const double* a, double beta, const double* b, double* c, int m)
{
    WeightedMatrixAdd(alpha, a, beta, b, c, m, 1);
}

double Utilities::QuadraticFormProduct(const double* Q, const double* p, int m)
{
    double total = 0.0;
    for (int r = 0; r < m; ++r) {
        double s = 0.0;
        for (int c = 0; c < m; ++c) s += Q[(size_t)r * m + c] * p[c];
        total += s * p[r];
    }
    return total;
}

double Utilities::DotProduct(const double* a, const double* b, int m)
{
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += a[i] * b[i];
    return s;
}

double Utilities::MaxAbs(const double* a, int m)
{
    double best = 0.0;
    for (int i = 0; i < m; ++i) {
        const double v = std::fabs(a[i]);
        if (v > best) best = v;
    }
    return best;
}

ReturnValue Utilities::readFromFile(double* data, int n, const char* datafilename)
{
    // one value per line, row-major, "inf"/"-inf" accepted (what the reference's loader reads)
    FILE* f = datafilename ? std::fopen(datafilename, "r") : nullptr;
    if (!f) return UNABLE_TO_READ_FILE;
    for (int i = 0; i < n; ++i) {
        if (std::fscanf(f, "%lf", &data[i]) != 1) { std::fclose(f); return UNABLE_TO_READ_FILE; }
    }
    std::fclose(f);
    return SUCCESSFUL_RETURN;
}

}  // namespace LCQPow

// -------------------------------------------------------------------------------------------------
// CSC half.  Storage is malloc'ed like in the reference so that ClearSparseMat can free what createCSC
// was handed (src/Utilities.cpp:268-309,469-484).
// -------------------------------------------------------------------------------------------------
#include <cstdlib>
#include <cstring>
#include <vector>

namespace LCQPow {

csc* Utilities::createCSC(int m, int n, int nnx, double* x, int* i, int* p)
{
    csc* M = (csc*)std::malloc(sizeof(csc));
    if (!M) return 0;
    M->m = m; M->n = n; M->p = p; M->i = i; M->x = x; M->nz = -1; M->nzmax = nnx;
    return M;
}

csc* Utilities::copyCSC(int m, int n, int nnx, const double* x, const int* i, const int* p)
{
    int* rows = (int*)std::malloc(sizeof(int) * (size_t)(nnx > 0 ? nnx : 1));
    double* data = (double*)std::malloc(sizeof(double) * (size_t)(nnx > 0 ? nnx : 1));
    int* cols = (int*)std::malloc(sizeof(int) * (size_t)(n + 1));
    if (nnx > 0) { std::memcpy(rows, i, sizeof(int) * (size_t)nnx); std::memcpy(data, x, sizeof(double) * (size_t)nnx); }
    std::memcpy(cols, p, sizeof(int) * (size_t)(n + 1));
    return createCSC(m, n, nnx, data, rows, cols);
}

csc* Utilities::copyCSC(const csc* src, bool toUpperTriangular)
{
    if (!src) return 0;
    if (!toUpperTriangular) return copyCSC(src->m, src->n, src->p[src->n], src->x, src->i, src->p);
    std::vector<int> rows;
    std::vector<double> data;
    int* cols = (int*)std::malloc(sizeof(int) * (size_t)(src->n + 1));
    cols[0] = 0;
    for (int c = 0; c < src->n; ++c) {
        for (int k = src->p[c]; k < src->p[c + 1]; ++k)
            if (src->i[k] <= c) { rows.push_back(src->i[k]); data.push_back(src->x[k]); }   // on or above the diagonal
        cols[c + 1] = (int)rows.size();
    }
    csc* M = copyCSC(src->m, src->n, (int)rows.size(), data.data(), rows.data(), cols);
    std::free(cols);
    return M;
}

void Utilities::ClearSparseMat(csc** M)
{
    if (!M || !*M) return;
    std::free((*M)->p); std::free((*M)->i); std::free((*M)->x);
    std::free(*M);
    *M = 0;
}

double* Utilities::csc_to_dns(const csc* S)
{
    const int m = S->m, n = S->n;
    double* full = new double[(size_t)m * n]();
    for (int c = 0; c < n; ++c)
        for (int k = S->p[c]; k < S->p[c + 1]; ++k) {
            if (k == S->nzmax) return full;
            if (S->i[k] < 0 || S->i[k] >= m) { delete[] full; return 0; }   // INDEX_OUT_OF_BOUNDS
            full[(size_t)S->i[k] * n + c] = S->x[k];
        }
    return full;
}

csc* Utilities::dns_to_csc(const double* full, int m, int n)
{
    std::vector<int> rows;
    std::vector<double> data;
    std::vector<int> cols(n + 1, 0);
    for (int c = 0; c < n; ++c) {
        for (int r = 0; r < m; ++r) {
            const double v = full[(size_t)r * n + c];
            if (v > 0 || v < 0) { rows.push_back(r); data.push_back(v); }
        }
        cols[c + 1] = (int)rows.size();
    }
    return copyCSC(m, n, (int)rows.size(), data.data(), rows.data(), cols.data());
}

void Utilities::MatrixMultiplication(const csc* A, const double* b, double* c)
{
    for (int r = 0; r < A->m; ++r) c[r] = 0.0;
    for (int col = 0; col < A->n; ++col)
        for (int k = A->p[col]; k < A->p[col + 1]; ++k) c[A->i[k]] += A->x[k] * b[col];
}

void Utilities::TransponsedMatrixMultiplication(const csc* A, const double* b, double* c)
{
    for (int col = 0; col < A->n; ++col) c[col] = 0.0;
    AddTransponsedMatrixMultiplication(A, b, c);
}

void Utilities::AddTransponsedMatrixMultiplication(const csc* A, const double* b, double* c)
{
    for (int col = 0; col < A->n; ++col) {
        double s = 0.0;
        for (int k = A->p[col]; k < A->p[col + 1]; ++k) s += b[A->i[k]] * A->x[k];
        c[col] += s;
    }
}

csc* Utilities::MatrixSymmetrizationProduct(const csc* L, const csc* R)
{
    // C = L'R + R'L = sum over the rows k of the outer products L_k' R_k + R_k' L_k: work proportional to sum_k nnz(L_k) nnz(R_k), not to n^2
    // (round 6: the column-by-column version with a dense accumulator took n^2 steps -- minutes at nV = 16 384; the reference's own,
    // src/Utilities.cpp:118-173, is of that kind).  Row lists from the compressed columns, triplets, sorted by (column, row), duplicates summed.
    const int n = L->n, m = L->m;
    std::vector<int> lp(m + 1, 0), rp(m + 1, 0);
    for (int k = 0; k < L->p[n]; ++k) lp[L->i[k] + 1]++;
    for (int k = 0; k < R->p[n]; ++k) rp[R->i[k] + 1]++;
    for (int r = 0; r < m; ++r) { lp[r + 1] += lp[r]; rp[r + 1] += rp[r]; }
    std::vector<int> lc(L->p[n]), rc(R->p[n]);
    std::vector<double> lv(L->p[n]), rv(R->p[n]);
    { std::vector<int> cur(lp.begin(), lp.end() - 1); for (int j = 0; j < n; ++j) for (int k = L->p[j]; k < L->p[j + 1]; ++k) { const int d = cur[L->i[k]]++; lc[d] = j; lv[d] = L->x[k]; } }
    { std::vector<int> cur(rp.begin(), rp.end() - 1); for (int j = 0; j < n; ++j) for (int k = R->p[j]; k < R->p[j + 1]; ++k) { const int d = cur[R->i[k]]++; rc[d] = j; rv[d] = R->x[k]; } }
    struct T { int col, row; double v; };
    std::vector<T> t;
    for (int r = 0; r < m; ++r)
        for (int a = lp[r]; a < lp[r + 1]; ++a)
            for (int b = rp[r]; b < rp[r + 1]; ++b) {
                const double v = lv[a] * rv[b];
                t.push_back(T{rc[b], lc[a], v});      // (L'R)[lc][rc]
                t.push_back(T{lc[a], rc[b], v});      // (R'L)[rc][lc]
            }
    std::stable_sort(t.begin(), t.end(), [](const T& x, const T& y) { return x.col != y.col ? x.col < y.col : x.row < y.row; });
    std::vector<int> rows, cols(n + 1, 0);
    std::vector<double> data;
    for (size_t e = 0; e < t.size();) {
        size_t f = e; double s = 0.0;
        for (; f < t.size() && t[f].col == t[e].col && t[f].row == t[e].row; ++f) s += t[f].v;
        if (std::fabs(s) > ZERO) { rows.push_back(t[e].row); data.push_back(s); cols[t[e].col + 1]++; }
        e = f;
    }
    for (int j = 0; j < n; ++j) cols[j + 1] += cols[j];
    if (rows.empty()) return 0;
    return copyCSC(n, n, (int)rows.size(), data.data(), rows.data(), cols.data());
}

void Utilities::AffineLinearTransformation(double alpha, const csc* S, const double* b, const double* c, double* d, int m)
{
    // column sums, i.e. S'b: equal to S b for the symmetric matrices it is used with (as in the reference)
    for (int col = 0; col < m; ++col) {
        double s = 0.0;
        for (int k = S->p[col]; k < S->p[col + 1]; ++k) s += S->x[k] * b[S->i[k]];
        d[col] = alpha * s + c[col];
    }
}

double Utilities::QuadraticFormProduct(const csc* S, const double* p, int m)
{
    double total = 0.0;
    for (int col = 0; col < m; ++col) {
        double s = 0.0;
        for (int k = S->p[col]; k < S->p[col + 1]; ++k) s += S->x[k] * p[S->i[k]];
        total += p[col] * s;
    }
    return total;
}

}  // namespace LCQPow
