#include "Utilities.hpp"

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
