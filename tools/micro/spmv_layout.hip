// Micro-benchmark for the sparse engine's products: 8 lanes per instance, lane per row, one pattern for the whole batch.
// (a) values in CSR order, gathered through a position map (what k_sparse_run does today)
// (b) values in sliced-ELL order (slice = the 8 rows of a lane group): every value load of a lane group is one 64-byte piece
// build: hipcc --offload-arch=gfx950 -O3 -o spmv_layout spmv_layout.hip ; run: ./spmv_layout [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
constexpr int G = 8, U = 4;
struct Pat { int rows, n, nnz; const int *ptr, *col; const int *sbase, *swidth, *scol; };
__global__ __launch_bounds__(64, 2) void k_csr(Pat p, const double* vals, const double* x, double* y, int B, int reps)
{
    const int inst = (blockIdx.x * 64 + threadIdx.x) / G, l = threadIdx.x % G;
    if (inst >= B) return;
    const double* v = vals + (size_t)inst * p.nnz; const double* xv = x + (size_t)inst * p.n; double* yo = y + (size_t)inst * p.rows;
    for (int rep = 0; rep < reps; rep++)
    for (int i0 = l; i0 < p.rows; i0 += U * G) {
        double s[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * G; s[u] = 0.0;
            if (i < p.rows) {
                const int a = p.ptr[i], b = p.ptr[i + 1];
                double acc[8]; int cc[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { const bool ok = a + q < b; cc[q] = ok ? p.col[a + q] : 0; acc[q] = ok ? v[a + q] : 0.0; }
#pragma unroll
                for (int q = 0; q < 8; q++) s[u] += acc[q] * xv[cc[q]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * G; if (i < p.rows) yo[i] = s[u] + rep; }
    }
}
__global__ __launch_bounds__(64, 2) void k_sliced(Pat p, const double* vals, const double* x, double* y, int B, int reps)
{
    const int inst = (blockIdx.x * 64 + threadIdx.x) / G, l = threadIdx.x % G;
    if (inst >= B) return;
    const double* v = vals + (size_t)inst * p.nnz; const double* xv = x + (size_t)inst * p.n; double* yo = y + (size_t)inst * p.rows;
    const int nt = (p.rows + G - 1) / G;
    for (int rep = 0; rep < reps; rep++)
    for (int t0 = 0; t0 < nt; t0 += U) {
        double s[U]; int base[U], wd[U]; int wmax = 0;
#pragma unroll
        for (int u = 0; u < U; u++) { const int t = t0 + u; const bool ok = t < nt; base[u] = ok ? p.sbase[t] : 0; wd[u] = ok ? p.swidth[t] : 0; wmax = max(wmax, wd[u]); s[u] = 0.0; }
#pragma unroll 2
        for (int q = 0; q < wmax; q++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool ok = q < wd[u];
                const int slot = base[u] + q * G + l;
                const int c = ok ? p.scol[slot] : 0;
                const double a = ok ? v[slot] : 0.0;
                s[u] += a * xv[c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = (t0 + u) * G + l; if (i < p.rows) yo[i] = s[u] + rep; }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 16384, reps = 20;
    const int rowsA = 2048, rowsL = 1024, rows = rowsA + rowsL, n = 4096;
    std::vector<int> ptr(rows + 1, 0), col;
    for (int r = 0; r < rows; r++) {
        if (r < rowsA) for (int q = 0; q < 6; q++) col.push_back(std::min(n - 1, 2 * r + q));
        else col.push_back((r - rowsA) * 4 + 1);
        ptr[r + 1] = (int)col.size();
    }
    const int nnz = (int)col.size(), nt = (rows + G - 1) / G;
    std::vector<int> sbase(nt), swidth(nt), scol;
    for (int t = 0; t < nt; t++) {
        int w = 0; for (int l = 0; l < G && t * G + l < rows; l++) w = std::max(w, ptr[t * G + l + 1] - ptr[t * G + l]);
        sbase[t] = (int)scol.size(); swidth[t] = w;
        for (int q = 0; q < w; q++) for (int l = 0; l < G; l++) { const int r = t * G + l; scol.push_back(r < rows && ptr[r] + q < ptr[r + 1] ? col[ptr[r] + q] : 0); }
    }
    const int nnzS = (int)scol.size();
    printf("B %d rows %d n %d nnz %d sliced slots %d\n", B, rows, n, nnz, nnzS);
    int *dptr, *dcol, *dsb, *dsw, *dsc; double *dv, *dx, *dy;
    CK(hipMalloc(&dptr, 4 * (rows + 1))); CK(hipMalloc(&dcol, 4 * nnz)); CK(hipMalloc(&dsb, 4 * nt)); CK(hipMalloc(&dsw, 4 * nt)); CK(hipMalloc(&dsc, 4 * nnzS));
    CK(hipMemcpy(dptr, ptr.data(), 4 * (rows + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(dcol, col.data(), 4 * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, sbase.data(), 4 * nt, hipMemcpyHostToDevice)); CK(hipMemcpy(dsw, swidth.data(), 4 * nt, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, scol.data(), 4 * nnzS, hipMemcpyHostToDevice));
    CK(hipMalloc(&dv, 8ull * B * nnzS)); CK(hipMalloc(&dx, 8ull * B * n)); CK(hipMalloc(&dy, 8ull * B * rows));
    CK(hipMemset(dv, 0, 8ull * B * nnzS)); CK(hipMemset(dx, 0, 8ull * B * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Pat p{rows, n, nnz, dptr, dcol, dsb, dsw, dsc};
    const int grid = (B * G + 63) / 64;
    for (int which = 0; which < 4; which++) {
        Pat q = p; if (which & 1) q.nnz = nnzS;
        CK(hipEventRecord(e0));
        if (which & 1) k_sliced<<<grid, 64>>>(q, dv, dx, dy, B, reps); else k_csr<<<grid, 64>>>(q, dv, dx, dy, B, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)B * reps * (8.0 * nnz + 8.0 * rows + 8.0 * n);
        printf("%s: %.2f ms for %d products -> %.3f ms per product of the batch, %.2f TB/s (values + x + y once)\n", which & 1 ? "sliced" : "csr   ", ms, reps, ms / reps, bytes / ms / 1e9);
    }
    return 0;
}
