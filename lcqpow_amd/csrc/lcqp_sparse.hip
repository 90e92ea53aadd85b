// lcqp_sparse.hip -- the SPARSE arm of the hot path on gfx950: B independent LCQPs that share one sparsity pattern, one persistent
// wavefront (64-thread workgroup) per instance (k_sparse_run), behind lcqp_hip_sparse_* (include/lcqp_hip.h).
//
// Restates, with the conventions of the reference's OSQP_SPARSE arm (src/LCQProblem.cpp:929-960: no box constraints, nC + 2 nComp
// duals, no box term in the stationarity :1246-1272, dual sign of src/SubsolverOSQP.cpp:196-199):
//   * runSolver (src/LCQProblem.cpp:444-560) with the CSC Utilities products (src/Utilities.cpp:49-59,75-82,189-199) done as CSR
//     row gathers (Q x, E x) and CSC column gathers (E'y) -- deterministic, no atomics; C x = L'(R x) + R'(L x), C never formed;
//   * the subsolver behind SubsolverOSQP (src/SubsolverOSQP.cpp:124-200; OSQP itself is an absent submodule -- its published
//     algorithm is what is built): ADMM on the quasi-definite KKT matrix [Q + sigma I, E'; E, -1/rho], factorised ONCE per LCQP,
//     and an active-set polish on [Q + delta I, Ea'; Ea, -delta2 I] in iterative-refinement form, refactorised only when the
//     working set changes.  oracle/lcqp_oracle_sparse.c is the same algorithm in scalar C.
// The KKT matrices are factorised as BAND matrices in a reverse Cuthill-McKee ordering computed once per pattern on the host
// (all instances of a batch share it): LDL' with a sliding (w+1) x (w+1) window in LDS (half bandwidth w <= 63), triangular
// solves by one wave that keeps the 64 pending rows in its lanes (axpy form both ways, no reductions in the chain).
// Patterns whose KKT band is wider (e.g. the arrow-shaped circle example) are refused here; the host layer runs them on the
// dense kernels behind the same OSQP_SPARSE surface.
#include "lcqp_wg.hpp"
#include "../../include/lcqp_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <queue>
#include <string>
#include <vector>

using namespace lcqp;

namespace {

constexpr int SP_WMAX = 63;
constexpr int WGS = 64;      // ONE wave per instance: the factorisation and the triangular solves are chains a single wave runs; many
                             // instances per CU (up to 32 waves) hide their latencies instead of three idle partner waves
enum { NV_G, NV_GTIL, NV_GPHI, NV_XK, NV_PK, NV_XNEW, NV_GK, NV_QX, NV_CX, NV_QP, NV_CP, NV_TMP, NV_XQ, NV_XA, NV_XT, NV_R1, NV_R1S, NV_GS,
       NV_X0, NV_NUM };
enum { MV_L, MV_U, MV_RHOV, MV_YQ, MV_YA, MV_ZA, MV_YT, MV_EX, MV_EXS, MV_YK, MV_Y0, MV_LX, MV_NUM };
enum { MI_ST, MI_STT, MI_STF, MI_NEW, MI_NUM };

struct SpInfo {
    int haveSolution, stfValid, hasY0, pad;
    double scale, sigma, delta, delta2, phiConst;
    double hist[8];
    double bytes;        // algorithmic bytes counted by the kernel
};

struct SpBatch {
    int B, n, m, nC, nComp, N, Np, w, ld, nnzQ, nnzE;     // Np: N rounded up to a multiple of 64 (padding rows of the band arrays: unit diagonal)
    int hasLbL, hasLbR;
    lcqp_options_t opt;
    const int *Qp, *Qi, *Ep, *Ei, *ETp, *ETi, *ETmap, *iperm, *bandQ, *bandE;
    double *Qx, *Ex;         // [B][nnzQ], [B][nnzE] (CSR order)
    double *Ka, *KaC, *KaD;  // ADMM KKT factor: rows [B][N*ld], columns [B][N*w], 1/D [B][N]
    double *Kp, *KpC, *KpD;  // polish KKT factor
    double *nv, *mv, *Nv;    // [B][NV_NUM][n], [B][MV_NUM][m], [B][2][N]
    double *lbL, *lbR;       // [B][nComp]
    int* mi;                 // [B][MI_NUM][m]
    SpInfo* info;
    lcqp_stats_t* stats;
    double *xout, *yout;     // [B][n], [B][m]
};

struct SpCtx {
    const SpBatch* db;
    int b, n, m, nC, nComp, N, w, ld;
    const double *Qx, *Ex;
    double *Ka, *KaC, *KaD, *Kp, *KpC, *KpD, *nv, *mv, *Nv;
    int* mi;
    SpInfo* info;
    Lds lds;
    double* win;     // LDS: (w+1)^2 window + staging
    int cAdmm, cTrials, cFact, cCorr, cSweeps;
    double bytes;
    __device__ __forceinline__ double* V(int k) const { return nv + (size_t)k * n; }
    __device__ __forceinline__ double* M(int k) const { return mv + (size_t)k * m; }
    __device__ __forceinline__ int* I(int k) const { return mi + (size_t)k * m; }
};


__device__ __forceinline__ double sp_sum(double v) { return uniform_d(wave_sum(v)); }
__device__ __forceinline__ double sp_max(double v) { return uniform_d(wave_max(v)); }
__device__ __forceinline__ int sp_any(int v) { return __any(v) ? 1 : 0; }
__device__ __forceinline__ int sp_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return uniform_i(v);
}

// ---- sparse products: thread per row (CSR) / per column (CSC); rows carry 3-8 entries, the vectors stay in L2 ----------------
__device__ __forceinline__ void sp_Qx(const SpCtx& c, const double* x, double* out)
{
    const int* Qp = c.db->Qp; const int* Qi = c.db->Qi;
    for (int i = threadIdx.x; i < c.n; i += WGS) {
        double s = 0.0;
        for (int k = Qp[i]; k < Qp[i + 1]; k++) s += c.Qx[k] * x[Qi[k]];
        out[i] = s;
    }
    __syncthreads();
}
__device__ __forceinline__ void sp_Ex(const SpCtx& c, const double* x, double* out)
{
    const int* Ep = c.db->Ep; const int* Ei = c.db->Ei;
    for (int r = threadIdx.x; r < c.m; r += WGS) {
        double s = 0.0;
        for (int k = Ep[r]; k < Ep[r + 1]; k++) s += c.Ex[k] * x[Ei[k]];
        out[r] = s;
    }
    __syncthreads();
}
// out[i] = base(i) - (E'y)[i]   (column gather over the CSC of E; rows r0 <= r < r1 only)
template <class Base>
__device__ __forceinline__ void sp_ETy(const SpCtx& c, const double* y, double* out, Base base, int r0 = 0, int r1 = 1 << 30)
{
    const int *Tp = c.db->ETp, *Ti = c.db->ETi, *Tm = c.db->ETmap;
    for (int i = threadIdx.x; i < c.n; i += WGS) {
        double s = 0.0;
        for (int k = Tp[i]; k < Tp[i + 1]; k++) { const int r = Ti[k]; if (r >= r0 && r < r1) s += c.Ex[Tm[k]] * y[r]; }
        out[i] = base(i) - s;
    }
    __syncthreads();
}
// C v = L'(R v) + R'(L v): lx = E v (rows of L: nC .. nC+nComp, of R: nC+nComp ..), then a column gather with swapped coefficients
__device__ __forceinline__ void sp_Cx(const SpCtx& c, const double* v, double* out)
{
    double* lx = c.M(MV_LX);
    sp_Ex(c, v, lx);
    const int *Tp = c.db->ETp, *Ti = c.db->ETi, *Tm = c.db->ETmap;
    const int nC = c.nC, nK = c.nComp;
    for (int i = threadIdx.x; i < c.n; i += WGS) {
        double s = 0.0;
        for (int k = Tp[i]; k < Tp[i + 1]; k++) {
            const int r = Ti[k];
            if (r >= nC + nK) s += c.Ex[Tm[k]] * lx[r - nK];          // R' (L v)
            else if (r >= nC) s += c.Ex[Tm[k]] * lx[r + nK];          // L' (R v)
        }
        out[i] = s;
    }
    __syncthreads();
}
__device__ __forceinline__ double sp_maxabs(const double* a, int n)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += WGS) s = fmax(s, fabs(a[i]));
    return sp_max(s);
}

// ---- KKT assembly into band storage: Kb[i*ld + k] = K[i][i-w+k] in the ordering iperm ----------------------------------------
// variables: Q + dprim I; row r: -(use ? ddual(r) : 1) on the diagonal, its entries of E only when used
template <class Dd, class Use>
__device__ __forceinline__ void sp_assemble(SpCtx& c, double* Kb, double dprim, Dd ddual, Use use)
{
    const SpBatch& db = *c.db;
    const int t = threadIdx.x, ld = c.ld, w = c.w;
    for (int e = t; e < c.N * ld; e += WGS) Kb[e] = 0.0;
    __syncthreads();
    for (int k = t; k < db.nnzQ; k += WGS) { const int o = db.bandQ[k]; if (o >= 0) Kb[o] = c.Qx[k]; }
    for (int r = t; r < c.m; r += WGS) {
        const bool on = use(r);
        if (on) for (int k = db.Ep[r]; k < db.Ep[r + 1]; k++) Kb[db.bandE[k]] = c.Ex[k];
        Kb[(size_t)db.iperm[c.n + r] * ld + w] = on ? -ddual(r) : -1.0;
    }
    __syncthreads();
    for (int i = t; i < c.n; i += WGS) Kb[(size_t)db.iperm[i] * ld + w] += dprim;
    __syncthreads();
    c.bytes += 8.0 * ((double)c.N * ld + db.nnzQ + db.nnzE) + 4.0 * (db.nnzQ + db.nnzE);
}

// LDS traffic of one wave is in order; this keeps the compiler from moving LDS accesses across the point and waits for them
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- band LDL' with a sliding (w+1)^2 window in LDS -------------------------------------------------------------------------------
// The elimination is a chain of N column steps with O(w^2) work each: one wave runs it wave-synchronously (no workgroup barriers
// inside the chain), the rows that enter the window are fetched 16 columns ahead.
// in: Kb assembled band; out: Kb = unit lower factor by rows (diagonal slot: D), Kc[j*w + a-1] = L[j+a][j], Kd = 1/D
__device__ __forceinline__ void sp_factor(SpCtx& c, double* Kb, double* Kc, double* Kd)
{
    const int N = c.N, w = c.w, ld = c.ld, W1 = w + 1, l = lane_id();
    double* win = c.win;                  // W1 * W1
    double* stage = win + W1 * W1;        // 16 rows x W1
    double* lvec = stage + 16 * W1;       // w
    {
        for (int e = l; e < W1 * W1; e += 64) {
            const int r = e / W1, k = e - r * W1, cc = r - w + k;
            if (r < N && cc >= 0) win[(r % W1) * W1 + (cc % W1)] = Kb[(size_t)r * ld + k];
        }
        double pre[16];                    // rows j0 + w + 1 .. j0 + w + 16 of the NEXT block of columns, one band entry per lane and row
#pragma unroll
        for (int q = 0; q < 16; q++) { const int rn = w + 1 + q; pre[q] = (l <= w && rn < N) ? Kb[(size_t)rn * ld + l] : 0.0; }
        for (int j0 = 0; j0 < N; j0 += 16) {
#pragma unroll
            for (int q = 0; q < 16; q++) if (l <= w) stage[q * W1 + l] = pre[q];
#pragma unroll
            for (int q = 0; q < 16; q++) { const int rn = j0 + 16 + w + 1 + q; pre[q] = (l <= w && rn < N) ? Kb[(size_t)rn * ld + l] : 0.0; }
            wave_sync();
            const int j1 = min(N, j0 + 16);
            for (int j = j0; j < j1; j++) {
                const int jm = j % W1;
                const double d = win[jm * W1 + jm];
                if (l < w) {
                    const int r = j + l + 1;
                    double la = 0.0;
                    if (r < N) { la = win[(r % W1) * W1 + jm] / d; Kb[(size_t)r * ld + (w - l - 1)] = la; }
                    Kc[(size_t)j * w + l] = la;
                    lvec[l] = la;
                }
                if (l == 0) { Kb[(size_t)j * ld + w] = d; Kd[j] = 1.0 / d; }
                wave_sync();
                for (int e = l; e < w * w; e += 64) {
                    const int a = e / w + 1, bb = e - (a - 1) * w + 1;
                    if (bb <= a && j + a < N) win[((j + a) % W1) * W1 + ((j + bb) % W1)] -= lvec[a - 1] * d * lvec[bb - 1];
                }
                wave_sync();
                const int rn = j + w + 1;
                if (l <= w && rn < N) win[(rn % W1) * W1 + ((j + 1 + l) % W1)] = stage[(j - j0) * W1 + l];
                wave_sync();
            }
        }
    }
    __syncthreads();
    c.bytes += 8.0 * (3.0 * (double)N * ld);
    c.cFact++;
}

// ---- band solve K z = b (b in band order, in place): wave 0 keeps the 64 pending rows in its lanes ---------------------------
// Row r lives in lane r mod 64 (forward) while steps r-63 .. r run; every step broadcasts the finished entry (v_readlane with a
// constant lane: the 64 steps of a block are unrolled) and every lane subtracts its multiple -- axpy form both ways, no reduction
// in the chain.  Factor entries are loaded 16 steps ahead, right-hand sides and results move 64 rows at a time (coalesced).
// The arrays are padded to a multiple of 64 rows (unit diagonal, zero off-diagonal), so no step needs a bounds test.
__device__ __forceinline__ void sp_solve(SpCtx& c, const double* Kb, const double* Kc, const double* Kd, double* b)
{
    const int Np = c.db->Np, w = c.w, ld = c.ld, l = lane_id();
    {
        // forward: L y = b; z = y / D is what is stored
        double cur = b[l];
        for (int blk = 0; blk < Np; blk += 64) {
            const double nxt = (blk + 64 < Np) ? b[blk + 64 + l] : 0.0;              // this lane's next row
            const double dl = Kd[blk + l];
            double res = 0.0;
#pragma unroll
            for (int s0 = 0; s0 < 64; s0 += 16) {
                double lk[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int a = (l - (s0 + q)) & 63;
                    lk[q] = (a >= 1 && a <= w) ? Kc[(size_t)(blk + s0 + q) * w + a - 1] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const double yj = wave_bcast(cur, s0 + q);
                    cur -= lk[q] * yj;
                    if (l == s0 + q) { res = yj; cur = nxt; }
                }
            }
            b[blk + l] = res * dl;
        }
        // backward: L' x = z; lane l holds row base - l - 64 q
        const int base = Np - 1;
        cur = b[base - l];
        for (int blk = 0; blk < Np; blk += 64) {
            const double nxt = (blk + 64 < Np) ? b[base - blk - 64 - l] : 0.0;
            double res = 0.0;
#pragma unroll
            for (int s0 = 0; s0 < 64; s0 += 16) {
                double lk[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int a = (l - (s0 + q)) & 63, i = base - (blk + s0 + q);
                    lk[q] = (a >= 1 && a <= w && i - a >= 0) ? Kb[(size_t)i * ld + (w - a)] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const double xi = wave_bcast(cur, s0 + q);
                    cur -= lk[q] * xi;
                    if (l == s0 + q) { res = xi; cur = nxt; }
                }
            }
            b[base - blk - l] = res;
        }
    }
    __syncthreads();
    c.bytes += 8.0 * (2.0 * (double)c.N * w + 4.0 * c.N);
}

// ---- ADMM iterations (OSQP, KKT form; oracle: sqp_admm) ----------------------------------------------------------------------
__device__ __forceinline__ void sp_admm(SpCtx& c, const double* g, int n_it)
{
    const SpBatch& db = *c.db;
    const int t = threadIdx.x, n = c.n, m = c.m;
    const double alpha = db.opt.admmAlpha, sigma = c.info->sigma;
    double *xa = c.V(NV_XA), *ya = c.M(MV_YA), *za = c.M(MV_ZA), *b = c.Nv;
    const double *l = c.M(MV_L), *u = c.M(MV_U), *rhov = c.M(MV_RHOV);
    for (int it = 0; it < n_it; it++) {
        for (int i = t; i < n; i += WGS) b[db.iperm[i]] = sigma * xa[i] - g[i];
        for (int r = t; r < m; r += WGS) b[db.iperm[n + r]] = za[r] - ya[r] / rhov[r];
        __syncthreads();
        sp_solve(c, c.Ka, c.KaC, c.KaD, b);
        for (int r = t; r < m; r += WGS) {
            const double rv = rhov[r];
            const double zt = za[r] + (b[db.iperm[n + r]] - ya[r]) / rv;
            const double zr = alpha * zt + (1.0 - alpha) * za[r];
            if (isinf(l[r]) && isinf(u[r])) { za[r] = zr; ya[r] = 0.0; continue; }
            const double zn = fmin(fmax(zr + ya[r] / rv, l[r]), u[r]);
            ya[r] += rv * (zr - zn);
            za[r] = zn;
        }
        for (int i = t; i < n; i += WGS) xa[i] = alpha * b[db.iperm[i]] + (1.0 - alpha) * xa[i];
        __syncthreads();
        c.cAdmm++;
    }
}

// ---- primal-dual active-set polish in correction form (oracle: sqp_polish) ---------------------------------------------------
__device__ __forceinline__ int sp_polish(SpCtx& c, const double* g, int reuse)
{
    const SpBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = threadIdx.x, n = c.n, m = c.m;
    double *x = c.V(NV_XT), *r1 = c.V(NV_R1), *qx = c.V(NV_TMP), *yt = c.M(MV_YT), *ex = c.M(MV_EX), *b = c.Nv;
    const double *l = c.M(MV_L), *u = c.M(MV_U);
    int *st = c.I(MI_STT), *stf = c.I(MI_STF), *newst = c.I(MI_NEW);
    const double gs = 1.0 + sp_maxabs(g, n);
    const double ytol = o.feasTol * gs;
    int fact_valid = 0;
    for (int trial = 0; trial < o.maxTrials; trial++) {
        c.cTrials++;
        if (trial == 0 && reuse) {
            const double *r1s = c.V(NV_R1S), *gs0 = c.V(NV_GS), *exs = c.M(MV_EXS);
            for (int i = t; i < n; i += WGS) r1[i] = r1s[i] + (gs0[i] - g[i]);
            for (int r = t; r < m; r += WGS) ex[r] = exs[r];
            __syncthreads();
        } else {
            sp_Qx(c, x, qx);
            sp_ETy(c, yt, r1, [&](int i) { return -g[i] - qx[i]; });
            sp_Ex(c, x, ex);
            c.cSweeps++;
            c.bytes += 12.0 * (db.nnzQ + 2.0 * db.nnzE) + 8.0 * (4.0 * n + 2.0 * m);
        }
        const double res_stat = sp_maxabs(r1, n);
        double res_eq = 0.0, bmax = 0.0;
        int chg = 0, act = 0;
        for (int r = t; r < m; r += WGS) {
            const int s = st[r];
            int ns = s;
            const double e = ex[r];
            if (s == ST_INACT) {
                const double ftol = o.feasTol * (1.0 + fabs(e));
                if (e < l[r] - ftol) ns = ST_LOWER;
                else if (e > u[r] + ftol) ns = ST_UPPER;
            } else {
                const double bb = (s == ST_UPPER) ? u[r] : l[r];
                res_eq = fmax(res_eq, fabs(bb - e));
                bmax = fmax(bmax, fabs(bb));
                if (s == ST_LOWER && yt[r] > ytol) ns = ST_INACT;
                if (s == ST_UPPER && yt[r] < -ytol) ns = ST_INACT;
            }
            newst[r] = ns;
            chg += (ns != s);
            act += (ns != ST_INACT);
        }
        const int changed = sp_sum_i(chg), nact = sp_sum_i(act);
        res_eq = sp_max(res_eq);
        bmax = sp_max(bmax);
        if (trial > 0 && !changed && res_stat <= o.resTol * gs && res_eq <= o.resTol * (1.0 + bmax)) {
            double *r1s = c.V(NV_R1S), *gs0 = c.V(NV_GS), *exs = c.M(MV_EXS);
            for (int i = t; i < n; i += WGS) { r1s[i] = r1[i]; gs0[i] = g[i]; }
            for (int r = t; r < m; r += WGS) exs[r] = ex[r];
            __syncthreads();
            return 1;
        }
        if (changed && trial > 0) {
            if (trial >= 2 && nact > n && changed > max(n / 2, 32)) return 0;       // overshooting cold start: hand over to ADMM
            // leaving rows: their multipliers leave the residual (r1 += E_r' y_r), then the new working set takes over
            double* ytmp = c.M(MV_LX);
            for (int r = t; r < m; r += WGS) ytmp[r] = (newst[r] == ST_INACT && st[r] != ST_INACT) ? -yt[r] : 0.0;
            __syncthreads();
            sp_ETy(c, ytmp, r1, [&](int i) { return r1[i]; });             // r1 - E'(-y_leaving) = r1 + E'y_leaving
            for (int r = t; r < m; r += WGS) { if (ytmp[r] != 0.0) yt[r] = 0.0; st[r] = newst[r]; }
            __syncthreads();
            fact_valid = 0;
        }
        if (!fact_valid) {
            int diff = (c.info->stfValid == 0);
            for (int r = t; r < m; r += WGS) diff |= ((stf[r] != ST_INACT) != (st[r] != ST_INACT));
            if (sp_any(diff)) {
                const double d2 = c.info->delta2;
                sp_assemble(c, c.Kp, c.info->delta, [=](int) { return d2; }, [=](int r) { return st[r] != ST_INACT; });
                sp_factor(c, c.Kp, c.KpC, c.KpD);
                for (int r = t; r < m; r += WGS) stf[r] = st[r];
                if (t == 0) c.info->stfValid = 1;
                __syncthreads();
            }
            fact_valid = 1;
        }
        // correction: [Q + delta I, Ea'; Ea, -delta2 I][dx; dy] = [r1; ba - Ea x]
        for (int i = t; i < n; i += WGS) b[db.iperm[i]] = r1[i];
        for (int r = t; r < m; r += WGS) {
            double v = 0.0;
            if (st[r] != ST_INACT) v = ((st[r] == ST_UPPER) ? u[r] : l[r]) - ex[r];
            b[db.iperm[n + r]] = v;
        }
        __syncthreads();
        sp_solve(c, c.Kp, c.KpC, c.KpD, b);
        for (int i = t; i < n; i += WGS) x[i] += b[db.iperm[i]];
        for (int r = t; r < m; r += WGS) if (st[r] != ST_INACT) yt[r] += b[db.iperm[n + r]];
        __syncthreads();
        c.cCorr++;
    }
    return 0;
}

// ---- SubsolverBase::solve on the OSQP arm (oracle: sqp_solve) ------------------------------------------------------------------
__device__ __forceinline__ int sp_qp_solve(SpCtx& c, int initial, const double* g, int* iterations)
{
    const SpBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = threadIdx.x, n = c.n, m = c.m;
    const int trials0 = c.cTrials, admm0 = c.cAdmm;
    double *xq = c.V(NV_XQ), *xa = c.V(NV_XA), *xt = c.V(NV_XT);
    double *yq = c.M(MV_YQ), *ya = c.M(MV_YA), *za = c.M(MV_ZA), *yt = c.M(MV_YT);
    const double *l = c.M(MV_L), *u = c.M(MV_U);
    int *st = c.I(MI_ST), *stt = c.I(MI_STT);
    *iterations = 0;
    int bad = 0;
    for (int r = t; r < m; r += WGS) bad |= (l[r] > u[r]);
    if (sp_any(bad)) return 2;
    if (initial) {
        const double *x0 = c.V(NV_X0), *y0 = c.M(MV_Y0);
        for (int i = t; i < n; i += WGS) xq[i] = x0[i];
        for (int r = t; r < m; r += WGS) yq[r] = c.info->hasY0 ? -y0[r] : 0.0;
        __syncthreads();
    }
    for (int i = t; i < n; i += WGS) xa[i] = xq[i];
    for (int r = t; r < m; r += WGS) ya[r] = yq[r];
    __syncthreads();
    int n_admm = initial ? o.admmFirst : o.admmHot;
    const int use_stored = (!initial && c.info->haveSolution && n_admm == 0);
    int solved = 0, admm_ready = 0;
    for (int round = 0; round < o.maxRounds && !solved; round++) {
        if (!admm_ready && (n_admm > 0 || !(round == 0 && use_stored))) {
            sp_Ex(c, xa, za);
            for (int r = t; r < m; r += WGS) { za[r] = fmin(fmax(za[r], l[r]), u[r]); if (isinf(l[r]) && isinf(u[r])) ya[r] = 0.0; }
            __syncthreads();
            admm_ready = 1;
        }
        if (n_admm > 0) sp_admm(c, g, n_admm);
        for (int r = t; r < m; r += WGS) {
            int s;
            if (round == 0 && use_stored) { s = st[r]; if (l[r] == u[r]) s = ST_EQ; }
            else {
                const double lo = l[r], hi = u[r], z = za[r], y = ya[r];
                s = ST_INACT;
                if (isfinite(lo) && (z - lo < -y)) s = ST_LOWER;
                if (isfinite(hi) && (hi - z < y)) s = ST_UPPER;
                if (lo == hi) s = ST_EQ;
            }
            stt[r] = s;
            yt[r] = (s != ST_INACT) ? ya[r] : 0.0;
        }
        for (int i = t; i < n; i += WGS) xt[i] = xa[i];
        __syncthreads();
        if (sp_polish(c, g, round == 0 && use_stored)) { solved = 1; break; }
        n_admm = 2 * n_admm;
        if (n_admm < 10) n_admm = 10;
        if (n_admm > 400) n_admm = 400;
    }
    *iterations = (c.cTrials - trials0) + (c.cAdmm - admm0);
    if (!solved) return 1;
    for (int i = t; i < n; i += WGS) xq[i] = xt[i];
    for (int r = t; r < m; r += WGS) { yq[r] = yt[r]; st[r] = stt[r]; }
    if (t == 0) c.info->haveSolution = 1;
    __syncthreads();
    return 0;
}

// ---- setup: scales, rho vector, phi expressions, the ONE factorisation of the ADMM KKT matrix ----------------------------------
__global__ __launch_bounds__(WGS) void k_sparse_setup(SpBatch db);
__global__ __launch_bounds__(WGS) void k_sparse_run(SpBatch db);

#define SP_LDS extern __shared__ double sp_dyn_lds[];  Lds lds{sp_dyn_lds, nullptr, nullptr};

__device__ __forceinline__ SpCtx sp_ctx(const SpBatch& db, int b, Lds lds)
{
    SpCtx c;
    c.db = &db; c.b = b; c.n = db.n; c.m = db.m; c.nC = db.nC; c.nComp = db.nComp; c.N = db.N; c.w = db.w; c.ld = db.ld;
    c.Qx = db.Qx + (size_t)b * db.nnzQ; c.Ex = db.Ex + (size_t)b * db.nnzE;
    c.Ka = db.Ka + (size_t)b * db.Np * db.ld; c.KaC = db.KaC + (size_t)b * db.Np * db.w; c.KaD = db.KaD + (size_t)b * db.Np;
    c.Kp = db.Kp + (size_t)b * db.Np * db.ld; c.KpC = db.KpC + (size_t)b * db.Np * db.w; c.KpD = db.KpD + (size_t)b * db.Np;
    c.nv = db.nv + (size_t)b * NV_NUM * db.n; c.mv = db.mv + (size_t)b * MV_NUM * db.m; c.Nv = db.Nv + (size_t)b * 2 * db.Np;
    c.mi = db.mi + (size_t)b * MI_NUM * db.m;
    c.info = db.info + b;
    c.lds = lds; c.win = lds.arena;
    c.cAdmm = c.cTrials = c.cFact = c.cCorr = c.cSweeps = 0;
    c.bytes = 0.0;
    return c;
}

__global__ __launch_bounds__(WGS) void k_sparse_setup(SpBatch db)
{
    SP_LDS
    SpCtx c = sp_ctx(db, blockIdx.x, lds);
    const int t = threadIdx.x, n = c.n, m = c.m, nC = c.nC, nK = c.nComp;
    double dmax = 0.0;
    for (int i = t; i < n; i += WGS)
        for (int k = db.Qp[i]; k < db.Qp[i + 1]; k++) if (db.Qi[k] == i) dmax = fmax(dmax, fabs(c.Qx[k]));
    double scale = sp_max(dmax);
    if (!(scale > 1e-300)) scale = 1.0;
    const double rho = db.opt.admmRho * scale;
    double *l = c.M(MV_L), *u = c.M(MV_U), *rhov = c.M(MV_RHOV);
    for (int r = t; r < m; r += WGS) {
        double rv = rho;
        if (isinf(l[r]) && isinf(u[r])) rv = 1e-6 * rho;
        else if (l[r] == u[r]) rv = rho * db.opt.rhoEqMult;
        rhov[r] = rv;
    }
    // phi expressions (src/LCQProblem.cpp:969-996)
    double phiConst = 0.0;
    double* gphi = c.V(NV_GPHI);
    if (db.hasLbL || db.hasLbR) {
        const double* lbL = db.lbL + (size_t)c.b * nK;
        const double* lbR = db.lbR + (size_t)c.b * nK;
        double s = 0.0;
        for (int i = t; i < nK; i += WGS) s += lbL[i] * lbR[i];
        phiConst = sp_sum(s);
        double* coef = c.M(MV_LX);
        for (int r = t; r < m; r += WGS) coef[r] = (r >= nC + nK) ? lbL[r - nC - nK] : ((r >= nC) ? lbR[r - nC] : 0.0);     // R'lbL + L'lbR
        __syncthreads();
        sp_ETy(c, coef, gphi, [](int) { return 0.0; });
    } else {
        for (int i = t; i < n; i += WGS) gphi[i] = 0.0;
    }
    if (t == 0) {
        c.info->scale = scale; c.info->sigma = db.opt.admmSigma * scale; c.info->delta = db.opt.proxBig * scale; c.info->delta2 = 1e-9 / scale;
        c.info->phiConst = phiConst; c.info->haveSolution = 0; c.info->stfValid = 0; c.info->bytes = 0.0;
    }
    __syncthreads();
    sp_assemble(c, c.Ka, db.opt.admmSigma * scale, [=](int r) { return 1.0 / rhov[r]; }, [](int) { return true; });
    sp_factor(c, c.Ka, c.KaC, c.KaD);
    if (t == 0) c.info->bytes = c.bytes;
}

// ---- LCQProblem::runSolver, OSQP_SPARSE arm (oracle: orc_sparse_lcqp_solve) ----------------------------------------------------
__global__ __launch_bounds__(WGS, 8) void k_sparse_run(SpBatch db)
{
    SP_LDS
    SpCtx c = sp_ctx(db, blockIdx.x, lds);
    const lcqp_options_t& o = db.opt;
    const int t = threadIdx.x, n = c.n, m = c.m, nC = c.nC, nK = c.nComp;
    double *g = c.V(NV_G), *gtil = c.V(NV_GTIL), *gphi = c.V(NV_GPHI), *xk = c.V(NV_XK), *pk = c.V(NV_PK), *xnew = c.V(NV_XNEW), *gk = c.V(NV_GK);
    double *Qx = c.V(NV_QX), *Cx = c.V(NV_CX), *Qp = c.V(NV_QP), *Cp = c.V(NV_CP), *tmp = c.V(NV_TMP);
    double *yk = c.M(MV_YK), *lx = c.M(MV_LX);
    const bool hasPhi = db.hasLbL || db.hasLbR;
    const double phiConst = c.info->phiConst;
    double* hist = c.info->hist;
    lcqp_stats_t st;
    memset(&st, 0, sizeof(st));
    int rc = 0, qpIter = 0, histLen = 0, algoStat = 0, totalIter = 0;
    double alphak = 1.0, rho = o.initialPenaltyParameter;
    uint64_t perturbCounter = 0;
    for (int i = t; i < n; i += WGS) { xk[i] = c.V(NV_X0)[i]; gtil[i] = g[i]; }
    __syncthreads();
    auto getPhi = [&]() -> double {
        double s = 0.0;
        for (int i = t; i < n; i += WGS) s += (hasPhi ? gphi[i] * xk[i] : 0.0) + 0.5 * xk[i] * Cx[i];
        return phiConst + sp_sum(s);
    };
    auto updatePenalty = [&]() {
        if (o.nDynamicPenalty > 0) histLen = 0;
        rho *= o.penaltyUpdateFactor;
        st.rhoOpt = rho;
        if (hasPhi) { for (int i = t; i < n; i += WGS) gtil[i] = g[i] + rho * gphi[i]; __syncthreads(); }
    };
    if (o.solveZeroPenaltyFirst) { for (int i = t; i < n; i += WGS) gk[i] = g[i]; __syncthreads(); }
    else { sp_Cx(c, xk, Cx); for (int i = t; i < n; i += WGS) gk[i] = rho * Cx[i] + gtil[i]; __syncthreads(); }
    int initial = 1;
    for (;;) {
        const int ef = sp_qp_solve(c, initial, gk, &qpIter);
        st.subproblemIter += qpIter; st.qpSolverExitFlag = ef; st.qpSolves++;
        if (ef != 0) { rc = LCQP_SUBPROBLEM_SOLVER_ERROR; break; }
        {
            const double *xq = c.V(NV_XQ), *yq = c.M(MV_YQ);
            for (int i = t; i < n; i += WGS) { xnew[i] = xq[i]; pk[i] = xq[i] - xk[i]; }
            for (int r = t; r < m; r += WGS) yk[r] = -yq[r];                         // src/SubsolverOSQP.cpp:196-199
            __syncthreads();
        }
        if (initial) st.rhoOpt = rho;
        else if (o.perturbStep) {
            for (int i = t; i < n; i += WGS) {
                uint64_t z = o.perturbSeed + (perturbCounter + (uint64_t)i + 1ULL) * 0x9E3779B97F4A7C15ULL;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z = z ^ (z >> 31);
                xk[i] += ((int)(z % 3ULL) - 1) * 2.221e-16;
            }
            perturbCounter += (uint64_t)n;
            __syncthreads();
        }
        sp_Qx(c, pk, Qp); sp_Qx(c, xk, Qx); sp_Cx(c, pk, Cp); sp_Cx(c, xk, Cx);
        c.bytes += 2.0 * 12.0 * db.nnzQ + 4.0 * 12.0 * db.nnzE;
        if (!initial) {
            double sq = 0.0, sl = 0.0;
            for (int i = t; i < n; i += WGS) { sq += pk[i] * (Qp[i] + rho * Cp[i]); sl += pk[i] * ((Qx[i] + rho * Cx[i]) + gtil[i]); }
            const double qk = sp_sum(sq), lk = sp_sum(sl);
            alphak = 1.0;
            if (qk > 0 && lk < 0) alphak = fmin(-lk / qk, 1.0);
        }
        initial = 0;
        for (int i = t; i < n; i += WGS) { xk[i] += alphak * pk[i]; Qx[i] += alphak * Qp[i]; Cx[i] += alphak * Cp[i]; }
        __syncthreads();
        // updateStationarity without a box term: statk = Qk xk + g_tilde - E'yk
        sp_ETy(c, yk, tmp, [&](int i) { return (Qx[i] + rho * Cx[i]) + gtil[i]; });
        const double statInf = sp_maxabs(tmp, n);
        totalIter++; st.iterTotal++;
        bool leyffer = false;
        const int nd = o.nDynamicPenalty;
        if (nd > 0) {
            const double cur = getPhi();
            if (histLen < nd) { if (t == 0) hist[histLen] = cur; histLen++; __syncthreads(); }
            else {
                if (!(cur < o.complementarityTolerance)) {
                    leyffer = true;
                    for (int i = 0; i < nd; i++) if (cur < o.etaDynamicPenalty * hist[i]) { leyffer = false; break; }
                }
                __syncthreads();
                if (t == 0) { for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1]; hist[nd - 1] = cur; }
                __syncthreads();
            }
        }
        if (leyffer) { updatePenalty(); st.iterOuter++; }
        if (statInf < o.stationarityTolerance) {
            if (getPhi() < o.complementarityTolerance) {
                sp_Ex(c, xk, lx);
                int sflag = 1, mflag = 1, wflag = 0;
                const double ctol = o.complementarityTolerance;
                for (int i = 0; i < nK; i++) {
                    const double Lx = lx[nC + i], Rx = lx[nC + nK + i];
                    if (!(Lx <= ctol && Rx <= ctol)) continue;
                    const double a = yk[nC + i], bq = yk[nC + nK + i];
                    const double dualProd = a * bq, dualMin = fmin(a, bq);
                    if (dualMin < 0) sflag = 0;
                    if (fabs(dualProd) >= ctol && dualMin <= 0) { if (dualProd <= ctol) { wflag = 1; break; } mflag = 0; }
                }
                algoStat = wflag ? 1 : (sflag ? 4 : (mflag ? 3 : 2));
                __syncthreads();
                for (int i = t; i < nK; i += WGS) { const double Lx = lx[nC + i], Rx = lx[nC + nK + i]; yk[nC + i] -= rho * Rx; yk[nC + nK + i] -= rho * Lx; }
                __syncthreads();
                rc = 0;
                break;
            }
            updatePenalty(); st.iterOuter++;
        }
        if (totalIter > o.maxIterations) { rc = LCQP_MAX_ITERATIONS_REACHED; break; }
        if (rho > o.maxPenaltyParameter) { rc = LCQP_MAX_PENALTY_REACHED; break; }
        for (int i = t; i < n; i += WGS) gk[i] = rho * Cx[i] + gtil[i];
        __syncthreads();
    }
    st.status = algoStat; st.returnValue = rc;
    st.admmIter = c.cAdmm; st.trials = c.cTrials; st.factorizations = c.cFact; st.corrections = c.cCorr; st.reserved = c.cSweeps;
    for (int i = t; i < n; i += WGS) db.xout[(size_t)c.b * n + i] = xk[i];
    for (int r = t; r < m; r += WGS) db.yout[(size_t)c.b * m + r] = yk[r];
    if (t == 0) { db.stats[c.b] = st; c.info->bytes += c.bytes; }
}

}  // namespace

// =================================================================================================
// host side
// =================================================================================================
static thread_local std::string g_sp_err;
extern "C" const char* lcqp_hip_sparse_last_error(void) { return g_sp_err.c_str(); }

struct lcqp_hip_sparse {
    SpBatch db;
    int device, nnzA;
    hipStream_t stream;
    hipEvent_t ev0, ev1, ev2;
    std::vector<void*> allocs;
    std::vector<int> csr2csc;      // value order: E (CSR) entry k comes from entry csr2csc[k] of the caller's CSC arrays
    std::vector<int> perm;
    bool loaded, ran;
};

#define SPCHK(call)                                                                             \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_sp_err = std::string(#call) + ": " + hipGetErrorString(e_); return LCQP_HIP_ERROR; } } while (0)

template <class T>
static T* sp_alloc(lcqp_hip_sparse* h, size_t count, const T* init = nullptr)
{
    void* p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    h->allocs.push_back(p);
    if (init) { if (hipMemcpy(p, init, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr; }
    else if (hipMemset(p, 0, bytes) != hipSuccess) return nullptr;
    return (T*)p;
}

// reverse Cuthill-McKee ordering of the KKT graph: nodes 0..n-1 variables, n..n+m-1 rows
static void rcm_order(int N, const std::vector<std::vector<int>>& adj, std::vector<int>& perm)
{
    std::vector<int> deg(N), order; std::vector<char> seen(N, 0);
    for (int i = 0; i < N; i++) deg[i] = (int)adj[i].size();
    order.reserve(N);
    auto bfs = [&](int start, std::vector<int>& out, std::vector<char>& mark) {
        std::queue<int> q; q.push(start); mark[start] = 1;
        while (!q.empty()) {
            const int v = q.front(); q.pop(); out.push_back(v);
            std::vector<int> nb;
            for (int u : adj[v]) if (!mark[u]) { mark[u] = 1; nb.push_back(u); }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
            for (int u : nb) q.push(u);
        }
    };
    for (int s0 = 0; s0 < N; s0++) {
        if (seen[s0]) continue;
        // pseudo-peripheral start: the last node of a BFS from the minimum-degree node of the component, twice
        int start = s0;
        for (int pass = 0; pass < 2; pass++) {
            std::vector<int> tmp; std::vector<char> mk(seen.begin(), seen.end());
            bfs(start, tmp, mk);
            if (pass == 0) { int best = tmp[0]; for (int v : tmp) if (deg[v] < deg[best]) best = v; start = best; }
            else start = tmp.back();
        }
        bfs(start, order, seen);
    }
    perm.assign(order.rbegin(), order.rend());
}

extern "C" lcqp_hip_sparse_t* lcqp_hip_sparse_create(int batch, int nV, int nC, int nComp, const int* Qp, const int* Qi, const int* Ap, const int* Ai, int device)
try {
    if (batch <= 0 || nV <= 0 || nC < 0 || nComp <= 0 || !Qp || !Qi || !Ap || !Ai) { g_sp_err = "invalid arguments"; return nullptr; }
    const int n = nV, m = nC + 2 * nComp, N = n + m, nnzQ = Qp[n], nnzA = Ap[n];
    // CSC of the stacked matrix -> CSR (pattern and the value permutation)
    std::vector<int> Ep(m + 1, 0), Ei(nnzA), csr2csc(nnzA), ETp(Ap, Ap + n + 1), ETi(Ai, Ai + nnzA), ETmap(nnzA);
    for (int k = 0; k < nnzA; k++) { if (Ai[k] < 0 || Ai[k] >= m) { g_sp_err = "row index out of bounds"; return nullptr; } Ep[Ai[k] + 1]++; }
    for (int r = 0; r < m; r++) Ep[r + 1] += Ep[r];
    { std::vector<int> cur(Ep.begin(), Ep.end() - 1);
      for (int c = 0; c < n; c++) for (int k = Ap[c]; k < Ap[c + 1]; k++) { const int d = cur[Ai[k]]++; Ei[d] = c; csr2csc[d] = k; ETmap[k] = d; } }
    // KKT graph and ordering
    std::vector<std::vector<int>> adj(N);
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) { const int j = Qi[k]; if (j < 0 || j >= n) { g_sp_err = "Q index out of bounds"; return nullptr; } if (j != i) adj[i].push_back(j); }
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) { adj[n + r].push_back(Ei[k]); adj[Ei[k]].push_back(n + r); }
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    std::vector<int> perm, iperm(N);
    rcm_order(N, adj, perm);
    for (int p = 0; p < N; p++) iperm[perm[p]] = p;
    int w = 0;
    for (int v = 0; v < N; v++) for (int u : adj[v]) w = std::max(w, std::abs(iperm[v] - iperm[u]));
    if (w > SP_WMAX) {
        g_sp_err = "KKT band of this pattern has half bandwidth " + std::to_string(w) + " > " + std::to_string(SP_WMAX) +
                   " after reverse Cuthill-McKee: not a banded problem (use the dense kernels)";
        return nullptr;
    }
    if (w < 1) w = 1;
    const int ld = w + 1;
    std::vector<int> bandQ(nnzQ, -1), bandE(nnzA);
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) { const int pi = iperm[i], pj = iperm[Qi[k]]; if (pj <= pi) bandQ[k] = pi * ld + w - (pi - pj); }
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) { const int pr = iperm[n + r], pc = iperm[Ei[k]]; const int hi = std::max(pr, pc), lo = std::min(pr, pc); bandE[k] = hi * ld + w - (hi - lo); }
    if (hipSetDevice(device) != hipSuccess) { g_sp_err = "hipSetDevice failed"; return nullptr; }
    lcqp_hip_sparse* h = new (std::nothrow) lcqp_hip_sparse();
    if (!h) return nullptr;
    h->device = device; h->nnzA = nnzA; h->loaded = false; h->ran = false; h->csr2csc = csr2csc; h->perm = perm;
    h->stream = nullptr; h->ev0 = h->ev1 = h->ev2 = nullptr;
    SpBatch& d = h->db;
    memset(&d, 0, sizeof(d));
    d.B = batch; d.n = n; d.m = m; d.nC = nC; d.nComp = nComp; d.N = N; d.Np = ((N + 63) / 64) * 64; d.w = w; d.ld = ld; d.nnzQ = nnzQ; d.nnzE = nnzA;
    const size_t Np = d.Np;
    lcqp_hip_options_default(&d.opt);
    bool ok = hipStreamCreate(&h->stream) == hipSuccess && hipEventCreate(&h->ev0) == hipSuccess && hipEventCreate(&h->ev1) == hipSuccess &&
              hipEventCreate(&h->ev2) == hipSuccess;
    const size_t B = batch;
    ok = ok && (d.Qp = sp_alloc<int>(h, n + 1, Qp)) && (d.Qi = sp_alloc<int>(h, nnzQ, Qi)) && (d.Ep = sp_alloc<int>(h, m + 1, Ep.data())) &&
         (d.Ei = sp_alloc<int>(h, nnzA, Ei.data())) && (d.ETp = sp_alloc<int>(h, n + 1, ETp.data())) && (d.ETi = sp_alloc<int>(h, nnzA, ETi.data())) &&
         (d.ETmap = sp_alloc<int>(h, nnzA, ETmap.data())) && (d.iperm = sp_alloc<int>(h, N, iperm.data())) &&
         (d.bandQ = sp_alloc<int>(h, nnzQ, bandQ.data())) && (d.bandE = sp_alloc<int>(h, nnzA, bandE.data()));
    ok = ok && (d.Qx = sp_alloc<double>(h, B * nnzQ)) && (d.Ex = sp_alloc<double>(h, B * nnzA)) &&
         (d.Ka = sp_alloc<double>(h, B * Np * ld)) && (d.KaC = sp_alloc<double>(h, B * Np * w)) && (d.KaD = sp_alloc<double>(h, B * Np)) &&
         (d.Kp = sp_alloc<double>(h, B * Np * ld)) && (d.KpC = sp_alloc<double>(h, B * Np * w)) && (d.KpD = sp_alloc<double>(h, B * Np)) &&
         (d.nv = sp_alloc<double>(h, B * NV_NUM * n)) && (d.mv = sp_alloc<double>(h, B * MV_NUM * m)) && (d.Nv = sp_alloc<double>(h, B * 2 * Np)) &&
         (d.lbL = sp_alloc<double>(h, B * nComp)) && (d.lbR = sp_alloc<double>(h, B * nComp)) && (d.mi = sp_alloc<int>(h, B * MI_NUM * m)) &&
         (d.info = sp_alloc<SpInfo>(h, B)) && (d.stats = sp_alloc<lcqp_stats_t>(h, B)) && (d.xout = sp_alloc<double>(h, B * n)) &&
         (d.yout = sp_alloc<double>(h, B * m));
    if (!ok) { g_sp_err = "device allocation failed"; lcqp_hip_sparse_destroy(h); return nullptr; }
    return h;
}
catch (...) { g_sp_err = "out of host memory"; return nullptr; }

extern "C" void lcqp_hip_sparse_destroy(lcqp_hip_sparse_t* h)
try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}
catch (...) { }

extern "C" int lcqp_hip_sparse_bandwidth(const lcqp_hip_sparse_t* h) { return h ? h->db.w : -1; }
extern "C" int lcqp_hip_sparse_get_ordering(const lcqp_hip_sparse_t* h, int* perm)
{
    if (!h || !perm) return LCQP_INVALID_ARGUMENT;
    memcpy(perm, h->perm.data(), sizeof(int) * h->perm.size());
    return 0;
}

extern "C" int lcqp_hip_sparse_set_options(lcqp_hip_sparse_t* h, const lcqp_options_t* opt)
{
    if (!h || !opt) return LCQP_INVALID_ARGUMENT;
    if (opt->nDynamicPenalty > 8) { g_sp_err = "nDynamicPenalty > 8 unsupported"; return LCQP_HIP_UNSUPPORTED; }
    h->db.opt = *opt;
    return 0;
}

static inline double spb(const double* p, size_t i, double dflt) { return p ? p[i] : dflt; }

// LCQProblem::loadLCQP (sparse overload, src/LCQProblem.cpp:390-441) for instances [first, first + count): values only -- the
// pattern was given to lcqp_hip_sparse_create.  Qx: [count][nnzQ]; Ax: [count][nnzA] in the CSC order of the stacked [A; L; R].
extern "C" int lcqp_hip_sparse_load(lcqp_hip_sparse_t* h, int first, int count, const double* Qx, const double* g, const double* Ax,
                                    const double* lbA, const double* ubA, const double* lbL, const double* ubL, const double* lbR,
                                    const double* ubR, const double* x0, const double* y0)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    SpBatch& d = h->db;
    const int n = d.n, m = d.m, nC = d.nC, nK = d.nComp;
    if (first < 0 || count <= 0 || first + count > d.B || !Qx || !Ax) return LCQP_INVALID_ARGUMENT;
    if (!g) return LCQP_INVALID_OBJECTIVE_LINEAR_TERM;
    SPCHK(hipSetDevice(h->device));
    const int hasL = lbL ? 1 : 0, hasR = lbR ? 1 : 0;
    if (!h->loaded || first == 0) { d.hasLbL = hasL; d.hasLbR = hasR; }
    else if (d.hasLbL != hasL || d.hasLbR != hasR) { g_sp_err = "lbL/lbR must be given for all instances of a batch or for none"; return LCQP_INVALID_ARGUMENT; }
    std::vector<double> ex(d.nnzE), nvb((size_t)NV_NUM * n), mvb((size_t)MV_NUM * m), lb(nK), rb(nK);
    for (int k = 0; k < count; k++) {
        const size_t b = (size_t)first + k;
        for (int e = 0; e < d.nnzE; e++) ex[e] = Ax[(size_t)k * d.nnzE + h->csr2csc[e]];
        std::fill(nvb.begin(), nvb.end(), 0.0); std::fill(mvb.begin(), mvb.end(), 0.0);
        for (int i = 0; i < n; i++) { nvb[(size_t)NV_G * n + i] = g[(size_t)k * n + i]; nvb[(size_t)NV_X0 * n + i] = x0 ? x0[(size_t)k * n + i] : 0.0; }
        double *lE = &mvb[(size_t)MV_L * m], *uE = &mvb[(size_t)MV_U * m];
        for (int r = 0; r < nC; r++) { lE[r] = spb(lbA, (size_t)k * nC + r, -INFINITY); uE[r] = spb(ubA, (size_t)k * nC + r, INFINITY); }
        for (int i = 0; i < nK; i++) {
            if (lbL && lbL[(size_t)k * nK + i] <= -INFINITY) return LCQP_INVALID_LOWER_COMPLEMENTARITY_BOUND;
            if (lbR && lbR[(size_t)k * nK + i] <= -INFINITY) return LCQP_INVALID_LOWER_COMPLEMENTARITY_BOUND;
            lE[nC + i] = spb(lbL, (size_t)k * nK + i, 0.0); uE[nC + i] = spb(ubL, (size_t)k * nK + i, INFINITY);
            lE[nC + nK + i] = spb(lbR, (size_t)k * nK + i, 0.0); uE[nC + nK + i] = spb(ubR, (size_t)k * nK + i, INFINITY);
            lb[i] = spb(lbL, (size_t)k * nK + i, 0.0); rb[i] = spb(lbR, (size_t)k * nK + i, 0.0);
        }
        if (y0) for (int r = 0; r < m; r++) mvb[(size_t)MV_Y0 * m + r] = y0[(size_t)k * m + r];
        SpInfo info; memset(&info, 0, sizeof(info)); info.hasY0 = y0 ? 1 : 0;
        SPCHK(hipMemcpy(d.Qx + b * d.nnzQ, Qx + (size_t)k * d.nnzQ, sizeof(double) * d.nnzQ, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.Ex + b * d.nnzE, ex.data(), sizeof(double) * d.nnzE, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.nv + b * NV_NUM * n, nvb.data(), sizeof(double) * nvb.size(), hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.mv + b * MV_NUM * m, mvb.data(), sizeof(double) * mvb.size(), hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.lbL + b * nK, lb.data(), sizeof(double) * nK, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.lbR + b * nK, rb.data(), sizeof(double) * nK, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.info + b, &info, sizeof(info), hipMemcpyHostToDevice));
    }
    h->loaded = true;
    return 0;
}
catch (...) { g_sp_err = "out of host memory"; return LCQP_HIP_ERROR; }

extern "C" int lcqp_hip_sparse_run(lcqp_hip_sparse_t* h)
try {
    if (!h || !h->loaded) return LCQP_LCQPOBJECT_NOT_SETUP;
    SPCHK(hipSetDevice(h->device));
    SPCHK(hipEventRecord(h->ev0, h->stream));
    const int W1 = h->db.w + 1;
    const size_t ldsBytes = sizeof(double) * (size_t)(W1 * W1 + 16 * W1 + W1 + 8);       // window, 16 staged rows, one column
    hipLaunchKernelGGL(k_sparse_setup, dim3(h->db.B), dim3(WGS), ldsBytes, h->stream, h->db);
    SPCHK(hipEventRecord(h->ev1, h->stream));
    hipLaunchKernelGGL(k_sparse_run, dim3(h->db.B), dim3(WGS), ldsBytes, h->stream, h->db);
    SPCHK(hipGetLastError());
    SPCHK(hipEventRecord(h->ev2, h->stream));
    h->ran = true;
    return 0;
}
catch (...) { g_sp_err = "out of host memory"; return LCQP_HIP_ERROR; }

extern "C" int lcqp_hip_sparse_synchronize(lcqp_hip_sparse_t* h)
{
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    SPCHK(hipSetDevice(h->device));
    SPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int lcqp_hip_sparse_last_timing(lcqp_hip_sparse_t* h, float* setup_ms, float* solve_ms)
{
    if (!h || !h->ran) return LCQP_INVALID_ARGUMENT;
    SPCHK(hipSetDevice(h->device));
    SPCHK(hipEventSynchronize(h->ev2));
    if (setup_ms) SPCHK(hipEventElapsedTime(setup_ms, h->ev0, h->ev1));
    if (solve_ms) SPCHK(hipEventElapsedTime(solve_ms, h->ev1, h->ev2));
    return 0;
}

extern "C" int lcqp_hip_sparse_get_solution(lcqp_hip_sparse_t* h, double* x, double* y, lcqp_stats_t* stats)
{
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    SPCHK(hipSetDevice(h->device));
    SpBatch& d = h->db;
    SPCHK(hipStreamSynchronize(h->stream));
    if (x) SPCHK(hipMemcpy(x, d.xout, sizeof(double) * (size_t)d.B * d.n, hipMemcpyDeviceToHost));
    if (y) SPCHK(hipMemcpy(y, d.yout, sizeof(double) * (size_t)d.B * d.m, hipMemcpyDeviceToHost));
    if (stats) SPCHK(hipMemcpy(stats, d.stats, sizeof(lcqp_stats_t) * (size_t)d.B, hipMemcpyDeviceToHost));
    return 0;
}

// algorithmic bytes of the last run (setup + homotopy), counted by the kernels: CSR values and indices of every sparse product,
// band storage read and written by every assembly, factorisation and solve
extern "C" double lcqp_hip_sparse_algorithmic_bytes(lcqp_hip_sparse_t* h)
try {
    if (!h) return 0.0;
    SpBatch& d = h->db;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return 0.0;
    std::vector<SpInfo> info(d.B);
    if (hipMemcpy(info.data(), d.info, sizeof(SpInfo) * (size_t)d.B, hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
    double tot = 0.0;
    for (auto& i : info) tot += i.bytes;
    return tot;
}
catch (...) { return 0.0; }
