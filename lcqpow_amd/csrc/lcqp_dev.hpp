// lcqp_dev.hpp -- device-side data layout and the per-instance solver logic (one workgroup = one LCQP).
//
// Restates, on top of the workgroup routines of lcqp_wg.hpp:
//   * the QP subsolver that stands where the reference calls qpOASES
//     (src/SubsolverQPOASES.cpp:134-181; algorithm: DESIGN.md §Subsolver), and
//   * LCQProblem::runSolver and its helpers (src/LCQProblem.cpp:444-560, 1105-1326, 1353-1482).
// The CPU oracle (oracle/lcqp_oracle.c) is the same algorithm in scalar C; tests compare the two.
#pragma once
#include "lcqp_wg.hpp"
#include "../../include/lcqp_hip.h"

namespace lcqp {

// per-instance vectors of length np (padded nV)
enum { V_G, V_GPHI, V_GTIL, V_XK, V_PK, V_XNEW, V_GK, V_QX, V_CX, V_QP, V_CP, V_STATK, V_TMP,
       V_XQ, V_XA, V_XT, V_R1, V_C, V_DU, V_W, V_RHS, V_LB, V_UB, V_X0, V_R1S, V_GS, V_ATY, V_QXN, V_XS, V_XREF, V_NUM };   // V_QXN: Q x at the last verified QP solution; V_XS: x of the last residual sweep; V_XREF: anchor of the proximal term (the point the solve started from)
// per-instance vectors of length mEcap (rows of E = [A; L; R; box rows])
enum { M_L, M_U, M_RHOV, M_YQ, M_YA, M_ZA, M_YT, M_EX, M_COEF, M_EXS, M_DY, M_RN, M_MG, M_YLV, M_HOTV, M_NUM };   // M_DY: change of ya in the last ADMM iteration; M_RN: |E_r|; M_MG: safe margins of the inactive rows (row screening); M_YLV: multipliers of the rows that left the working set in the current trial (zero otherwise)
enum { I_ST, I_STT, I_DEP, I_PRIO, I_SLOT, I_LIST, I_LIST2, I_HOTC, I_NUM };   // I_HOTC / M_HOTV: column and value of a row of E with a single non-zero (-1: any other row; wg_row_norms)   // I_DEP: row is active but linearly dependent on the rows of the factor; I_PRIO: promotion stamp (0: none); I_SLOT: slot of the row in the inverse factor (-1: none); I_LIST, I_LIST2: scratch lists of rows / slots
enum { S_R2, S_DY, S_D0, S_SV, S_W, S_NUM };   // slot-space vectors (length capS): S_SV / S_W: column of S and inv(S) times it when a row is appended

struct InstInfo {
    int mE, nfin, hasY0, setupFail, haveSolution, isSetup, nT, prioCtr;   // nT: rows of the inverse factor Ti (= rows of the working set it holds); prioCtr: promotion stamps in use (I_PRIO)
    int ndep, ns;                                                         // ndep: active rows flagged dependent (I_DEP); ns: slots of Ti in use (high-water mark, free slots inside count)
    int cNnz, kReady, rnReady, nHot;                                                   // cNnz: non-zeros of C held in compressed rows (k_compress_C), -1: C is swept as a dense matrix; kReady: L_K exists (qp_build_K); rnReady: M_RN holds the row norms of E (and I_HOTC / M_HOTV the rows with a single non-zero, nHot of them)
    double scale, sigma, spv, rhoAdmm, phiConst;
    double hist[64];     // the last nDynamicPenalty complementarity values (src/LCQProblem.cpp:1344-1375; the reference's default is 3)
    double work[6];   // exact work sums for the byte accounting: [0] rows of Et read by the corrections, [1] sum(nT*ns) over corrections (pass over Ti), [2] bytes of Ti and M moved by working-set updates and predicted corrections, [3] number of updates, [4] rows of E read by the residual sweeps (both stages), [5] triangular solves with L1
};

struct DevBatch {
    int B, n, np, nC, nComp, mA, boxcap, mEcap, capS, nblk, nd;   // nd = n + mA (dual vector, reference layout)
    int hasLbL, hasLbR;
    lcqp_options_t opt;
    double *Q, *C, *E, *Et, *F1, *FK, *S, *D1, *dscr;   // per-instance matrix blocks (S holds the inverse factor Ti, capS x capS)
    double *S2, *DS;                                     // [B][capS][capS], [B][capS/64][4096]: Cholesky factor of a working-set matrix built in one piece, its inverted diagonal blocks
    double* MM;                                          // [B][mMld][mMld]: M = Et Et', every entry of every working-set matrix (k_build_M)
    int mMld;
    int* crow;                                           // [B][capS]: row of Ti that was appended together with the slot
    int *Cp, *Ci; double* Cv; int capC;                  // C in compressed rows when it is sparse: [B][np+1], [B][capC], [B][capC]
    double *nv, *mv, *sv;                                // vector pools
    int *mi, *idx, *boxidx;
    double *lbL, *lbR;                                   // [B][nComp]
    double *yk, *y0;                                     // [B][nd]
    double *xout, *yout;                                 // [B][n], [B][nd]
    lcqp_stats_t* stats;
    InstInfo* info;
    unsigned long long* prof;   // [B][16] per-phase cycle counters (filled only by -DLCQP_PROFILE builds)
    // per-iterate tracking (options.storeSteps, src/LCQProblem.cpp:1365-1378): [B][traceCap][8] = (|statk|inf, phi, rho, alphak, obj, merit, |pk|inf, QP iterations)
    // and [B][traceCap][n] = xk; traceLen[B].  traceCap == 0: not allocated.
    double *traceS, *traceX;
    int* traceLen;
    int traceCap;
};

template <int NCH>
struct Ctx {
    static constexpr int np = 128 * NCH;
    const DevBatch* db;
    int b, n, nC, nComp, mA, mE, capS, nblk;
    double *Q, *C, *E, *Et, *F1, *FK, *S, *S2, *DS, *D1, *dscr, *MM;
    double *nv, *mv, *sv;
    int *mi, *idx, *boxidx, *crow;
    InstInfo* info;
    Lds lds;
    // work counters (uniform)
    int cAdmm, cTrials, cFact, cCorr, cSweeps;
#ifdef LCQP_PROFILE
    unsigned long long prof[16], tlast;
#endif

    __device__ __forceinline__ double* V(int k) const { return nv + (size_t)k * np; }
    __device__ __forceinline__ double* M(int k) const { return mv + (size_t)k * db->mEcap; }
    __device__ __forceinline__ int* I(int k) const { return mi + (size_t)k * db->mEcap; }
    __device__ __forceinline__ double* Sv(int k) const { return sv + (size_t)k * db->capS; }
};

template <int NCH>
__device__ __forceinline__ Ctx<NCH> make_ctx(const DevBatch& db, int b, Lds lds)
{
    Ctx<NCH> c;
    constexpr int np = 128 * NCH;
    c.db = &db; c.b = b; c.n = db.n; c.nC = db.nC; c.nComp = db.nComp; c.mA = db.mA; c.capS = db.capS; c.nblk = db.nblk;
    c.Q = db.Q + (size_t)b * np * np; c.C = db.C + (size_t)b * np * np;
    c.E = db.E + (size_t)b * db.mEcap * np; c.Et = db.Et + (size_t)b * db.mEcap * np;
    c.F1 = db.F1 + (size_t)b * np * np; c.FK = db.FK + (size_t)b * np * np;
    c.S = db.S + (size_t)b * db.capS * db.capS;
    c.MM = db.MM + (size_t)b * db.mMld * db.mMld;
    c.S2 = db.S2 + (size_t)b * db.capS * db.capS; c.DS = db.DS + (size_t)b * (db.capS / 64) * 4096;
    c.crow = db.crow + (size_t)b * db.capS;
    c.D1 = db.D1 + (size_t)b * db.nblk * 4096; c.dscr = db.dscr + (size_t)b * 4096;
    c.nv = db.nv + (size_t)b * V_NUM * np; c.mv = db.mv + (size_t)b * M_NUM * db.mEcap;
    c.sv = db.sv + (size_t)b * S_NUM * db.capS;
    c.mi = db.mi + (size_t)b * I_NUM * db.mEcap; c.idx = db.idx + (size_t)b * db.capS;
    c.boxidx = db.boxidx + (size_t)b * np;
    c.info = db.info + b;
    c.mE = uniform_i(c.info->mE);
    c.lds = lds;
    c.cAdmm = c.cTrials = c.cFact = c.cCorr = c.cSweeps = 0;
#ifdef LCQP_PROFILE
    for (int k = 0; k < 16; k++) c.prof[k] = 0;
    c.tlast = clock64();
#endif
    return c;
}

// phase buckets of the diagnostic build (tools/gpu.py phase_profile)
enum { P_LCQP = 0, P_RESID = 1, P_GRAM = 2, P_CHOL = 3, P_CORR_L1 = 4, P_CORR_ROWS = 5, P_CORR_S = 6, P_ADMM = 7, P_MISC = 8, P_DEL = 9, P_UPD_PRE = 10 };
#if defined(LCQP_PROFILE) && !defined(LCQP_PROFILE_BULK)
#define PROF(c, k) do { unsigned long long t_ = clock64(); (c).prof[k] += t_ - (c).tlast; (c).tlast = t_; } while (0)
#else
#define PROF(c, k) do { } while (0)
#endif
// -DLCQP_PROFILE -DLCQP_PROFILE_BULK: the buckets are the stages of the one-piece factor rebuild instead (ti_bulk)
#if defined(LCQP_PROFILE) && defined(LCQP_PROFILE_BULK)
#define PROFB0(c) do { (c).tlast = clock64(); } while (0)
#define PROFB(c, k) do { unsigned long long t_ = clock64(); (c).prof[k] += t_ - (c).tlast; (c).tlast = t_; } while (0)
#else
#define PROFB0(c) do { } while (0)
#define PROFB(c, k) do { } while (0)
#endif

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------
// ADMM iterations with the constant factor FK (oracle: qp_admm).  State: V_XA, M_YA, M_ZA.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ void qp_admm(Ctx<NCH>& c, const double* g, int n_it)
{
    constexpr int np = 128 * NCH;
    const lcqp_options_t& o = c.db->opt;
    const int t = tid_here(), mE = c.mE;
    const double alpha = o.admmAlpha, sigma = c.info->sigma;
    double *xa = c.V(V_XA), *rhs = c.V(V_RHS), *w = c.V(V_W);
    double *ya = c.M(M_YA), *za = c.M(M_ZA), *rhov = c.M(M_RHOV), *coef = c.M(M_COEF), *ex = c.M(M_EX), *dyl = c.M(M_DY);
    const double *l = c.M(M_L), *u = c.M(M_U);
    PROF(c, P_MISC);
    for (int it = 0; it < n_it; it++) {
        const int t = tid_here();      // per iteration (nothing derived from the thread number is carried around a loop: it would be hoisted and spilled)
        for (int r = t; r < mE; r += WG) coef[r] = rhov[r] * za[r] - ya[r];
        __syncthreads();
        // rhs = sigma*xa - g + E'(rho.z - y)
        wg_rows<NCH>(c.E, nullptr, mE, nullptr, nullptr, coef, c.lds,
                     [&](int i, double s) { rhs[i] = sigma * xa[i] - g[i] + s; });
        wg_trsv<wg_ncopy(NCH) == 2>(c.FK, np, c.nblk, rhs, true, c.lds);
        wg_trsv<wg_ncopy(NCH) == 2>(c.FK, np, c.nblk, rhs, false, c.lds);       // rhs = xt
        wg_rows<NCH>(c.E, nullptr, mE, rhs, ex, nullptr, c.lds, [](int, double) {});   // ex = E xt
        const bool last = (it == n_it - 1);      // the change of (ya, xa) in the last iteration feeds qp_certificate
        for (int r = t; r < mE; r += WG) {
            const double zr = alpha * ex[r] + (1.0 - alpha) * za[r];
            const double rv = rhov[r];
            const double yold = ya[r];
            double yn;
            if (rv > 0.0) {
                const double zn = clipd(zr + yold / rv, l[r], u[r]);
                yn = yold + rv * (zr - zn);
                za[r] = zn;
            } else {
                za[r] = zr;
                yn = 0.0;
            }
            ya[r] = yn;
            if (last) dyl[r] = yn - yold;
        }
        for (int i = t; i < np; i += WG) {
            const double xold = xa[i], xn = alpha * rhs[i] + (1.0 - alpha) * xold;
            xa[i] = xn;
            if (last) w[i] = xn - xold;
        }
        __syncthreads();
        c.cAdmm++;
    }
    PROF(c, P_ADMM);
}

// ---------------------------------------------------------------------------------------------
// OSQP's certificates from the last ADMM step (Stellato et al., Math. Prog. Comp. 12, 2020, section 3.4; oracle:
// qp_certificate), relative tolerance 1e-4: primal infeasibility from dy = y_k - y_{k-1} (M_DY), unboundedness from
// dx = x_k - x_{k-1} (V_W).  Returns the exit flag 4 (infeasible), 5 (unbounded) or 0, uniform.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ int qp_certificate(Ctx<NCH>& c, const double* g)
{
    constexpr int np = 128 * NCH;
    constexpr double eps = 1e-4;
    const int t = tid_here(), mE = c.mE;
    const double *dy = c.M(M_DY), *dx = c.V(V_W), *l = c.M(M_L), *u = c.M(M_U);
    double *tv = c.V(V_RHS), *ex = c.M(M_EX);
    const double ny = wg_maxabs(dy, mE, c.lds);
    if (ny > 1e-30) {
        double sup = 0.0; int bad = 0;
        for (int r = t; r < mE; r += WG) {
            const double d = dy[r];
            if (d > 0.0) { if (!isfinite(u[r])) bad |= (d > eps * ny); else sup += u[r] * d; }
            else if (d < 0.0) { if (!isfinite(l[r])) bad |= (-d > eps * ny); else sup += l[r] * d; }
        }
        sup = block_sum(sup, c.lds);
        bad = block_or(bad, c.lds);
        if (!bad && sup <= -eps * ny) {
            wg_rows<NCH>(c.E, nullptr, mE, nullptr, nullptr, dy, c.lds, [&](int i, double s) { tv[i] = s; });   // E' dy
            if (wg_maxabs(tv, np, c.lds) <= eps * ny) return 4;
        }
    }
    const double nx = wg_maxabs(dx, np, c.lds);
    if (nx > 1e-30) {
        const double gd = wg_dot(g, dx, np, c.lds);
        if (gd <= -eps * nx) {
            wg_symv<NCH>(c.Q, nullptr, c.n, dx, nullptr, tv, nullptr, nullptr, nullptr, c.lds);                 // Q dx
            if (wg_maxabs(tv, c.n, c.lds) <= eps * nx) {
                wg_rows<NCH>(c.E, nullptr, mE, dx, ex, nullptr, c.lds, [](int, double) {});                      // E dx
                int viol = 0;
                for (int r = t; r < mE; r += WG) {
                    const double e = ex[r];
                    viol |= (isfinite(u[r]) && e > eps * nx) || (isfinite(l[r]) && e < -eps * nx);
                }
                if (!block_or(viol, c.lds)) return 5;
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K = Q + sigma I + E' diag(rho) E and its Cholesky factor L_K, by this workgroup alone.  ADMM is the fallback of the subsolver
// (about one instance in a hundred of the synthetic workload ever runs it), so the factor is built when the first ADMM
// iteration of an instance needs it, not for every instance at setup; rho adaptation rebuilds it the same way.  The tiles are
// the ones k_build_K computed when this was a setup kernel.  Returns 1 (uniform) when a pivot is not positive.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ int qp_build_K(Ctx<NCH>& c)
{
    constexpr int np = 128 * NCH;
    const double* rhov = c.M(M_RHOV);
    const int mE = c.mE;
    const double sigma = c.info->sigma;
    for (int I = 0; I < c.nblk; I++)
        for (int J = 0; J <= I; J++) {
            double acc[4][4];
            wg_tile_tn(acc, c.E, np, 64 * I, c.E, np, 64 * J, mE, [=](int r) { return rhov[r]; }, c.lds);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int gi = 64 * I + tile_li(i, j), gj = 64 * J + tile_lj(i, j);
                    const double v = acc[i][j] + c.Q[(size_t)gi * np + gj] + (gi == gj ? sigma : 0.0);
                    c.FK[(size_t)gi * np + gj] = v;
                    // mirror only off-diagonal tiles: inside a diagonal tile (i,j) and (j,i) are both computed, and
                    // (rho*e_i)*e_j != (rho*e_j)*e_i in the last bit
                    if (I != J) c.FK[(size_t)gj * np + gi] = v;
                }
        }
    __syncthreads();
    int* fail = c.lds.ired + 15;
    if (tid_here() == 0) *fail = 0;
    __syncthreads();
    wg_chol(c.FK, np, c.nblk, c.n, 0.0, c.dscr, nullptr, fail, c.lds, 0);
    __syncthreads();
    const int failed = uniform_i(*fail);
    __syncthreads();
    if (tid_here() == 0) c.info->kReady = failed ? 0 : 1;      // a factor with a non-positive pivot is never reused by a later hot start
    return failed;
}

// ---------------------------------------------------------------------------------------------
// OSQP's rho adaptation for the fallback rounds (oracle: qp_adapt_rho): after a failed round, scale all rho_i by
//   sqrt( (|E xa - za| / max(|E xa|, |za|)) / (|Q xa + g + E'ya| / max(|Q xa|, |E'ya|, |g|)) )   (infinity norms, clipped to
// [1e-3, 1e3]) when that factor is above 5 or below 1/5, and refactorise K = Q + sigma I + E' diag(rho) E in place (the
// work k_build_K and k_factor do at setup, here by this workgroup alone).  Returns 1 (uniform) when rho changed, 0 when it stays,
// -1 when neither the new nor the old rho gives a positive definite K.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ int qp_adapt_rho(Ctx<NCH>& c, const double* g)
{
    constexpr int np = 128 * NCH;
    const int t = tid_here(), mE = c.mE;
    double *xa = c.V(V_XA), *tq = c.V(V_RHS), *ty = c.V(V_TMP);
    double *ya = c.M(M_YA), *za = c.M(M_ZA), *rhov = c.M(M_RHOV), *ex = c.M(M_EX);
    wg_symv<NCH>(c.Q, nullptr, c.n, xa, nullptr, tq, nullptr, nullptr, nullptr, c.lds);                 // Q xa
    wg_rows<NCH>(c.E, nullptr, mE, xa, ex, ya, c.lds, [&](int i, double s) { ty[i] = s; });            // E xa, E'ya
    double rp = 0.0, nax = 0.0, nz = 0.0;
    for (int r = t; r < mE; r += WG) {
        rp = fmax(rp, fabs(ex[r] - za[r])); nax = fmax(nax, fabs(ex[r])); nz = fmax(nz, fabs(za[r]));
    }
    double rd = 0.0, nq = 0.0, naty = 0.0, gm = 0.0;
    for (int i = t; i < np; i += WG) {
        rd = fmax(rd, fabs(tq[i] + g[i] + ty[i])); nq = fmax(nq, fabs(tq[i])); naty = fmax(naty, fabs(ty[i])); gm = fmax(gm, fabs(g[i]));
    }
    rp = block_max(rp, c.lds); nax = block_max(nax, c.lds); nz = block_max(nz, c.lds);
    rd = block_max(rd, c.lds); nq = block_max(nq, c.lds); naty = block_max(naty, c.lds); gm = block_max(gm, c.lds);
    const double num = rp / fmax(fmax(nax, nz), 1e-30), den = rd / fmax(fmax(fmax(nq, naty), gm), 1e-30);
    double fac = sqrt(num / fmax(den, 1e-30));
    fac = fmin(fmax(fac, 1e-3), 1e3);
    if (!(fac > 5.0 || fac < 0.2)) return 0;
    for (int r = t; r < mE; r += WG) rhov[r] *= fac;
    if (t == 0) c.info->rhoAdmm *= fac;
    __syncthreads();
    // one call site for the factorisation: a second pass with the old rho when a pivot is not positive at the new one
    int failed = 1, restored = 0;
    for (int attempt = 0; attempt < 2 && failed; attempt++) {
        c.cFact++;
        failed = qp_build_K<NCH>(c);
        if (failed && attempt == 0) {
            for (int r = t; r < mE; r += WG) rhov[r] /= fac;
            if (t == 0) c.info->rhoAdmm /= fac;
            __syncthreads();
            restored = 1;
        }
    }
    return failed ? -1 : (restored ? 0 : 1);
}

// ---------------------------------------------------------------------------------------------
// Dependent-row rules of the subsolver (ROBUST = true in every kernel since round 2: k_lcqp_run, k_qp_solve; oracle: q->robust).
// (1) Rows the last factorisation of S flagged as linearly dependent on the rows before them: the correction neither
//     moved their multipliers nor enforced their equations.  Strictly inside its bound: the row is not active.
//     Violated: it must be active, so it is promoted to the front of the list and another row becomes the dependent
//     one.  Returns bit 0 (a row left) | bit 1 (a row was promoted), per thread.
// (2) The promoted part of the ordered active list: latest promotion first, ascending row index within one promotion.
//     Every promoted active row counts the rows that precede it.  Returns (number of promoted active rows) << 1 |
//     (the list differs from the stored one), uniform.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int polish_dependent_rows(int* st, const int* dep, int* prio, const double* ex, const double* l,
                                                     const double* u, int mE, double feasTol, int stamp)
{
    int chg = 0;
    for (int r = tid_here(); r < mE; r += WG) {
        const int s = st[r];
        if (s == ST_INACT || !dep[r]) continue;
        const double e = ex[r], ftol = feasTol * (1.0 + fabs(e));
        bool viol, inside = false;
        if (s == ST_LOWER) { viol = e < l[r] - ftol; inside = e > l[r] + ftol; }
        else if (s == ST_UPPER) { viol = e > u[r] + ftol; inside = e < u[r] - ftol; }
        else viol = fabs(e - l[r]) > ftol;
        if (inside) { st[r] = ST_INACT; chg |= 1; }
        else if (viol) { prio[r] = stamp; chg |= 2; }
    }
    return chg;
}

__device__ __forceinline__ int polish_promoted_list(const int* st, const int* prio, int* idx, int mE, int r0, int r1e, int cap, Lds lds)
{
    int mine = 0, differs = 0;
    for (int r = r0; r < r1e; r++) mine += (st[r] != ST_INACT && prio[r] > 0);
    const int nprom = block_sum_i(mine, lds);
    if (nprom <= cap)
        for (int r = r0; r < r1e; r++) {
            const int pr = prio[r];
            if (st[r] == ST_INACT || pr == 0) continue;
            int rank = 0;
            for (int q = 0; q < mE; q++) {
                const int pq = prio[q];
                rank += (pq > 0 && st[q] != ST_INACT && (pq > pr || (pq == pr && q < r)));
            }
            differs |= (idx[rank] != r);
            idx[rank] = r;
        }
    differs = block_or(differs, lds);
    return (nprom << 1) | differs;
}

// ---------------------------------------------------------------------------------------------
// Inverse factor Ti of the working-set matrix S_W = Et_W Et_W' (oracle: ti_reset / ti_apply / ti_append / ti_delete).
// The reference's subsolver UPDATES its factors when the working set changes on a hot start (qp.hotstart,
// src/SubsolverQPOASES.cpp:158); so does this one: Ti (nT rows x ns slots, row-major in c.S, ld = capS) with Ti'Ti = inv(S_W)
// gains a row when a constraint enters and loses one by a sweep of row rotations when a constraint leaves.  S dy = t is then one
// fused pass over Ti: dy = Ti'(Ti t) -- no triangular solves, no sequential chains.
//   slot_row = c.idx[capS] (row of E held by a slot, -1: free), row_slot = I_SLOT[mE], crow = c.crow[capS] (the row of Ti that
//   was appended together with the slot: column s is zero in the rows above crow[s]).  Entries of S come from M = Et Et'.
// Slot-space vectors (S_* pool) have capS entries, zero on free slots.
// ---------------------------------------------------------------------------------------------
constexpr int TI_FAST_CHUNKS = 4;    // the fused apply keeps up to 256 slots in registers (lane l: slots l, l+64, l+128, l+192)

template <int NCH>
__device__ __forceinline__ void ti_reset(Ctx<NCH>& c, int& nT, int& ns)
{
    int* rslot = c.I(I_SLOT);
    for (int r = tid_here(); r < c.mE; r += WG) rslot[r] = -1;
    for (int a = tid_here(); a < c.capS; a += WG) c.idx[a] = -1;
    nT = 0; ns = 0;
    __syncthreads();
}

// The fused pass for up to 64*NK slots: a wave takes rows j = w, w+4, ..., keeps D rows (D*NK loads of 512 bytes) in flight, and
// for each row forms u = row.t (one wavefront reduction) and accumulates u*row -- the row is read once.
// Column s of Ti is zero in the rows above crow[s] (Ti is triangular in creation order), and a lane does not load what it knows to be zero:
// half the bytes of the pass (-DLCQP_TI_FULL_ROWS reads the full rows: cross-check).
template <int NK, int D>
__device__ __forceinline__ void ti_apply_fast(const double* __restrict__ Ti, int ld, const double* tv, double* out, int nT, int ns, double* red, const int* crow, const int* slotRow)
{
    const int l = lane_id(), w = wave_id(), t = tid_here();
    double tr[NK], acc[NK];
    int first[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int sl = 64 * k + l; tr[k] = (sl < ns) ? tv[sl] : 0.0; acc[k] = 0.0;
#ifdef LCQP_TI_FULL_ROWS
        first[k] = (sl < ns) ? 0 : (1 << 30);
#else
        first[k] = (sl < ns && slotRow[sl] >= 0) ? crow[sl] : (1 << 30);      // (a free slot: its column is zero everywhere)
#endif
    }
    for (int j0 = w; j0 < nT; j0 += NWAVE * D) {
        double rv[D][NK];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int j = j0 + NWAVE * d;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int sl = 64 * k + l;
                rv[d][k] = (j < nT && j >= first[k]) ? Ti[(size_t)j * ld + sl] : 0.0;
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            double dsum = 0.0;
#pragma unroll
            for (int k = 0; k < NK; k++) dsum += rv[d][k] * tr[k];
            const double u = wave_sum(dsum);
#pragma unroll
            for (int k = 0; k < NK; k++) acc[k] += rv[d][k] * u;
        }
    }
    __syncthreads();               // tv fully read (out may alias it), arena free
#pragma unroll
    for (int k = 0; k < NK; k++) red[w * 256 + 64 * k + l] = acc[k];
    __syncthreads();
    for (int sl = t; sl < 64 * NK; sl += WG) out[sl] = (sl < ns) ? (red[sl] + red[256 + sl]) + (red[512 + sl] + red[768 + sl]) : 0.0;
    __syncthreads();
}

// out = Ti' (Ti tv) over the slots [0, ns); tv and out: global, capS entries, tv zero on free slots.  out may alias tv.
template <int NCH>
__device__ __forceinline__ void ti_apply(Ctx<NCH>& c, const double* tv, double* out, int nT, int ns)
{
    const int ld = c.capS, l = lane_id(), w = wave_id(), t = tid_here();
    const int nk = (ns + 63) >> 6;
    const double* Ti = c.S;
    if (nk <= TI_FAST_CHUNKS) {
        double* red = c.lds.arena;     // 4 waves x 256 partial sums
        if (nk <= 1) ti_apply_fast<1, 16>(Ti, ld, tv, out, nT, ns, red, c.crow, c.idx);
        else if (nk == 2) ti_apply_fast<2, 8>(Ti, ld, tv, out, nT, ns, red, c.crow, c.idx);
        else if (nk == 3) ti_apply_fast<3, 6>(Ti, ld, tv, out, nT, ns, red, c.crow, c.idx);
        else ti_apply_fast<4, 4>(Ti, ld, tv, out, nT, ns, red, c.crow, c.idx);
    } else {
        // wide working sets (more than 256 slots; only after an ADMM round guessed many rows): two passes, u = Ti tv staged in LDS
        constexpr int MAXK = (max_active(NCH) + 63) / 64;
        double* ts = c.lds.arena;                 // 64 nk entries
        double* us = c.lds.arena + 64 * nk;       // nT entries   (64 nk + nT <= 2 capS; the partial sums below reuse the arena)
        for (int sl = t; sl < 64 * nk; sl += WG) ts[sl] = (sl < ns) ? tv[sl] : 0.0;
        __syncthreads();
        for (int j = w; j < nT; j += NWAVE) {
            double dsum = 0.0;
#pragma unroll 4
            for (int k = 0; k < nk; k++) { const int sl = 64 * k + l; dsum += (sl < ns ? Ti[(size_t)j * ld + sl] : 0.0) * ts[sl]; }
            dsum = wave_sum(dsum);
            if (l == 0) us[j] = dsum;
        }
        __syncthreads();
        double acc[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; k++) acc[k] = 0.0;
        for (int j = w; j < nT; j += NWAVE) {
            const double u = us[j];
#pragma unroll
            for (int k = 0; k < MAXK; k++) { const int sl = 64 * k + l; if (k < nk && sl < ns) acc[k] += Ti[(size_t)j * ld + sl] * u; }
        }
        __syncthreads();                          // ts / us consumed
        double* red = c.lds.arena;                // 4 waves x 64 nk partial sums (<= 4 * 64 * MAXK = arena size of the np = 512 build)
#pragma unroll
        for (int k = 0; k < MAXK; k++) if (k < nk) red[w * 64 * nk + 64 * k + l] = acc[k];
        __syncthreads();
        for (int sl = t; sl < 64 * nk; sl += WG)
            out[sl] = (sl < ns) ? (red[sl] + red[64 * nk + sl]) + (red[2 * 64 * nk + sl] + red[3 * 64 * nk + sl]) : 0.0;
        __syncthreads();
    }
}

// The inverse factor built in one piece for the ordered row list idx[0..na) (slot a = list position a): S_W gathered from M,
// blocked Cholesky L (fp64 MFMA tiles, safeguarded pivots, wg_chol), then the blocked inverse Ti = inv(L):
//   Ti_JJ = D_J (the inverted diagonal blocks wg_chol leaves),  Ti_IJ = -D_I * sum_{K=J}^{I-1} L_IK Ti_KJ   (I > J).
// Used when the factor is empty or most of it would change (oracle: ti_reset followed by appends in list order -- the same
// matrix up to rounding, since inv(S_W) = inv(L)' inv(L)).  Rows the factorisation flags as dependent lose their slot and are
// marked in I_DEP.  Returns the number of flagged rows (uniform).
template <int NCH>
__device__ __forceinline__ int ti_bulk(Ctx<NCH>& c, int na, double tau, int& nT, int& ns)
{
    const int ld = c.capS, t = tid_here(), mMld = c.db->mMld;
    const int nb = (na + 63) >> 6, nn = 64 * nb;
    int *idx = c.idx, *rslot = c.I(I_SLOT), *dep = c.I(I_DEP);
    double *F = c.S2, *Ti = c.S, *DS = c.DS;
    nT = na; ns = na;
    if (na == 0) return 0;
    PROFB0(c);
    // element loops with eight gathers in flight per thread (a load - store - load chain pays a memory round trip per element)
    {
        const int tot = nn * nn;
        int i = t / nn, j = t - i * nn;
        for (int e0 = t; e0 < tot; e0 += 8 * WG) {
            double v[8]; int ii[8], jj[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                ii[u] = i; jj[u] = j;
                const bool in = (e0 + u * WG < tot) && (j <= i);
                v[u] = (i == j) ? 1.0 : 0.0;
                if (in && i < na) { const int ri = idx[i], rj = idx[j]; v[u] = c.MM[(size_t)max(ri, rj) * mMld + min(ri, rj)]; }      // M holds its lower triangle
                if (!in) ii[u] = -1;
                j += WG; while (j >= nn) { j -= nn; i++; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (ii[u] >= 0) F[(size_t)ii[u] * ld + jj[u]] = v[u];
        }
    }
    __syncthreads();
    PROFB(c, 0);
    wg_chol(F, ld, nb, na, tau, DS, c.Sv(S_D0), nullptr, c.lds, 4096);
    PROFB(c, 1);
    // Ti: zero, diagonal blocks D_J (dense lower copies in DS)
    {
        const int tot = nn * nn;
        int i = t / nn, j = t - i * nn;
        for (int e0 = t; e0 < tot; e0 += 8 * WG) {
            double v[8]; int ii[8], jj[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                ii[u] = (e0 + u * WG < tot && (j >> 6) >= (i >> 6)) ? i : -1; jj[u] = j;      // the blocks below the diagonal are written by the products that follow
                v[u] = 0.0;
                if (ii[u] >= 0 && (i >> 6) == (j >> 6)) v[u] = DS[(size_t)(i >> 6) * 4096 + (i & 63) * 64 + (j & 63)];
                j += WG; while (j >= nn) { j -= nn; i++; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (ii[u] >= 0) Ti[(size_t)ii[u] * ld + jj[u]] = v[u];
        }
    }
    __syncthreads();
    PROFB(c, 2);
    auto ident = [](int r) { return (long)r; };
    for (int J = 0; J + 1 < nb; J++)
        for (int I = J + 1; I < nb; I++) {
            double acc[4][4];
            // X = sum_K L_IK Ti_KJ: rows 64J .. 64I-1 of the upper part of F hold L_IK' , the same rows of Ti hold Ti_KJ
            wg_tile_tn(acc, F + (size_t)(64 * J) * ld, ld, 64 * I, Ti + (size_t)(64 * J) * ld, ld, 64 * J, 64 * (I - J), [](int) { return 1.0; }, c.lds);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) c.dscr[tile_lj(i, j) * 64 + tile_li(i, j)] = acc[i][j];       // X'
            __syncthreads();
            double acc2[4][4];
            wg_tile_nt(acc2, DS + (size_t)I * 4096, 64, ident, c.dscr, 64, ident, 64, c.lds);              // D_I X
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) Ti[(size_t)(64 * I + tile_li(i, j)) * ld + 64 * J + tile_lj(i, j)] = -acc2[i][j];
            __syncthreads();
        }
    PROFB(c, 3);
    // slots; rows whose pivot was flagged (stored inverse diagonal 1/1e150) are dependent on the rows before them
    int nflag = 0;
    for (int a = t; a < na; a += WG) {
        const bool flagged = DS[(size_t)(a >> 6) * 4096 + (a & 63) * 65] < 1e-100;
        const int r = idx[a];
        c.crow[a] = a;
        if (flagged) { nflag++; dep[r] = 1; idx[a] = -1; } else { rslot[r] = a; dep[r] = 0; }
    }
    nflag = block_sum_i(nflag, c.lds);
    if (nflag > 0) {
        // the flagged rows and columns of Ti are ~1e-150 by construction: clear them, the slots are free
        for (int e = t; e < na * na; e += WG) {
            const int i = e / na, j = e - i * na;
            if (idx[i] < 0 || idx[j] < 0) Ti[(size_t)i * ld + j] = 0.0;
        }
        __syncthreads();
    }
    if (t == 0) c.info->work[2] += 8.0 * 3.0 * (double)na * na;      // S gathered, factor written, inverse written (lower halves: about na^2 / 2 each, read and written)
    PROFB(c, 4);
    return nflag;
}

// Row r enters the working set.  With s = S_W,r (entries of M), w = inv(S_W) s and d2 = S_rr - s'w (the Schur complement, i.e.
// the squared Cholesky pivot of the row behind the rows of W) the new last row of Ti is [-w'/delta, 1/delta], delta = sqrt(d2).
// d2 <= tau S_rr: the row is linearly dependent on W and stays out of the factor (returns 0).  Returns 1 when appended, -1 when
// the factor is full.  Uniform.
template <int NCH>
__device__ __forceinline__ int ti_append(Ctx<NCH>& c, int r, double tau, int capNa, int& nT, int& ns)
{
    const int ld = c.capS, t = tid_here();
    double *sv = c.Sv(S_SV), *wv = c.Sv(S_W);
    int *idx = c.idx, *rslot = c.I(I_SLOT);
    double* Ti = c.S;
    const int mMld = c.db->mMld;
    const int nsp = 64 * ((ns + 63) >> 6);
    int freeSlot = 1 << 30;
    for (int sl = t; sl < nsp; sl += WG) {
        const int rr = (sl < ns) ? idx[sl] : -2;
        sv[sl] = (rr >= 0) ? c.MM[(size_t)max(r, rr) * mMld + min(r, rr)] : 0.0;      // M holds its lower triangle
        if (rr == -1) freeSlot = min(freeSlot, sl);
    }
    __syncthreads();
    double d2 = c.MM[(size_t)r * mMld + r];
    const double srr = uniform_d(d2);
    if (nT > 0) {
        ti_apply<NCH>(c, sv, wv, nT, ns);
        double part = 0.0;
        for (int sl = t; sl < ns; sl += WG) part += sv[sl] * wv[sl];
        d2 = srr - block_sum(part, c.lds);
    } else {
        d2 = srr;
    }
    if (t == 0) c.info->work[2] += 8.0 * (((ns <= 64 * TI_FAST_CHUNKS) ? 0.5 * (double)nT * (nT + 1) : (double)nT * ns) + 3.0 * ns);
    int snew = -block_max((double)(-freeSlot), c.lds) + 0.5;      // smallest free slot (exact in double), 2^30: none
    if (!(d2 > tau * srr) || !(d2 > 0.0)) return 0;
    if (snew >= ns) {
        if (ns >= capNa || ns >= c.capS) return -1;
        snew = ns;
        ns++;
        for (int j = t; j < nT; j += WG) Ti[(size_t)j * ld + snew] = 0.0;       // a fresh column
    }
    if (nT >= capNa) return -1;
    const double delta = sqrt(d2);
    double* row = Ti + (size_t)nT * ld;
    for (int sl = t; sl < ns; sl += WG) row[sl] = (sl == snew) ? 1.0 / delta : ((idx[sl] >= 0) ? -wv[sl] / delta : 0.0);
    __syncthreads();
    if (t == 0) { idx[snew] = r; rslot[r] = snew; c.crow[snew] = nT; }
    nT++;
    __syncthreads();
    return 1;
}

// The row held by slot p leaves: rotations of neighbouring rows of Ti, from row crow[p] downwards, collect column p in the last
// row, which is dropped; Ti'Ti then is the inverse of S without row and column p.  The rotations follow from column p alone
// (rho_j^2 = sum_{i<=j} Ti[i][p]^2), so they are formed first (a prefix sum) and the rows are then swept column by column:
// thread = column, a lane-local recurrence over the rows, sixteen rows in flight.
template <int NCH>
__device__ __forceinline__ void ti_delete(Ctx<NCH>& c, int p, int& nT, int& ns)
{
    const int ld = c.capS, t = tid_here(), l = lane_id(), w = wave_id();
    int *idx = c.idx, *rslot = c.I(I_SLOT), *crow = c.crow;
    double* Ti = c.S;
    const int i0 = uniform_i(crow[p]), m = nT - i0;
    double* b = c.lds.arena;            // m entries each: column p, cosines, sines   (3 capS <= arena)
    double* cs = b + c.capS;
    double* sn = cs + c.capS;
    for (int j = t; j < m; j += WG) b[j] = Ti[(size_t)(i0 + j) * ld + p];
    __syncthreads();
    {
        const int per = (m + WG - 1) / WG, j0 = t * per, j1 = min(m, j0 + per);
        double loc = 0.0;
        for (int j = j0; j < j1; j++) loc += b[j] * b[j];
        double incl = loc;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const double v = __shfl_up(incl, ofs, 64); if (l >= ofs) incl += v; }
        if (l == 63) c.lds.red[8 + w] = incl;
        __syncthreads();
        double run = incl - loc;
        for (int ww = 0; ww < w; ww++) run += c.lds.red[8 + ww];
        for (int j = j0; j < j1; j++) {
            const double rp = (j == 1) ? b[0] : sqrt(run);      // rho_0 = Ti[i0][p] carries its sign, rho_j > 0 afterwards
            run += b[j] * b[j];
            const double rn = sqrt(run);
            const bool ok = rn > 0.0;
            cs[j] = ok ? b[j] / rn : 1.0;
            sn[j] = ok ? -rp / rn : 0.0;
        }
        __syncthreads();
    }
    for (int col = t; col < ns; col += WG) {
        // column col is zero above row crow[col] (everywhere for a free slot): rotations of zeros are skipped, in whole blocks of 16 rows
#ifdef LCQP_TI_FULL_ROWS
        const int cr = i0;
#else
        const int cr = (col == p) ? i0 : ((idx[col] >= 0) ? crow[col] : (1 << 30));
#endif
        double carry = (cr <= i0) ? Ti[(size_t)i0 * ld + col] : 0.0;
        const int jz = max(1, min(cr - i0, m));                 // rows i0 + 1 ... i0 + jz - 1 of this column are zero, and so is the carry
        for (int j = 1 + 16 * ((jz - 1) / 16); j < m; j += 16) {
            double rv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) rv[q] = (j + q < m) ? Ti[(size_t)(i0 + j + q) * ld + col] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (j + q < m) {
                    const double cc = cs[j + q], ss = sn[j + q];
                    const double o = cc * carry + ss * rv[q];
                    carry = cc * rv[q] - ss * carry;
                    Ti[(size_t)(i0 + j + q - 1) * ld + col] = (col == p) ? 0.0 : o;
                }
        }
    }
    int top = 0;
    for (int sl = t; sl < ns; sl += WG) {
        const int rr = idx[sl];
        if (sl != p && rr >= 0) { if (crow[sl] > i0) crow[sl]--; top = max(top, sl + 1); }
    }
    const int rOut = uniform_i(idx[p]);
    __syncthreads();
    if (t == 0) { idx[p] = -1; rslot[rOut] = -1; c.info->work[2] += 16.0 * (double)m * ns; }
    ns = (int)(block_max((double)top, c.lds) + 0.5);     // free slots at the end are given back
    nT--;
}

// ---------------------------------------------------------------------------------------------
// Primal-dual active-set polish in correction form (oracle: qp_polish).
// In/out: x = V_XT, multipliers M_YT (OSQP sign, zero on inactive rows), active set I_STT.
// Returns 1 (uniform) on a verified KKT point.
//
// Round 3: a trial does not evaluate the whole KKT residual before it knows whether it needs it.  After a correction the state is
// known up to rounding: the stationarity residual is sigma_p dx (taken as 0) and the rows in the factor sit on their bounds.  A trial:
//   (a) rows whose multiplier has the wrong sign leave (multipliers remembered in M_YLV);
//   (b) STAGE 1: E_r x of the inactive rows the screening cannot rule out (and of active rows flagged dependent); violated rows enter;
//   (c) nothing changed: STAGE 2, the true residual -- one sweep over Q and one over the active rows of E; only these true residuals
//       accept a point (accuracy is what it was), else a full correction with them follows;
//   (d) the set changed: the factor follows, then a PREDICTED correction: r1 = sum over the rows that left of y_r E_r, so
//       c = L1^-1 r1 = sum y_r Et_r (no forward solve), Et_W c = sum y_r M[r][W] (entries of M = Et Et', no pass over Et_W), r2 = 0 on
//       the rows that were in the factor and b - E_r x on those that entered; dx = L1^-T (c - Et_W' dy): one pass over Et_W, one
//       backward solve.  Intermediate trials read neither Q nor the active rows of E.
// What is solved is the PROXIMAL QP  min 1/2 x'Qx + g'x + sigma_p/2 |x - xref|^2,  xref = the point the solve started from (V_XREF) and
// sigma_p the weight the constant factor L1 = chol(Q + sigma_p I) carries anyway: the corrections are exact Newton steps of the problem
// whose residual is tested (the predicted residual after a correction is exactly zero), and the QP has ONE solution also when Q has
// flat directions (the minimiser nearest xref up to O(sigma_p)), whatever path -- working sets, ADMM rounds -- leads there.  It is
// returned when it also satisfies the QP as given to the tolerance (always so for sigma_p = 1e-12 max|Q_ii|); otherwise xref moves to
// it and the iteration continues: the proximal-point method, every step of which has a unique solution.
// Row screening (round 2): an inactive row that lay m_r inside its (tolerance-widened) bounds when it was last evaluated cannot be
// violated while |E_r| * (sum of |x - x_last sweep| since) < m_r (Cauchy-Schwarz), so it is not read.  The decisions are those of a
// sweep over all rows.  Margins are void after anything but this routine wrote x-dependent state (cold entry: every row is read).
// reuse != 0 (hot start from the last verified solution): the first trial needs no sweep, because
// r1 = r1_last + (g_last - g) and E x = ex_last hold exactly for an unchanged (x, y).
// ---------------------------------------------------------------------------------------------
// Round 5 (oracle: the same two additions, the same arithmetic):
//   * ACTIVE ROWS TO THEIR ROUNDING FLOOR.  runSolver ends on phi < complementarityTolerance = 1e3 eps (src/LCQProblem.cpp:511-534,
//     src/Options.cpp:297), a sum of products in which one factor is the residual of an active row of this QP.  A point that passes the
//     residual tests is accepted only when every row of the factor also holds to 16 eps (|b_r| + |E_r| |x|), the rounding of a computed
//     E_r x; else one more correction (iterative refinement, at most two per polish) comes first.  On well-conditioned QPs the rows are
//     at that floor after every correction: nothing changes (the synthetic workload is bit-identical).
//   * damp != 0 (the rounds after DAMP_ROUND failed ones): ONE change of the working set per trial -- the row with the largest
//     wrong-signed multiplier leaves, else the most violated row enters -- and 4 n + 32 more trials: the full primal-dual update
//     thrashes on LP-like QPs (singular Hessian, |g| ~ 1e7 at the end of a penalty homotopy).
constexpr int DAMP_ROUND = 3;
template <int NCH, bool ROBUST, bool LR>
__device__ __forceinline__ int qp_polish(Ctx<NCH>& c, const double* g, int reuse, double ytol, double rtolG, int damp)
{
    constexpr int np = 128 * NCH;
    // every size sweeps through row lists (round 2 kept the plain sweep at np = 1024 because the list sweep returned wrong residuals in that
    // instantiation of the old kernel; in the round-3 kernel it is correct on every np = 1024 shape and on its own:
    // tests/test_gpu_parity.py::test_row_list_sweep); -DLCQP_PLAIN_SWEEPS keeps the plain sweeps as a cross-check
#ifdef LCQP_PLAIN_SWEEPS
    constexpr bool LISTS = false;
#else
    constexpr bool LISTS = true;
#endif
    const lcqp_options_t& o = c.db->opt;
    const int t = tid_here(), mE = c.mE, capS = c.capS;
    double *x = c.V(V_XT), *r1 = c.V(V_R1), *cv = c.V(V_C), *du = c.V(V_DU), *qx = c.V(V_TMP), *xs = c.V(V_XS);
    double* xref = c.V(V_XREF);
    const double spv = uniform_d(c.info->spv);
    // LR: multipliers, E x, margins and status of the rows live in LDS for the whole launch (lcqp_wg.hpp, LDS_ROWS_*)
    double* yt = LR ? c.lds.arena + LDS_ROWS_OFF : c.M(M_YT);
    double* ex = LR ? c.lds.arena + LDS_ROWS_OFF + LDS_ROWS_MAX : c.M(M_EX);
    double* mg = LR ? c.lds.arena + LDS_ROWS_OFF + 2 * LDS_ROWS_MAX : c.M(M_MG);
    int* st = LR ? reinterpret_cast<int*>(c.lds.arena + LDS_ROWS_OFF + 3 * LDS_ROWS_MAX) : c.I(I_STT);
    double *ylv = c.M(M_YLV), *rn = c.M(M_RN);
    int* hotc = c.I(I_HOTC);
    const double* hotv = c.M(M_HOTV);
    const double *l = c.M(M_L), *u = c.M(M_U);
    int *dep = c.I(I_DEP), *prio = c.I(I_PRIO), *list = c.I(I_LIST), *rslot = c.I(I_SLOT);
    double *r2 = c.Sv(S_R2), *dy = c.Sv(S_DY);
    int* idx = c.idx;
    // ytol = feasTol (1 + |g|_inf), rtolG = resTol (1 + |g|_inf): formed by the caller, once per QP (long-lived uniform doubles are held in scalar registers: uniform_d)
    const double shrinkF = uniform_d(1.0 + 1e-6 + o.feasTol);
    int na = 0, nsl = 0, fact_valid = 0;
    int prioCtr = ROBUST ? uniform_i(c.info->prioCtr) : 0;
    const int capNa = min(min(min(max(2 * c.n, 64), mE), capS), max_active(NCH));   // room for the degenerate vertices of small problems

    // |x - x_last sweep| for the margins; x becomes the x of the last sweep.  Returns the shrink of the margins per unit row norm.
    double xnrm = 0.0;      // |x|_2 at the top of the current trial (scale of the rounding floor of E_r x)
    auto sweep_distance = [&]() -> double {
        double d2 = 0.0, x2 = 0.0;
        for (int i = tid_here(); i < c.n; i += WG) { const double xv = x[i], dd = xv - xs[i]; d2 += dd * dd; x2 += xv * xv; xs[i] = xv; }     // the n variables, not the padding
        // |E_r (x - x_last)| <= |E_r| |x - x_last|; the factor covers the tolerance that moves with E_r x (feasTol (1 + |E_r x|)),
        // the second term the rounding of a computed E_r x (~ eps |E_r| |x|)
        double sd2, sx2;
        block_sum2(d2, x2, sd2, sx2, c.lds);
        xnrm = uniform_d(sqrt(sx2));
        return uniform_d(sqrt(sd2) * shrinkF + 1e-13 * xnrm);
    };

    // a polish that gives up leaves no multipliers of leaving rows behind (M_YLV is zero between trials)
    auto give_up = [&]() -> int {
        for (int r = tid_here(); r < mE; r += WG) ylv[r] = 0.0;
        __syncthreads();
        return 0;
    };

    int capOn = 0;      // the cap on entering rows applies to polishes that start from an empty working set (oracle: cap_on)
    int nrefine = 0;    // refinement corrections taken for the active rows alone
    const int maxTrials = damp ? o.maxTrials + 4 * c.n + 32 : o.maxTrials;
    auto violation = [&](int r) -> double {
        if (st[r] != ST_INACT) return 0.0;
        const double e = ex[r], ftol = o.feasTol * (1.0 + fabs(e));
        return (e < l[r] - ftol) ? l[r] - e : ((e > u[r] + ftol) ? e - u[r] : 0.0);
    };
    for (int trial = 0; trial < maxTrials; trial++) {
        const int t = tid_here();      // per trial: nothing derived from the thread number is carried around the loop (it would be hoisted and spilled)
        c.cTrials++;
        int changed = 0, nlv = 0, have_true = 0, need_true = 0;
        const bool cold = (trial == 0 && !reuse);
        PROF(c, P_MISC);
        if (trial == 0) {
            if (reuse) {
                const double *r1s = c.V(V_R1S), *gs0 = c.V(V_GS), *exs = c.M(M_EXS);
                for (int i = t; i < np; i += WG) r1[i] = (r1s[i] + (gs0[i] - g[i])) - spv * (x[i] - xref[i]);
                wg_map<4>(mE, [&](int r) { return exs[r]; }, [&](int r, double v) { ex[r] = v; });
                __syncthreads();
                have_true = 1;      // the first trial only corrects: the working set it was handed stays
            } else {
                // cold entry: the whole residual, every row of E, fresh margins.  (Whatever ran before -- ADMM, rho adaptation, a solve
                // with other bounds -- may have left M_EX and the margins in any state.)
                if (!uniform_i(c.info->rnReady)) {
                    wg_row_norms<NCH>(c.E, mE, rn, hotc, c.M(M_HOTV));
                    int nh = 0;
                    for (int r = t; r < mE; r += WG) nh += (hotc[r] >= 0);
                    nh = block_sum_i(nh, c.lds);
                    if (t == 0) { c.info->rnReady = 1; c.info->nHot = nh; }
                }
                int cntA = 0;
                wg_map<4>(mE, [&](int r) { return MapID{st[r], yt[r]}; },
                          [&](int r, MapID v) { if (ROBUST && v.s == ST_INACT && v.a != 0.0) yt[r] = 0.0; ylv[r] = 0.0; cntA += (v.s != ST_INACT); });
                capOn = (block_sum_i(cntA, c.lds) == 0);
                (void)sweep_distance();
                need_true = 1;
            }
        } else {
            // (a) leaving rows; margins of the inactive rows shrink by |E_r| * |x - x_last sweep|
            const double dl = sweep_distance();
            int cntLv = 0;
            int rLeave = -1;      // damped: the one row that may leave (largest wrong-signed multiplier, lowest index among equals)
            if (damp) {
                double vb = 0.0;
                for (int r = t; r < mE; r += WG) { const int s = st[r]; const double yv = yt[r]; if ((s == ST_LOWER && yv > ytol) || (s == ST_UPPER && yv < -ytol)) vb = fmax(vb, fabs(yv)); }
                const double ylvmax = block_max(vb, c.lds);
                int rb = 1 << 30;
                for (int r = t; r < mE; r += WG) { const int s = st[r]; const double yv = yt[r]; if (((s == ST_LOWER && yv > ytol) || (s == ST_UPPER && yv < -ytol)) && fabs(yv) >= ylvmax) rb = min(rb, r); }
                rLeave = (int)(-block_max((double)(-rb), c.lds) + 0.5);
            }
            wg_map<4>(mE, [&](int r) { return MapID3{st[r], yt[r], mg[r], rn[r]}; },
                      [&](int r, MapID3 v) {
                          int s = v.s;
                          if (((s == ST_LOWER && v.a > ytol) || (s == ST_UPPER && v.a < -ytol)) && (!damp || r == rLeave)) { ylv[r] = v.a; yt[r] = 0.0; st[r] = ST_INACT; s = ST_INACT; cntLv++; }
                          if (LISTS) mg[r] = (s != ST_INACT) ? -1.0 : v.b - v.c * dl;
                      });
            nlv = block_sum_i(cntLv, c.lds);
            changed = nlv > 0;
            // (b) stage 1: E_r x of the inactive rows the margins cannot rule out, and of the active rows flagged dependent
            const bool depRows = ROBUST && uniform_i(c.info->ndep) > 0;
            int nread;
            if (LISTS) {
#ifdef LCQP_SCREEN_ALL      // test hook: every inactive row is read (the list machinery without the screening)
                nread = wg_compact(mE, [&](int r) { return st[r] == ST_INACT || (depRows && dep[r]); }, list, c.lds);
#else
                nread = wg_compact(mE, [&](int r) { return (st[r] == ST_INACT) ? !(mg[r] > 1e-10) : (depRows && dep[r] != 0); }, list, c.lds);
#endif
                wg_rows<NCH, true>(c.E, list, nread, x, ex, nullptr, c.lds, [](int, double) {}, hotc, hotv);
            } else {
                nread = mE;
                wg_rows<NCH>(c.E, nullptr, mE, x, ex, nullptr, c.lds, [](int, double) {}, hotc, hotv);
            }
            // Entering rows are capped (oracle: qp_polish, same arithmetic): when more than max(n/8, 16) inactive rows are violated -- a cold
            // start, where every violated row would enter at once, overshoot and oscillate for eight to ten trials with a factor rebuild
            // each -- only those at or above a cut enter (twelve bisection steps on [0, largest violation]); the others stay inactive with
            // a negative margin, i.e. they are read again in the next trial.
            double vcut = 0.0;
            int rEnter = -1;      // damped: the one row that may enter (most violated, lowest index among equals), and only when no row left
            if (damp) {
                vcut = INFINITY;
                if (!changed) {
                    double vm = 0.0;
                    for (int a = t; a < nread; a += WG) vm = fmax(vm, violation(LISTS ? list[a] : a));
                    vm = block_max(vm, c.lds);
                    if (vm > 0.0) {
                        int rb = 1 << 30;
                        for (int a = t; a < nread; a += WG) { const int r = LISTS ? list[a] : a; if (violation(r) >= vm) rb = min(rb, r); }
                        rEnter = (int)(-block_max((double)(-rb), c.lds) + 0.5);
                        vcut = vm;
                    }
                }
            } else
            if (capOn) {
                double vm = 0.0, cv = 0.0, vmax, nviol;
                for (int a = t; a < nread; a += WG) { const double v = violation(LISTS ? list[a] : a); if (v > 0.0) { cv += 1.0; vm = fmax(vm, v); } }
                block_max_sum(vm, cv, vmax, nviol, c.lds);
#ifndef LCQP_CAP_DIV
#define LCQP_CAP_DIV 8      // (experiment switch; the oracle uses 8.  Same-box A/B of n/3, /4, /5, /6, /8, /12, /16: 31.3, 31.5-32.0, 30.6, 30.3-30.6, 30.0-30.2, 31.1, 31.9 ms)
#endif
                const int cap = max(c.n / LCQP_CAP_DIV, 16);
                if (nviol > (double)cap) {
                    double lo = 0.0, hi = vmax;
                    for (int it = 0; it < 12; it++) {
                        const double mid = 0.5 * (lo + hi);
                        int cnt = 0;
                        for (int a = t; a < nread; a += WG) cnt += (violation(LISTS ? list[a] : a) >= mid);
                        if (block_sum_i(cnt, c.lds) > cap) lo = mid; else hi = mid;
                    }
                    vcut = uniform_d(lo);      // the lower end: a few more than cap rows (with the upper end a tie of many equally violated rows would never enter)
                }
            }
            // violated rows enter; fresh margins for the others; the two rules for rows flagged dependent
            int chg = 0, cntLv2 = 0, nDense = 0;
            for (int a = t; a < nread; a += WG) {
                const int r = LISTS ? list[a] : a;
                const int s = st[r];
                nDense += (hotc[r] < 0);
                const double e = ex[r], ftol = o.feasTol * (1.0 + fabs(e));
                if (s == ST_INACT) {
                    const bool may = !damp || r == rEnter;
                    if (may && e < l[r] - ftol && l[r] - e >= vcut) { st[r] = ST_LOWER; chg |= 1; }
                    else if (may && e > u[r] + ftol && e - u[r] >= vcut) { st[r] = ST_UPPER; chg |= 1; }
                    else if (LISTS) mg[r] = fmin(e - (l[r] - ftol), (u[r] + ftol) - e);      // (negative for a violated row that waits)
                } else if (depRows && dep[r]) {
                    bool viol, inside = false;
                    if (s == ST_LOWER) { viol = e < l[r] - ftol; inside = e > l[r] + ftol; }
                    else if (s == ST_UPPER) { viol = e > u[r] + ftol; inside = e < u[r] - ftol; }
                    else viol = fabs(e - l[r]) > ftol;
                    if (inside) { const double yv = yt[r]; ylv[r] = yv; yt[r] = 0.0; st[r] = ST_INACT; cntLv2 += (yv != 0.0); chg |= 1; }
                    else if (viol) { prio[r] = prioCtr + 1; chg |= 2; }
                }
            }
            if (depRows) nlv += block_sum_i(cntLv2, c.lds);
            // one reduction for the two change flags and the number of rows that were really read (rows with one non-zero are made up)
            // (twelve bits for the count: with more than 4000 rows it is left out of the sum and every row read counts as dense)
            const bool countDense = mE <= 4000;
            const unsigned packed = (unsigned)block_sum_i((chg & 1) | (((chg >> 1) & 1) << 10) | ((countDense ? nDense : 0) << 20), c.lds);
            const int chgBits = ((packed & 1023u) ? 1 : 0) | (((packed >> 10) & 1023u) ? 2 : 0);
            if (t == 0) c.info->work[4] += countDense ? (double)(packed >> 20) : (double)nread;
            if (ROBUST && (chgBits & 2)) { prioCtr++; if (t == 0) c.info->prioCtr = prioCtr; }
            changed |= (chgBits != 0);
            need_true = !changed;
        }
        PROF(c, P_RESID);
        if (need_true) {
            // (c) stage 2: the true residual -- Q and the active rows of E (cold entry: every row)
            int nact = mE;
            int* lact = c.I(I_LIST2);
            wg_symv<NCH>(c.Q, nullptr, c.n, x, nullptr, qx, nullptr, nullptr, nullptr, c.lds);
            if (LISTS) {
                nact = wg_compact(mE, [&](int r) { return cold || st[r] != ST_INACT; }, lact, c.lds);
                // du: the residual of the QP as given (the next hot start and A'y need it without the proximal term); r1: with it
                wg_rows<NCH, true>(c.E, lact, nact, x, ex, yt, c.lds, [&](int i, double s) { const double ro = -g[i] - qx[i] - s; du[i] = ro; r1[i] = ro - spv * (x[i] - xref[i]); cv[i] = fabs(g[i]) + fabs(qx[i]) + fabs(s); }, hotc, hotv);
            } else {
                wg_rows<NCH>(c.E, nullptr, mE, x, ex, yt, c.lds, [&](int i, double s) { const double ro = -g[i] - qx[i] - s; du[i] = ro; r1[i] = ro - spv * (x[i] - xref[i]); cv[i] = fabs(g[i]) + fabs(qx[i]) + fabs(s); }, hotc, hotv);
            }
            c.cSweeps++;
            if (cold) {
                if (t == 0) c.info->work[4] += (double)(nact - uniform_i(c.info->nHot));
                if (LISTS) {
                    wg_map<4>(mE, [&](int r) { return MapID3{st[r], ex[r], l[r], u[r]}; },
                              [&](int r, MapID3 v) {
                                  const double e = v.a, ftol = o.feasTol * (1.0 + fabs(e));
                                  mg[r] = (v.s != ST_INACT) ? -1.0 : fmin(e - (v.b - ftol), (v.c + ftol) - e);
                              });
                    __syncthreads();
                }
            } else {
                // (round 6; oracle: the same) THE RESIDUAL'S OWN ROUNDING FLOOR.  r1 is a sum of three vectors, g, Qx and E'y: it cannot be evaluated,
                // let alone reduced by a correction, below a few dozen roundings of the largest.  On a QP whose solution lies far out along a flat
                // direction of Q (fuzz seed 8 id 370: |g| = 2, |Qx| = |E'y| = 1e3) resTol (1 + |g|) asked for 3e-15 relative to the terms and the
                // refinement stagnated at 7e-12 with the RIGHT working set, for forty rounds.  The tolerance is at least 64 eps max_i(|g_i| + |Qx|_i + |E'y|_i);
                // well-scaled QPs never see it.
                double res_stat, rscale;
                wg_maxabs2(r1, cv, np, c.n, res_stat, rscale, c.lds);
                const double rtolS = uniform_d(fmax(rtolG, 64.0 * 2.221e-16 * rscale));
                double res_eq = 0.0, bmax = 0.0, nDense2 = 0.0;
                for (int a = t; a < nact; a += WG) {
                    const int r = LISTS ? lact[a] : a;
                    const int s = st[r];
                    nDense2 += (hotc[r] < 0) ? 1.0 : 0.0;
                    if (s == ST_INACT) continue;
                    const double bb = (s == ST_UPPER) ? u[r] : l[r];
                    const double rr = fabs(bb - ex[r]);
                    // (round 6) a row at the rounding floor of its computed E_r x cannot be held more exactly: it does not count as a residual
                    const bool above = rr > 16.0 * 2.221e-16 * (fabs(bb) + rn[r] * xnrm);
                    if (above) res_eq = fmax(res_eq, rr);
                    bmax = fmax(bmax, fabs(bb));
                    // a row of the factor that is not at that floor (counted in units of 2^20 beside the dense-row count: one reduction)
                    if (rslot[r] >= 0 && above) nDense2 += 1048576.0;
                }
                int nloose;
                { double re, nd; block_max_sum(res_eq, nDense2, re, nd, c.lds); res_eq = re; nloose = (int)(nd * (1.0 / 1048576.0)); nd -= 1048576.0 * nloose; if (t == 0) c.info->work[4] += nd; }
                bmax = block_max(bmax, c.lds);
                // the proximal QP is solved: is it the QP as given (sigma_p |x - xref| below the tolerance too)?  Else (PSD Hessians far from
                // xref) the next step of the proximal-point iteration is anchored here
#ifdef LCQP_TRACE_QP
                if (tid_here() == 0) printf("  hip trial %d stage 2: res_stat %.3e (tol %.3e) res_eq %.3e (tol %.3e) loose %d scale %.2e |x| %.2e\n", trial, res_stat, rtolS, res_eq, o.resTol * (1.0 + bmax), nloose, rscale, xnrm);
#endif
                if (res_stat <= rtolS && res_eq <= o.resTol * (1.0 + bmax) && nloose > 0 && nrefine < 2 && trial + 1 < maxTrials) {
                    nrefine++;      // solved to the residual tolerance, but the active rows can be held more exactly: one more correction
                } else
                if (res_stat <= rtolS && res_eq <= o.resTol * (1.0 + bmax) && !(wg_maxabs(du, np, c.lds) <= rtolS)) {
                    for (int i = t; i < np; i += WG) { xref[i] = x[i]; r1[i] = du[i]; }
                    __syncthreads();
                } else
                if (res_stat <= rtolS && res_eq <= o.resTol * (1.0 + bmax)) {
                    double *r1s = c.V(V_R1S), *gs0 = c.V(V_GS), *exs = c.M(M_EXS), *aty = c.V(V_ATY);
                    // A'y_A + y_box = -E'y = g + Qx + r1 at the verified point (all three are direct sums of this trial)
                    double* qxn = c.V(V_QXN);
                    for (int i = t; i < np; i += WG) { const double ro = du[i]; r1s[i] = ro; gs0[i] = g[i]; aty[i] = g[i] + qx[i] + ro; qxn[i] = qx[i]; }
                    wg_map<4>(mE, [&](int r) { return ex[r]; }, [&](int r, double v) { exs[r] = v; });
                    __syncthreads();
                    PROF(c, P_RESID);
                    return 1;
                }
            }
            have_true = 1;
            PROF(c, P_RESID);
        }
        if (changed) fact_valid = 0;
        if (!fact_valid) {
            // bring the inverse factor to the working set st[]: rows that left are rotated out, rows that entered (and rows flagged
            // dependent earlier, which may have become independent) are appended in ascending row order (oracle: the same)
            int nT = uniform_i(c.info->nT), ns = uniform_i(c.info->ns);
            int touched = 0, ndepNow = 0;
            const int ndel = wg_compact(ns, [&](int sl) { const int r = idx[sl]; return r >= 0 && st[r] == ST_INACT; }, list, c.lds);
            int cntAdd = 0;
            wg_map<4>(mE, [&](int r) { return MapID{st[r], (double)rslot[r]}; }, [&](int, MapID v) { cntAdd += (v.s != ST_INACT && v.a < 0.0); });
            const int nadd = block_sum_i(cntAdd, c.lds);
            // more active rows than variables while the set still changes by more than max(n/2, 32) rows per trial: the primal-dual update has
            // overshot (a cold start far from the solution, where every violated row enters at once) and more trials only thrash with
            // factors at full rank -- give up and let ADMM produce a working set (oracle: the same rule)
            if (trial >= 2 && nT - ndel + nadd > c.n && ndel + nadd > max(c.n / 2, 32)) return give_up();
            // in one piece when the factor is empty, when most of it would change, or when promotions dictate the order
            // (oracle: the same rule; there "in one piece" is a reset followed by appends in list order)
            PROF(c, P_UPD_PRE);
            // ... or when the row-by-row updates would cost more than the rebuild (measured: a rotation ~ 3, an append ~ 7, a rebuild ~ 96 units of
            // 7 us under load; same-box A/B 35.1 -> 34.5 ms)
            const bool bulk = (ROBUST && prioCtr > 0) || (nT == 0 && nadd > 0) || (ndel > 0 && ndel >= max(nT / 2, 8)) || nadd >= 16 || (3 * ndel + 7 * nadd >= 96);
            int naAll = 0;
            if (bulk) {
                int cntAct = 0;
                for (int r = t; r < mE; r += WG) cntAct += (st[r] != ST_INACT);
                naAll = block_sum_i(cntAct, c.lds);
                ti_reset<NCH>(c, nT, ns);
            }
            if (bulk && naAll <= capNa) {
                int na2 = 0;
                if (ROBUST)      // latest promotion first (ascending row index within one), then the rest
                    for (int stamp = prioCtr; stamp >= 1; stamp--)
                        na2 += wg_compact(mE, [&](int r) { return st[r] != ST_INACT && prio[r] == stamp; }, idx + na2, c.lds);
                na2 += wg_compact(mE, [&](int r) { return st[r] != ST_INACT && (!ROBUST || prioCtr == 0 || prio[r] == 0); }, idx + na2, c.lds);
                if (LR) {      // the Cholesky tile of the one-piece rebuild takes the whole arena: the row state waits in its global arrays
                    double *gyt = c.M(M_YT), *gex = c.M(M_EX), *gmg = c.M(M_MG); int* gst = c.I(I_STT);
                    for (int r = t; r < mE; r += WG) { gyt[r] = yt[r]; gex[r] = ex[r]; gmg[r] = mg[r]; gst[r] = st[r]; }
                    __syncthreads();
                }
                ndepNow = ti_bulk<NCH>(c, na2, o.depTau, nT, ns);
                if (LR) {
                    const double *gyt = c.M(M_YT), *gex = c.M(M_EX), *gmg = c.M(M_MG); const int* gst = c.I(I_STT);
                    __syncthreads();
                    for (int r = t; r < mE; r += WG) { yt[r] = gyt[r]; ex[r] = gex[r]; mg[r] = gmg[r]; st[r] = gst[r]; }
                    __syncthreads();
                }
                touched = 1;
                PROF(c, P_GRAM);
#ifdef LCQP_PROFILE
                c.prof[11] += 1; c.prof[12] += (unsigned long long)na2;      // one-piece rebuilds and their rows
#endif
            } else {
                // row by row: the usual small change -- or more candidate rows than the factor has room for (most of them
                // dependent, e.g. duplicated constraints): each is tested against the factor and only independent rows take a slot
                if (!bulk)
                    for (int k = 0; k < ndel; k++) {
                        const int sl = uniform_i(list[k]);
                        ti_delete<NCH>(c, sl, nT, ns);
                        touched = 1;
                    }
                PROF(c, P_DEL);
#ifdef LCQP_PROFILE
                if (!bulk) c.prof[13] += (unsigned long long)ndel;
#endif
                for (int stamp = (ROBUST && bulk) ? prioCtr : 0; stamp >= 0; stamp--) {
                    // stamp > 0: the rows of one promotion; stamp == 0: every active row that is not in the factor yet
                    const int cnt = wg_compact(mE, [&](int r) { return st[r] != ST_INACT && rslot[r] < 0 && (stamp == 0 || prio[r] == stamp); }, list, c.lds);
                    for (int k = 0; k < cnt; k++) {
                        const int r = uniform_i(list[k]);
                        const int rc = ti_append<NCH>(c, r, o.depTau, capNa, nT, ns);
                        if (rc < 0) { if (t == 0) { c.info->nT = nT; c.info->ns = ns; } __syncthreads(); return give_up(); }
                        if (ROBUST && t == 0) dep[r] = (rc == 0);
                        ndepNow += (rc == 0);
                        touched = 1;
#ifdef LCQP_PROFILE
                        c.prof[14] += 1;
#endif
                    }
                }
            }
            if (ROBUST) {
                if (ndepNow > 0 || uniform_i(c.info->ndep) > 0) {
                    __syncthreads();
                    for (int r = t; r < mE; r += WG) if (st[r] == ST_INACT || rslot[r] >= 0) dep[r] = 0;
                }
                if (t == 0) c.info->ndep = ndepNow;
            }
            PROF(c, P_CHOL);
            na = nT; nsl = ns;
            if (touched) { c.cFact++; if (t == 0) c.info->work[3] += 1.0; }
            if (t == 0) { c.info->nT = nT; c.info->ns = ns; }
            __syncthreads();
            fact_valid = 1;
        }
        const int nsp = 64 * ((nsl + 63) >> 6);
#ifdef LCQP_TRACE_QP      // diagnostic build (tools/fuzz_case.py --trace): one line per trial, the oracle prints the same lines when its trace switch is on
        if (tid_here() == 0) printf("  hip trial %d damp %d: left %d changed %d true %d na %d ns %d dep %d\n", trial, damp, nlv, changed, have_true, na, nsl, ROBUST ? c.info->ndep : 0);
#endif
        if (have_true) {
            // full correction:  c = L1^-1 r1 ;  S dy = T c - r2 ;  dx = L1^-T (c - T' dy)      (T: the rows of Et in the slots of the factor)
            for (int a = t; a < nsp; a += WG) {
                double v = 0.0;
                const int r = (a < nsl) ? idx[a] : -1;
                if (r >= 0) { const double bb = (st[r] == ST_UPPER) ? u[r] : l[r]; v = bb - ex[r]; }
                r2[a] = v;
            }
            wg_copy(cv, r1, np);
            PROF(c, P_MISC);
            wg_trsv<wg_ncopy(NCH) == 2>(c.F1, np, c.nblk, cv, true, c.lds);
            PROF(c, P_CORR_L1);
            if (na > 0) {
                // (plain loads: the same rows are read again a few microseconds later; same-box A/B 35.1 -> 33.8 ms)
#ifdef LCQP_NO_ET_KEEP
                wg_rows<NCH>(c.Et, idx, nsl, cv, dy, nullptr, c.lds, [](int, double) {});
#else
                wg_rows<NCH, false, true>(c.Et, idx, nsl, cv, dy, nullptr, c.lds, [](int, double) {});
#endif
                for (int a = t; a < nsp; a += WG) dy[a] = (a < nsl && idx[a] >= 0) ? dy[a] - r2[a] : 0.0;
                __syncthreads();
                PROF(c, P_CORR_ROWS);
            }
            if (t == 0) { c.info->work[0] += 2.0 * na; c.info->work[5] += 2.0; }
        } else {
            // predicted correction: c = sum over the rows that left of y_r Et_r; t = sum y_r M[r][W] - r2
            int nl = 0;
            if (nlv > 0) {
                nl = wg_compact(mE, [&](int r) { return ylv[r] != 0.0; }, list, c.lds);
                wg_rows<NCH, true>(c.Et, list, nl, nullptr, nullptr, ylv, c.lds, [&](int i, double s) { cv[i] = s; });
            }
            const int mMld = c.db->mMld;
            for (int a = t; a < nsp; a += WG) {
                double v = 0.0;
                const int r = (a < nsl) ? idx[a] : -1;
                if (r >= 0) {
                    const double bb = (st[r] == ST_UPPER) ? u[r] : l[r];
                    for (int k = 0; k < nl; k++) { const int rl = list[k]; v += ylv[rl] * c.MM[(size_t)max(rl, r) * mMld + min(rl, r)]; }
                    v -= bb - ex[r];
                }
                dy[a] = v;
            }
            __syncthreads();
            for (int k = t; k < nl; k += WG) ylv[list[k]] = 0.0;
            if (t == 0) { c.info->work[0] += (double)(na + nl); c.info->work[5] += 1.0; c.info->work[2] += 8.0 * (double)nl * nsl; }
            nlv = nl;        // 0: c = 0
            PROF(c, P_CORR_ROWS);
        }
        if (na > 0) {
            ti_apply<NCH>(c, dy, dy, na, nsl);
            PROF(c, P_CORR_S);
            const bool withC = have_true || nlv > 0;      // c = 0 when nothing left: cv is not read
            wg_rows<NCH>(c.Et, idx, nsl, nullptr, nullptr, dy, c.lds, [&](int i, double s) { du[i] = (withC ? cv[i] : 0.0) - s; });
            PROF(c, P_CORR_ROWS);
        } else {
            if (have_true || nlv > 0) wg_copy(du, cv, np);
            else wg_fill(du, 0.0, np);
        }
        wg_trsv<wg_ncopy(NCH) == 2>(c.F1, np, c.nblk, du, false, c.lds);
        PROF(c, P_CORR_L1);
        for (int i = t; i < np; i += WG) x[i] += du[i];
        for (int a = t; a < nsl; a += WG) {
            const int r = idx[a];
            if (r >= 0) { yt[r] += dy[a]; ex[r] = (st[r] == ST_UPPER) ? u[r] : l[r]; }      // the rows of the factor now sit on their bounds (up to rounding)
        }
        if (t == 0) c.info->work[1] += (nsl <= 64 * TI_FAST_CHUNKS) ? 0.5 * (double)na * (na + 1) : (double)na * nsl;      // entries of Ti read: its triangle (full rows beyond 256 slots)
        __syncthreads();
        c.cCorr++;
    }
    return give_up();
}

// ---------------------------------------------------------------------------------------------
// SubsolverBase::solve on the device (oracle: orc_qp_solve).  Bounds, E, factors are already set up
// (kernels k_prepare / k_factor).  initial: start from V_X0 / y0 (reference layout) like qp.init
// (src/SubsolverQPOASES.cpp:152); else continue from the stored solution and working set like
// qp.hotstart (:158).  On success the solution is left in V_XQ / M_YQ / I_ST.
// Returns 0, or the exit flag (1 max rounds, 2 infeasible bounds, 3 setup failure).
// ---------------------------------------------------------------------------------------------
// ADAPT: rho adaptation between fallback rounds (qp_adapt_rho); on in every kernel (the switch stays for A/B builds).
template <int NCH, bool ROBUST, bool ADAPT, bool LR = false>
__device__ __forceinline__ int qp_solve(Ctx<NCH>& c, int initial, const double* g, const double* y0ref, int* iterations, double gmaxHint = -1.0, int checkBounds = 1)
{
    constexpr int np = 128 * NCH;
    const lcqp_options_t& o = c.db->opt;
    const int t = tid_here(), mE = c.mE, n = c.n, nC = c.mA;
    const int trials0 = c.cTrials, admm0 = c.cAdmm;
    *iterations = 0;
    if (uniform_i(c.info->setupFail)) return 3;
    double *xq = c.V(V_XQ), *xa = c.V(V_XA), *xt = c.V(V_XT);
    double *yq = c.M(M_YQ), *ya = c.M(M_YA), *za = c.M(M_ZA), *ex = c.M(M_EX);      // (M_EX here: scratch of the ADMM start)
    double* yt = LR ? c.lds.arena + LDS_ROWS_OFF : c.M(M_YT);                               // working multipliers / working set of the polish
    int* stt = LR ? reinterpret_cast<int*>(c.lds.arena + LDS_ROWS_OFF + 3 * LDS_ROWS_MAX) : c.I(I_STT);
    const double *l = c.M(M_L), *u = c.M(M_U), *rhov = c.M(M_RHOV);
    int* st = c.I(I_ST);
    if (checkBounds) {      // (a homotopy checks once: its bounds do not change between the QPs)
        int bad = 0;
        wg_map<4>(mE, [&](int r) { return MapD4{l[r], u[r], 0.0, 0.0}; }, [&](int, MapD4 v) { bad |= (v.a > v.b); });
        if (block_or(bad, c.lds)) return 2;
    }
    // the tolerances scale with 1 + |g|_inf: handed over by a caller that has just formed g, else one pass
    const double gsc = uniform_d(1.0 + (gmaxHint >= 0.0 ? gmaxHint : wg_maxabs(g, c.n, c.lds)));
    const double ytolQ = uniform_d(o.feasTol * gsc), rtolQ = uniform_d(o.resTol * gsc);
    if (ROBUST && uniform_i(c.info->prioCtr) != 0) {     // promotions of dependent rows last for one solve
        int* prio = c.I(I_PRIO);
        for (int r = t; r < mE; r += WG) prio[r] = 0;
        __syncthreads();
        if (t == 0) c.info->prioCtr = 0;
        __syncthreads();
    }
    if (initial) {
        const double* x0 = c.V(V_X0);
        for (int i = t; i < np; i += WG) xq[i] = x0[i];
        for (int r = t; r < mE; r += WG) {
            double yr = 0.0;
            if (y0ref) yr = (r < nC) ? -y0ref[n + r] : -y0ref[c.boxidx[r - nC]];
            yq[r] = yr;
        }
        __syncthreads();
    }
    for (int i = t; i < np; i += WG) { const double v = xq[i]; xa[i] = v; c.V(V_XREF)[i] = v; }      // V_XREF: anchor of the proximal term (qp_polish)
    wg_map<4>(mE, [&](int r) { return yq[r]; }, [&](int r, double v) { ya[r] = v; });
    __syncthreads();
    int n_admm = initial ? o.admmFirst : o.admmHot;
    const int use_stored = (!initial && uniform_i(c.info->haveSolution) && n_admm == 0);
    // Rows flagged dependent keep their multiplier while a solve runs (their equations are not in the factor).  ACROSS the solves of a
    // homotopy that let the multipliers of two parallel rows -- duplicated or redundant equalities -- drift apart without bound (1e11
    // against -1e11 after eight penalty updates: only their sum is determined), until the cancellation error of A'y exceeded the
    // stationarity tolerance at a point that IS stationary (fuzz seed 22 id 283: MAX_ITERATIONS_REACHED).  A hot start therefore hands a
    // flagged row's multiplier back: it starts at zero, the stored residual no longer belongs to the stored point, and the polish takes its
    // cold entry (the true residual, every row) on the stored working set.  (round 5; oracle: orc_qp_solve)
    int reuse_stored = use_stored;
    if (ROBUST && use_stored && uniform_i(c.info->ndep) > 0) {
        const int* dep = c.I(I_DEP);
        int z = 0;
        for (int r = t; r < mE; r += WG) if (dep[r] && st[r] != ST_INACT && yq[r] != 0.0) { yq[r] = 0.0; ya[r] = 0.0; z = 1; }
        if (block_or(z, c.lds)) reuse_stored = 0;
    }
    int solved = 0, admm_ready = 0, certificate = 0;   // za = clip(E xa) is only needed once ADMM runs
    for (int round = 0; round < o.maxRounds && !solved; round++) {
        const int t = tid_here();      // per round (see qp_admm)
        if (!admm_ready && (n_admm > 0 || !(round == 0 && use_stored))) {
            wg_rows<NCH>(c.E, nullptr, mE, xa, ex, nullptr, c.lds, [](int, double) {});
            for (int r = t; r < mE; r += WG) {
                za[r] = clipd(ex[r], l[r], u[r]);
                if (rhov[r] == 0.0) ya[r] = 0.0;
            }
            __syncthreads();
            admm_ready = 1;
        }
        if (n_admm > 0) {
            if (!uniform_i(c.info->kReady) && qp_build_K<NCH>(c)) return 3;     // first ADMM iteration of this instance: L_K is built now
            qp_admm<NCH>(c, g, n_admm);
        }
        if (round == 0 && use_stored) {
            wg_map<4>(mE, [&](int r) { return MapID3{st[r], l[r], u[r], ya[r]}; },
                      [&](int r, MapID3 v) { const int s = (v.a == v.b) ? (int)ST_EQ : v.s; stt[r] = s; yt[r] = (s != ST_INACT) ? v.c : 0.0; });
        } else {
            wg_map<4>(mE, [&](int r) { return MapD4{l[r], u[r], za[r], ya[r]}; },
                      [&](int r, MapD4 v) {
                          const double lo = v.a, hi = v.b, z = v.c, y = v.d;
                          int s = ST_INACT;
                          if (isfinite(lo) && (z - lo < -y)) s = ST_LOWER;
                          if (isfinite(hi) && (hi - z < y)) s = ST_UPPER;
                          if (lo == hi) s = ST_EQ;
                          stt[r] = s;
                          yt[r] = (s != ST_INACT) ? y : 0.0;
                      });
        }
        for (int i = t; i < np; i += WG) xt[i] = xa[i];
        __syncthreads();
#ifdef LCQP_TRACE_QP
        if (tid_here() == 0) printf(" hip round %d: admm %d stored %d reuse %d trials so far %d\n", round, n_admm, use_stored, round == 0 && reuse_stored, c.cTrials - trials0);
#endif
        if (qp_polish<NCH, ROBUST, LR>(c, g, round == 0 && reuse_stored, ytolQ, rtolQ, round >= DAMP_ROUND)) { solved = 1; break; }
        if (ADAPT && round >= 1 && n_admm > 0 && qp_adapt_rho<NCH>(c, g) < 0) return 3;      // no usable ADMM factor left
        if (round >= 2) {    // at least 20 ADMM iterations behind us: is the QP infeasible or unbounded?
            certificate = qp_certificate<NCH>(c, g);
            if (certificate) break;
        }
        n_admm = 2 * n_admm;
        if (n_admm < 10) n_admm = 10;
        if (n_admm > 400) n_admm = 400;
    }
    *iterations = uniform_i((c.cTrials - trials0) + (c.cAdmm - admm0));
    if (!solved) return certificate ? certificate : 1;
    for (int i = tid_here(); i < np; i += WG) xq[i] = xt[i];
    wg_map<4>(mE, [&](int r) { return MapID{stt[r], yt[r]}; }, [&](int r, MapID v) { yq[r] = v.a; st[r] = v.s; });
    if (tid_here() == 0) c.info->haveSolution = 1;
    __syncthreads();
    return 0;
}

// write the solution in the qpOASES layout/sign: y[0:n] box duals, y[n:] row duals
template <int NCH>
__device__ __forceinline__ void qp_export(Ctx<NCH>& c, double* xdst /*np or n*/, int xlen, double* yref /*n + mA*/)
{
    const int t = tid_here(), n = c.n, nC = c.mA;
    const double *xq = c.V(V_XQ), *yq = c.M(M_YQ);
    for (int i = t; i < xlen; i += WG) xdst[i] = xq[i];
    wg_map<4>(n + nC, [&](int i) { return (i < n) ? 0.0 : -yq[i - n]; }, [&](int i, double v) { yref[i] = v; });
    __syncthreads();
    for (int k = t; k < c.info->nfin; k += WG) yref[c.boxidx[k]] = -yq[nC + k];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// LCQProblem::runSolver for one instance (oracle: orc_lcqp_solve).  src/LCQProblem.cpp:444-560.
// ---------------------------------------------------------------------------------------------
// ROBUST: the dependent-row rules of the subsolver (polish_dependent_rows); on in every kernel since round 2 (the switch stays for A/B builds).
// LR: the row state of the subsolver lives in LDS (k_lcqp_run at np <= 256).
template <int NCH, bool ROBUST, bool LR = false>
__device__ __forceinline__ void lcqp_run(Ctx<NCH>& c)
{
    constexpr int np = 128 * NCH;
    const DevBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = tid_here(), n = c.n, nC = c.nC, nComp = c.nComp, mA = c.mA;
    double *g = c.V(V_G), *gphi = c.V(V_GPHI), *gtil = c.V(V_GTIL), *xk = c.V(V_XK), *pk = c.V(V_PK), *xnew = c.V(V_XNEW);
    double *gk = c.V(V_GK), *Qx = c.V(V_QX), *Cx = c.V(V_CX), *Qp = c.V(V_QP), *Cp = c.V(V_CP), *statk = c.V(V_STATK);
    double* yk = db.yk + (size_t)c.b * db.nd;
    const bool hasPhi = db.hasLbL || db.hasLbR;
    lcqp_stats_t st;
    st.iterTotal = st.iterOuter = st.subproblemIter = st.status = st.qpSolverExitFlag = st.returnValue = 0;
    st.rhoOpt = 0.0;
    st.admmIter = st.trials = st.factorizations = st.corrections = st.qpSolves = st.reserved = 0;
    int rc = 0, qpIter = 0, histLen = 0, algoStat = 0, totalIter = 0;
    double alphak = 1.0, rho = o.initialPenaltyParameter;                     // :999-1000
    const double phiConst = uniform_d(c.info->phiConst);
    uint64_t perturbCounter = 0;
    double* hist = c.info->hist;
    if (t == 0) {
        c.info->work[0] = c.info->work[1] = c.info->work[2] = c.info->work[3] = c.info->work[4] = c.info->work[5] = 0.0;
        if (db.traceCap > 0) db.traceLen[c.b] = 0;     // a run that records nothing leaves an empty trace, not the last run's
    }

    // xk = x0, g_tilde = g   (setInitialGuess .ipp:133-158, :966-967)
    for (int i = t; i < np; i += WG) { xk[i] = c.V(V_X0)[i]; gtil[i] = g[i]; }
    __syncthreads();

    auto updatePenalty = [&]() {      // :1199-1214 (Qk = Q + rho C is never materialised: Qk v = Qv + rho Cv; g_tilde follows in the fused pass)
        if (o.nDynamicPenalty > 0) histLen = 0;
        rho = uniform_d(rho * o.penaltyUpdateFactor);      // (uniform scalars of the homotopy live in scalar registers)
        st.rhoOpt = rho;
    };
    double gmax = -1.0;      // max |gk| over the n variables, handed to the subsolver (it scales its tolerances with 1 + |g|_inf); < 0: not known
    auto solveQP = [&](int initial) -> int {   // :1115-1148 (getSolution, yk_A and pk = xnew - xk follow in the fused pass / at the exit)
        const double* y0 = (initial && c.info->hasY0) ? db.y0 + (size_t)c.b * db.nd : nullptr;
        PROF(c, P_LCQP);
        const int ef = uniform_i(qp_solve<NCH, ROBUST, true, LR>(c, initial, gk, y0, &qpIter, gmax, /*checkBounds=*/initial));
        qpIter = uniform_i(qpIter);
        PROF(c, P_MISC);
        st.subproblemIter += qpIter;
        st.qpSolverExitFlag = ef;
        st.qpSolves++;
        return ef != 0 ? LCQP_SUBPROBLEM_SOLVER_ERROR : 0;
    };

    // first QP (:452-467)
    if (o.solveZeroPenaltyFirst) {
        wg_copy(gk, g, np);
    } else {
        wg_symv<NCH>(c.C, nullptr, n, xk, nullptr, Cx, nullptr, nullptr, nullptr, c.lds);
        for (int i = t; i < np; i += WG) gk[i] = rho * Cx[i] + gtil[i];
        __syncthreads();
    }
    // One call site for the QP subsolver (the kernel carries a single copy of it): the pass below starts with the QP whose linear
    // term gk is current -- the first QP of :452-467, then the hot starts of :545 -- and continues with the top of the reference's
    // loop.  Everything between two QPs is ONE fused pass (round 3): thread t holds the entries t, t + 256, ... of every vector in
    // registers from the loads to the stores; pk goes through LDS for the product with the compressed rows of C; the scalars meet
    // in three workgroup reductions (step length; stationarity and complementarity; |gk| for the subsolver).  Q xk and C xk are kept up
    // to date by linearity (Q x_new is the direct product of the subsolver's last verification, V_QXN), A'yk_A + yk_box is taken from
    // the verified KKT residual (V_ATY): no sweep over a matrix at this level after the first.
    {
        constexpr int EPT = (np + WG - 1) / WG;      // entries per thread
        int initial = 1;
        wg_symv<NCH>(c.Q, c.C, n, xk, nullptr, Qx, Cx, nullptr, nullptr, c.lds);      // Q x0, C x0: the one sweep over Q and C of the homotopy
        const double *xq = c.V(V_XQ), *qxn = c.V(V_QXN), *aty = c.V(V_ATY);
        double* sP = c.lds.arena;                     // pk and (behind it) xk for the gathers through the rows of C
        double* sX = c.lds.arena + np;
        for (;;) {
            rc = solveQP(initial);
            const int t = tid_here();      // per iterate (see qp_polish)
            if (rc != 0) break;
            if (initial) st.rhoOpt = rho;   // :473
            const int cnz = uniform_i(c.info->cNnz);
            const int* cp = db.Cp + (size_t)c.b * (np + 1);
            const int* ci = db.Ci + (size_t)c.b * db.capC;
            const double* ccv = db.Cv + (size_t)c.b * db.capC;
            double vx[EPT], vp[EPT], vQx[EPT], vQp[EPT], vCx[EPT], vCp[EPT], vgt[EPT], vaty[EPT], vg[EPT], vgp[EPT];
            int c0[EPT], c1[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = t + e * WG;
                const bool in = i < np;
                vx[e] = in ? xk[i] : 0.0;
                const double xn = in ? xq[i] : 0.0;           // getSolution :1138
                vQx[e] = in ? Qx[i] : 0.0; vQp[e] = (in ? qxn[i] : 0.0) - vQx[e];      // Q pk = Q x_new - Q xk
                vCx[e] = in ? Cx[i] : 0.0; vgt[e] = in ? gtil[i] : 0.0; vaty[e] = in ? aty[i] : 0.0;
                vg[e] = (hasPhi && in) ? g[i] : 0.0; vgp[e] = (hasPhi && in) ? gphi[i] : 0.0;
                c0[e] = (cnz >= 0 && i < n) ? cp[i] : 0; c1[e] = (cnz >= 0 && i < n) ? cp[i + 1] : 0;
                vp[e] = xn - vx[e];                             // pk = xnew - xk :1145
                if (!initial && o.perturbStep && i < n) {
                    // perturbStep :1353-1362 (seeded SplitMix64 instead of time-seeded rand()); pk is the unperturbed difference
                    uint64_t z = o.perturbSeed + (perturbCounter + (uint64_t)i + 1ULL) * opaque_u64(0x9E3779B97F4A7C15ULL);
                    z = (z ^ (z >> 30)) * opaque_u64(0xBF58476D1CE4E5B9ULL);
                    z = (z ^ (z >> 27)) * opaque_u64(0x94D049BB133111EBULL);
                    z = z ^ (z >> 31);
                    vx[e] += ((int)(z % 3ULL) - 1) * 2.221e-16;
                }
                if (in) { sP[i] = vp[e]; sX[i] = vx[e]; }
            }
            if (!initial && o.perturbStep) perturbCounter += (uint64_t)n;
            __syncthreads();
            // C pk (:1217-1237 need Qk pk) and C xk: from the compressed rows of C when it is sparse (one-hot L, R: 2 nComp non-zeros), else
            // one sweep with both vectors.  C xk is formed from xk itself in every iterate (round 5; it used to follow the steps by
            // linearity, C xk += alpha C pk): phi = phi_const + g_phi'xk + xk'C xk / 2 (getPhi :1172-1185) cancels to 1e3 eps at a solution,
            // and forty roundings of size eps |C xk| carried along drowned that -- the reference forms the product anew each time.
            if (cnz >= 0) {
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    double sdot = 0.0, sx = 0.0;
                    for (int k = c0[e]; k < c1[e]; k++) { const double cv = ccv[k]; const int j = ci[k]; sdot += cv * sP[j]; sx += cv * sX[j]; }
                    vCp[e] = sdot;
                    vCx[e] = sx;
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int e = 0; e < EPT; e++) { const int i = t + e * WG; if (i < np) { pk[i] = vp[e]; xk[i] = vx[e]; } }
                __syncthreads();
                wg_symv_t<NCH, false, true, 1>(c.C, nullptr, n, pk, xk, Cp, nullptr, Cx, nullptr, c.lds);      // (one row in flight: the vectors of this pass stay in registers across the sweep)
#pragma unroll
                for (int e = 0; e < EPT; e++) { const int i = t + e * WG; vCp[e] = (i < np) ? Cp[i] : 0.0; vCx[e] = (i < np) ? Cx[i] : 0.0; }
            }
            if (!initial) {
                // getOptimalStepLength :1217-1237: qk = pk'Qk pk, lk = pk'(Qk xk + g_tilde)
                double sq = 0.0, sl = 0.0;
#pragma unroll
                for (int e = 0; e < EPT; e++)
                    if (t + e * WG < n) {
                        sq += vp[e] * (vQp[e] + rho * vCp[e]);
                        sl += vp[e] * ((vQx[e] + rho * vCx[e]) + vgt[e]);
                    }
                double qk, lk;
                block_sum2(sq, sl, qk, lk, c.lds);
                alphak = 1.0;
                if (qk > 0 && lk < 0) alphak = uniform_d(fmin(-lk / qk, 1.0));
            }
            initial = 0;
            // updateStep :1240-1243, updateStationarity :1246-1272 (statk = Qk xk + g_tilde - A' yk_A - yk_box), getPhi :1172-1185
            double smax = 0.0, sphi = 0.0, pmax = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                vx[e] += alphak * vp[e];
                vQx[e] += alphak * vQp[e];
                vCx[e] += alphak * vCp[e];
                if (t + e * WG < n) {
                    smax = fmax(smax, fabs((vQx[e] + rho * vCx[e]) + vgt[e] - vaty[e]));
                    sphi += vgp[e] * vx[e] + 0.5 * vx[e] * vCx[e];
                    pmax = fmax(pmax, fabs(vp[e]));
                }
            }
            double statInf, phiNow;
            block_max_sum(smax, sphi, statInf, phiNow, c.lds);
            phiNow = uniform_d(phiNow + phiConst);
            if (db.traceCap > 0 && totalIter < db.traceCap) {   // storeSteps :488-490
                // getObj :1161-1169, getMerit :1188-1196 and the step size of updateTrackingVectors (src/OutputStatistics.cpp:131-164)
                double so = 0.0, sm = 0.0;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int i = t + e * WG;
                    if (i < n) { so += g[i] * vx[e] + 0.5 * vx[e] * vQx[e]; sm += 0.5 * rho * vx[e] * vCx[e]; }
                }
                double objNow, mer;
                block_sum2(so, sm, objNow, mer, c.lds);
                const double meritNow = objNow + mer;
                const double stepNow = block_max(pmax, c.lds);
                double* ts = db.traceS + ((size_t)c.b * db.traceCap + totalIter) * 8;
                double* tx = db.traceX + ((size_t)c.b * db.traceCap + totalIter) * n;
                if (t == 0) {
                    ts[0] = statInf; ts[1] = phiNow; ts[2] = rho; ts[3] = alphak;
                    ts[4] = objNow; ts[5] = meritNow; ts[6] = stepNow; ts[7] = (double)qpIter;
#ifdef LCQP_PROFILE_STAMPS      // diagnostic: when did this iterate end (shader clock), instead of the QP iteration count
                    ts[7] = (double)clock64();
#endif
                    db.traceLen[c.b] = totalIter + 1;
                }
#pragma unroll
                for (int e = 0; e < EPT; e++) { const int i = t + e * WG; if (i < n) tx[i] = vx[e]; }
            }
            totalIter++; st.iterTotal++;
            // leyfferCheckPositive :1275-1313 (getPhi is the value of this iterate wherever the reference calls it)
            bool leyffer = false;
            {
                const int nd = o.nDynamicPenalty;
                if (nd > 0) {
                    const double cur = phiNow;
                    if (histLen < nd) { if (t == 0) hist[histLen] = cur; histLen++; __syncthreads(); }
                    else if (cur < o.complementarityTolerance) {
                        __syncthreads();
                        if (t == 0) { for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1]; hist[nd - 1] = cur; }
                        __syncthreads();
                    } else {
                        leyffer = true;
                        for (int i = 0; i < nd; i++) if (cur < o.etaDynamicPenalty * uniform_d(hist[i])) { leyffer = false; break; }
                        __syncthreads();
                        if (t == 0) { for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1]; hist[nd - 1] = cur; }
                        __syncthreads();
                    }
                }
            }
            bool penaltyUpdated = false;
            if (leyffer) { updatePenalty(); st.iterOuter++; penaltyUpdated = true; }
            // stationarity / complementarity checks :511-534
            bool converged = false;
            if (statInf < o.stationarityTolerance) {
                if (phiNow < o.complementarityTolerance) converged = true;
                else { updatePenalty(); st.iterOuter++; penaltyUpdated = true; }
            }
            // xk, Q xk, C xk; g_tilde = g + rho g_phi (updatePenalty :1199-1214); updateLinearization :1105-1112: gk = rho C xk + g_tilde
            double gm = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = t + e * WG;
                if (i < np) {
                    xk[i] = vx[e]; Qx[i] = vQx[e]; Cx[i] = vCx[e];
                    // (g_tilde = g until the first penalty update: initializeSolver :966-967 leaves rho * g_phi out, updatePenalty puts it in)
                    const double gt = (hasPhi && penaltyUpdated) ? vg[e] + rho * vgp[e] : vgt[e];
                    if (hasPhi && penaltyUpdated) gtil[i] = gt;
                    const double gkv = rho * vCx[e] + gt;
                    gk[i] = gkv;
                    if (i < n) gm = fmax(gm, fabs(gkv));
                }
            }
            gmax = block_max(gm, c.lds);
            if (converged) { rc = 0; algoStat = -1; break; }
            if (totalIter > o.maxIterations) { rc = LCQP_MAX_ITERATIONS_REACHED; break; }
            if (rho > o.maxPenaltyParameter) { rc = LCQP_MAX_PENALTY_REACHED; break; }
        }
        // getSolution (:1138-1142): the duals of the last QP that was solved, whatever the exit
        if (uniform_i(c.info->haveSolution)) qp_export<NCH>(c, xnew, np, yk);
        const int t = tid_here();
        if (algoStat == -1) {
            // transformDuals :1381-1409 (rows of L, R are rows nC.., nC+nComp.. of E)
            double* lx = c.M(M_COEF);      // scratch (M_EX belongs to the subsolver's hot start)
            wg_rows<NCH>(c.E, nullptr, mA, xk, lx, nullptr, c.lds, [](int, double) {});
            // determineStationarityType :1412-1453 on the untransformed duals, weak set :1456-1482
            int sflag = 1, mflag = 1, wflag = 0;
            const double ctol = o.complementarityTolerance;
            for (int i = 0; i < nComp; i++) {   // uniform scalar loop, order matters for the W exit
                const double Lx = lx[nC + i], Rx = lx[nC + nComp + i];
                if (!(Lx <= ctol && Rx <= ctol)) continue;
                const double a = yk[n + nC + i], bq = yk[n + nC + nComp + i];
                const double dualProd = a * bq, dualMin = fmin(a, bq);
                if (dualMin < 0) sflag = 0;
                if (fabs(dualProd) >= ctol && dualMin <= 0) {
                    if (dualProd <= ctol) { wflag = 1; break; }
                    mflag = 0;
                }
            }
            algoStat = wflag ? 1 : (sflag ? 4 : (mflag ? 3 : 2));
            __syncthreads();
            for (int i = t; i < nComp; i += WG) {
                const double Lx = lx[nC + i], Rx = lx[nC + nComp + i];
                yk[n + nC + i] -= rho * Rx;
                yk[n + nC + nComp + i] -= rho * Lx;
            }
            __syncthreads();
        }
    }
    const int tx = tid_here();
    st.status = algoStat;
    st.returnValue = rc;
    st.admmIter = c.cAdmm; st.trials = c.cTrials; st.factorizations = c.cFact; st.corrections = c.cCorr; st.reserved = c.cSweeps;
    for (int i = tx; i < n; i += WG) db.xout[(size_t)c.b * n + i] = xk[i];
    for (int i = tx; i < db.nd; i += WG) db.yout[(size_t)c.b * db.nd + i] = yk[i];
    if (tx == 0) db.stats[c.b] = st;
    PROF(c, P_LCQP);
#ifdef LCQP_PROFILE
    if (tx == 0) for (int k = 0; k < 16; k++) db.prof[(size_t)c.b * 16 + k] = c.prof[k];
#endif
    __syncthreads();
}

}  // namespace lcqp
