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
       V_XQ, V_XA, V_XT, V_R1, V_C, V_DU, V_W, V_RHS, V_LB, V_UB, V_X0, V_R1S, V_GS, V_ATY, V_NUM };
// per-instance vectors of length mEcap (rows of E = [A; L; R; box rows])
enum { M_L, M_U, M_RHOV, M_YQ, M_YA, M_ZA, M_YT, M_EX, M_COEF, M_EXS, M_DY, M_NUM };   // M_DY: change of ya in the last ADMM iteration
enum { I_ST, I_STT, I_DEP, I_PRIO, I_NUM };   // I_DEP: row flagged dependent by the last factorisation of S; I_PRIO: promotion stamp (0: none)
enum { S_R2, S_DY, S_D0, S_NUM };

struct InstInfo {
    int mE, nfin, hasY0, setupFail, haveSolution, isSetup, cacheNa, prioCtr;   // cacheNa: active rows the stored factor of S belongs to (-1: none); prioCtr: promotion stamps in use (I_PRIO)
    int ndep, pad2;                                                            // ndep: rows the stored factor of S flagged as dependent (I_DEP)
    double scale, sigma, spv, rhoAdmm, phiConst;
    double hist[8];
    double work[4];   // exact work sums for the byte accounting: sum(na), sum(na^2) over corrections; the same over factorisations
};

struct DevBatch {
    int B, n, np, nC, nComp, mA, boxcap, mEcap, capS, nblk, nd;   // nd = n + mA (dual vector, reference layout)
    int hasLbL, hasLbR;
    lcqp_options_t opt;
    double *Q, *C, *E, *Et, *F1, *FK, *S, *D1, *dscr;   // per-instance matrix blocks
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
    double *Q, *C, *E, *Et, *F1, *FK, *S, *D1, *dscr;
    double *nv, *mv, *sv;
    int *mi, *idx, *boxidx;
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

// phase buckets of the diagnostic build (tools/gpu_phase_profile.py)
enum { P_LCQP = 0, P_RESID = 1, P_GRAM = 2, P_CHOL = 3, P_CORR_L1 = 4, P_CORR_ROWS = 5, P_CORR_S = 6, P_ADMM = 7, P_MISC = 8 };
#ifdef LCQP_PROFILE
#define PROF(c, k) do { unsigned long long t_ = clock64(); (c).prof[k] += t_ - (c).tlast; (c).tlast = t_; } while (0)
#else
#define PROF(c, k) do { } while (0)
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
    const int t = threadIdx.x, mE = c.mE;
    const double alpha = o.admmAlpha, sigma = c.info->sigma;
    double *xa = c.V(V_XA), *rhs = c.V(V_RHS), *w = c.V(V_W);
    double *ya = c.M(M_YA), *za = c.M(M_ZA), *rhov = c.M(M_RHOV), *coef = c.M(M_COEF), *ex = c.M(M_EX), *dyl = c.M(M_DY);
    const double *l = c.M(M_L), *u = c.M(M_U);
    PROF(c, P_MISC);
    for (int it = 0; it < n_it; it++) {
        for (int r = t; r < mE; r += WG) coef[r] = rhov[r] * za[r] - ya[r];
        __syncthreads();
        // rhs = sigma*xa - g + E'(rho.z - y)
        wg_rows<NCH>(c.E, nullptr, mE, nullptr, nullptr, coef, c.lds,
                     [&](int i, double s) { rhs[i] = sigma * xa[i] - g[i] + s; });
        wg_trsv(c.FK, np, c.nblk, rhs, true, c.lds);
        wg_trsv(c.FK, np, c.nblk, rhs, false, c.lds);       // rhs = xt
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
    const int t = threadIdx.x, mE = c.mE;
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
// OSQP's rho adaptation for the fallback rounds (oracle: qp_adapt_rho): after a failed round, scale all rho_i by
//   sqrt( (|E xa - za| / max(|E xa|, |za|)) / (|Q xa + g + E'ya| / max(|Q xa|, |E'ya|, |g|)) )   (infinity norms, clipped to
// [1e-3, 1e3]) when that factor is above 5 or below 1/5, and refactorise K = Q + sigma I + E' diag(rho) E in place (the
// work k_build_K and k_factor do at setup, here by this workgroup alone).  Returns 1 (uniform) when rho changed.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ int qp_adapt_rho(Ctx<NCH>& c, const double* g)
{
    constexpr int np = 128 * NCH;
    const int t = threadIdx.x, mE = c.mE;
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
                    if (I != J) c.FK[(size_t)gj * np + gi] = v;     // see k_build_K
                }
        }
    __syncthreads();
    wg_chol(c.FK, np, c.nblk, c.n, 0.0, c.dscr, nullptr, nullptr, c.lds, 0);
    __syncthreads();
    c.cFact++;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Dependent-row rules of the single-QP kernel (k_qp_solve, ROBUST = true; oracle: q->robust).  k_lcqp_run runs without
// them: inside the persistent kernel they cost 6 % through register allocation.
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
    for (int r = threadIdx.x; r < mE; r += WG) {
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
// Primal-dual active-set polish in correction form (oracle: qp_polish).
// In/out: x = V_XT, multipliers M_YT (OSQP sign, zero on inactive rows), active set I_STT.
// Returns 1 (uniform) on a verified KKT point.
// reuse != 0 (hot start from the last verified solution): the first trial needs no sweep, because
// r1 = r1_last + (g_last - g) and E x = ex_last hold exactly for an unchanged (x, y).
// ---------------------------------------------------------------------------------------------
template <int NCH, bool ROBUST>
__device__ __forceinline__ int qp_polish(Ctx<NCH>& c, const double* g, int reuse)
{
    constexpr int np = 128 * NCH;
    const lcqp_options_t& o = c.db->opt;
    const int t = threadIdx.x, mE = c.mE, capS = c.capS;
    double *x = c.V(V_XT), *r1 = c.V(V_R1), *cv = c.V(V_C), *du = c.V(V_DU), *qx = c.V(V_TMP);
    double *yt = c.M(M_YT), *ex = c.M(M_EX), *coef = c.M(M_COEF);
    const double *l = c.M(M_L), *u = c.M(M_U);
    int* st = c.I(I_STT);
    int *dep = c.I(I_DEP), *prio = c.I(I_PRIO);
    double *r2 = c.Sv(S_R2), *dy = c.Sv(S_DY), *d0 = c.Sv(S_D0);
    int* idx = c.idx;
    const double gs = 1.0 + wg_maxabs(g, c.n, c.lds);
    const double ytol = o.feasTol * gs;
    int na = 0, nblkS = 0, fact_valid = 0;
    int prioCtr = ROBUST ? uniform_i(c.info->prioCtr) : 0;
    const int capNa = min(min(min(max(2 * c.n, 64), mE), capS), max_active(NCH));   // room for the degenerate vertices of small problems

    for (int trial = 0; trial < o.maxTrials; trial++) {
        c.cTrials++;
        int chg = 0;
        PROF(c, P_MISC);
        // leaving rows (wrong-signed multipliers) drop out before the residual is formed
        for (int r = t; r < mE; r += WG) {
            double yv = yt[r];
            if (trial > 0) {
                const int s = st[r];
                if ((s == ST_LOWER && yv > ytol) || (s == ST_UPPER && yv < -ytol)) { st[r] = ST_INACT; yv = 0.0; yt[r] = 0.0; chg = 1; }
                else if (ROBUST && s == ST_INACT && yv != 0.0) { yv = 0.0; yt[r] = 0.0; }   // left as "dependent, inside" in the last trial
            }
            coef[r] = yv;
        }
        __syncthreads();
        if (trial == 0 && reuse) {
            const double *r1s = c.V(V_R1S), *gs0 = c.V(V_GS), *exs = c.M(M_EXS);
            for (int i = t; i < np; i += WG) r1[i] = r1s[i] + (gs0[i] - g[i]);
            for (int r = t; r < mE; r += WG) ex[r] = exs[r];
            __syncthreads();
        } else {
            // residual evaluation: one sweep over Q, one over E
            wg_symv<NCH>(c.Q, nullptr, c.n, x, nullptr, qx, nullptr, nullptr, nullptr, c.lds);
            wg_rows<NCH>(c.E, nullptr, mE, x, ex, coef, c.lds, [&](int i, double s) { r1[i] = -g[i] - qx[i] - s; });
            c.cSweeps++;
        }
        const double res_stat = wg_maxabs(r1, np, c.lds);
        double res_eq = 0.0, bmax = 0.0;
        for (int r = t; r < mE; r += WG) {
            const int s = st[r];
            const double e = ex[r];
            if (s == ST_INACT) {
                if (trial > 0) {
                    const double ftol = o.feasTol * (1.0 + fabs(e));
                    if (e < l[r] - ftol) { st[r] = ST_LOWER; chg = 1; }
                    else if (e > u[r] + ftol) { st[r] = ST_UPPER; chg = 1; }
                }
            } else {
                const double bb = (s == ST_UPPER) ? u[r] : l[r];
                res_eq = fmax(res_eq, fabs(bb - e));
                bmax = fmax(bmax, fabs(bb));
            }
        }
        int changed;
        if (ROBUST) {
            if (trial > 0 && uniform_i(c.info->ndep) > 0) chg |= polish_dependent_rows(st, dep, prio, ex, l, u, mE, o.feasTol, prioCtr + 1);
            const int chgBits = block_or_bits(chg, c.lds);
            if (chgBits & 2) { prioCtr++; if (t == 0) c.info->prioCtr = prioCtr; }
            changed = chgBits != 0;
        } else {
            changed = block_or(chg, c.lds);
        }
        res_eq = block_max(res_eq, c.lds);
        bmax = block_max(bmax, c.lds);
        PROF(c, P_RESID);
        if (trial > 0 && !changed && res_stat <= o.resTol * gs && res_eq <= o.resTol * (1.0 + bmax)) {
            double *r1s = c.V(V_R1S), *gs0 = c.V(V_GS), *exs = c.M(M_EXS), *aty = c.V(V_ATY);
            // A'y_A + y_box = -E'y = g + Qx + r1 at the verified point (all three are direct sums of this trial)
            for (int i = t; i < np; i += WG) { r1s[i] = r1[i]; gs0[i] = g[i]; aty[i] = g[i] + qx[i] + r1[i]; }
            for (int r = t; r < mE; r += WG) exs[r] = ex[r];
            __syncthreads();
            return 1;
        }
        if (changed) fact_valid = 0;
        if (!fact_valid) {
            // ordered list of active rows (ascending row index, as the oracle builds it)
            const int per = (mE + WG - 1) / WG;
            const int r0 = t * per, r1e = min(mE, r0 + per);
            int differs = 0, nprom = 0;
            if (ROBUST && prioCtr > 0) {
                // promoted rows come first in the list
                __syncthreads();
                const int packed = polish_promoted_list(st, prio, idx, mE, r0, r1e, capNa, c.lds);
                nprom = packed >> 1; differs = packed & 1;
                if (nprom > capNa) return 0;
            }
            int cnt = 0;
            for (int r = r0; r < r1e; r++) cnt += (st[r] != ST_INACT && (!ROBUST || prioCtr == 0 || prio[r] == 0));
            // exclusive scan over 256 threads
            int incl = cnt;
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) { int v = __shfl_up(incl, ofs, 64); if (lane_id() >= ofs) incl += v; }
            if (lane_id() == 63) c.lds.ired[8 + wave_id()] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave_id(); w++) base += c.lds.ired[8 + w];
            na = uniform_i(c.lds.ired[8] + c.lds.ired[9] + c.lds.ired[10] + c.lds.ired[11]) + nprom;
            __syncthreads();
            if (na > capNa) return 0;
            // the factor of S only depends on the list: reuse it when the list is the one it was built for
            const int cachedNa = c.info->cacheNa;
            int pos = nprom + base + incl - cnt;
            differs |= (cachedNa != na);
            for (int r = r0; r < r1e; r++)
                if (st[r] != ST_INACT && (!ROBUST || prioCtr == 0 || prio[r] == 0)) { differs |= (idx[pos] != r); idx[pos++] = r; }
            nblkS = (na + 63) >> 6;
            for (int a = na + t; a < 64 * nblkS; a += WG) idx[a] = -1;
            const int rebuild = block_or(differs, c.lds);
            if (rebuild) {
            // S = T T' (lower tiles), T = rows idx[] of Et; padded rows get a unit diagonal
            for (int Ib = 0; Ib < nblkS; Ib++)
                for (int Jb = 0; Jb <= Ib; Jb++) {
                    double acc[4][4];
                    const int* ia = idx + 64 * Ib;
                    const int* ib = idx + 64 * Jb;
                    wg_tile_nt(acc, c.Et, np, [=](int r) { return (long)ia[r]; }, c.Et, np, [=](int r) { return (long)ib[r]; }, np, c.lds, na - 64 * Ib);
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int gi = 64 * Ib + tile_li(i, j), gj = 64 * Jb + tile_lj(i, j);
                            double v = acc[i][j];
                            if (gi == gj && gi >= na) v = 1.0;
                            c.S[(size_t)gi * capS + gj] = v;
                        }
                    __syncthreads();
                }
            PROF(c, P_GRAM);
            if (nblkS > 0) wg_chol(c.S, capS, nblkS, na, o.depTau, c.dscr, d0, nullptr, c.lds, 0);
            PROF(c, P_CHOL);
            c.cFact++;
            if (ROBUST) {
                // rows the safeguarded factorisation flagged as dependent: the diagonal of the stored inverse blocks is 1/l_aa = 1e-150
                int nflag = 0;
                for (int a = t; a < na; a += WG) nflag += (c.S[(size_t)a * capS + a] < 1e-100);
                nflag = block_sum_i(nflag, c.lds);
                if (nflag > 0 || uniform_i(c.info->ndep) > 0) {
                    for (int r = t; r < mE; r += WG) dep[r] = 0;
                    __syncthreads();
                    for (int a = t; a < na; a += WG) if (c.S[(size_t)a * capS + a] < 1e-100) dep[idx[a]] = 1;
                }
                if (t == 0) c.info->ndep = nflag;
            }
            if (t == 0) { c.info->cacheNa = na; c.info->work[2] += (double)na; c.info->work[3] += (double)na * na; }
            __syncthreads();
            }   // rebuild
            fact_valid = 1;
        }
        // correction:  c = L1^-1 r1 ;  S dy = T c - r2 ;  dx = L1^-T (c - T' dy)
        for (int a = t; a < 64 * nblkS; a += WG) {
            double v = 0.0;
            if (a < na) { const int r = idx[a]; const double bb = (st[r] == ST_UPPER) ? u[r] : l[r]; v = bb - ex[r]; }
            r2[a] = v;
        }
        wg_copy(cv, r1, np);
        PROF(c, P_MISC);
        wg_trsv(c.F1, np, c.nblk, cv, true, c.lds);
        PROF(c, P_CORR_L1);
        if (na > 0) {
            wg_rows<NCH>(c.Et, idx, na, cv, dy, nullptr, c.lds, [](int, double) {});
            for (int a = t; a < 64 * nblkS; a += WG) dy[a] = (a < na) ? dy[a] - r2[a] : 0.0;
            __syncthreads();
            PROF(c, P_CORR_ROWS);
            wg_trsv(c.S, capS, nblkS, dy, true, c.lds);
            wg_trsv(c.S, capS, nblkS, dy, false, c.lds);
            PROF(c, P_CORR_S);
            wg_rows<NCH>(c.Et, idx, na, nullptr, nullptr, dy, c.lds, [&](int i, double s) { du[i] = cv[i] - s; });
            PROF(c, P_CORR_ROWS);
        } else {
            wg_copy(du, cv, np);
        }
        wg_trsv(c.F1, np, c.nblk, du, false, c.lds);
        PROF(c, P_CORR_L1);
        for (int i = t; i < np; i += WG) x[i] += du[i];
        for (int a = t; a < na; a += WG) yt[idx[a]] += dy[a];
        if (t == 0) { c.info->work[0] += (double)na; c.info->work[1] += (double)na * na; }
        __syncthreads();
        c.cCorr++;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// SubsolverBase::solve on the device (oracle: orc_qp_solve).  Bounds, E, factors are already set up
// (kernels k_prepare / k_factor).  initial: start from V_X0 / y0 (reference layout) like qp.init
// (src/SubsolverQPOASES.cpp:152); else continue from the stored solution and working set like
// qp.hotstart (:158).  On success the solution is left in V_XQ / M_YQ / I_ST.
// Returns 0, or the exit flag (1 max rounds, 2 infeasible bounds, 3 setup failure).
// ---------------------------------------------------------------------------------------------
// ADAPT: rho adaptation between fallback rounds (qp_adapt_rho); on in every kernel (the switch stays for A/B builds).
template <int NCH, bool ROBUST, bool ADAPT>
__device__ __forceinline__ int qp_solve(Ctx<NCH>& c, int initial, const double* g, const double* y0ref, int* iterations)
{
    constexpr int np = 128 * NCH;
    const lcqp_options_t& o = c.db->opt;
    const int t = threadIdx.x, mE = c.mE, n = c.n, nC = c.mA;
    const int trials0 = c.cTrials, admm0 = c.cAdmm;
    *iterations = 0;
    if (c.info->setupFail) return 3;
    double *xq = c.V(V_XQ), *xa = c.V(V_XA), *xt = c.V(V_XT);
    double *yq = c.M(M_YQ), *ya = c.M(M_YA), *za = c.M(M_ZA), *yt = c.M(M_YT), *ex = c.M(M_EX);
    const double *l = c.M(M_L), *u = c.M(M_U), *rhov = c.M(M_RHOV);
    int *st = c.I(I_ST), *stt = c.I(I_STT);
    int bad = 0;
    for (int r = t; r < mE; r += WG) bad |= (l[r] > u[r]);
    if (block_or(bad, c.lds)) return 2;
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
    wg_copy(xa, xq, np);
    for (int r = t; r < mE; r += WG) ya[r] = yq[r];
    __syncthreads();
    int n_admm = initial ? o.admmFirst : o.admmHot;
    const int use_stored = (!initial && c.info->haveSolution && n_admm == 0);
    int solved = 0, admm_ready = 0, certificate = 0;   // za = clip(E xa) is only needed once ADMM runs
    for (int round = 0; round < o.maxRounds && !solved; round++) {
        if (!admm_ready && (n_admm > 0 || !(round == 0 && use_stored))) {
            wg_rows<NCH>(c.E, nullptr, mE, xa, ex, nullptr, c.lds, [](int, double) {});
            for (int r = t; r < mE; r += WG) {
                za[r] = clipd(ex[r], l[r], u[r]);
                if (rhov[r] == 0.0) ya[r] = 0.0;
            }
            __syncthreads();
            admm_ready = 1;
        }
        if (n_admm > 0) qp_admm<NCH>(c, g, n_admm);
        for (int r = t; r < mE; r += WG) {
            int s;
            if (round == 0 && use_stored) {
                s = st[r];
                if (l[r] == u[r]) s = ST_EQ;
            } else {
                const double lo = l[r], hi = u[r], z = za[r], y = ya[r];
                s = ST_INACT;
                if (isfinite(lo) && (z - lo < -y)) s = ST_LOWER;
                if (isfinite(hi) && (hi - z < y)) s = ST_UPPER;
                if (lo == hi) s = ST_EQ;
            }
            stt[r] = s;
            yt[r] = (s != ST_INACT) ? ya[r] : 0.0;
        }
        for (int i = t; i < np; i += WG) xt[i] = xa[i];
        __syncthreads();
        if (qp_polish<NCH, ROBUST>(c, g, round == 0 && use_stored)) { solved = 1; break; }
        if (ADAPT && round >= 1 && n_admm > 0) qp_adapt_rho<NCH>(c, g);
        if (round >= 2) {    // at least 20 ADMM iterations behind us: is the QP infeasible or unbounded?
            certificate = qp_certificate<NCH>(c, g);
            if (certificate) break;
        }
        n_admm = 2 * n_admm;
        if (n_admm < 10) n_admm = 10;
        if (n_admm > 400) n_admm = 400;
    }
    *iterations = (c.cTrials - trials0) + (c.cAdmm - admm0);
    if (!solved) return certificate ? certificate : 1;
    for (int i = t; i < np; i += WG) xq[i] = xt[i];
    for (int r = t; r < mE; r += WG) { yq[r] = yt[r]; st[r] = stt[r]; }
    if (t == 0) c.info->haveSolution = 1;
    __syncthreads();
    return 0;
}

// write the solution in the qpOASES layout/sign: y[0:n] box duals, y[n:] row duals
template <int NCH>
__device__ __forceinline__ void qp_export(Ctx<NCH>& c, double* xdst /*np or n*/, int xlen, double* yref /*n + mA*/)
{
    const int t = threadIdx.x, n = c.n, nC = c.mA;
    const double *xq = c.V(V_XQ), *yq = c.M(M_YQ);
    for (int i = t; i < xlen; i += WG) xdst[i] = xq[i];
    for (int i = t; i < n + nC; i += WG) yref[i] = (i < n) ? 0.0 : -yq[i - n];
    __syncthreads();
    for (int k = t; k < c.info->nfin; k += WG) yref[c.boxidx[k]] = -yq[nC + k];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// LCQProblem::runSolver for one instance (oracle: orc_lcqp_solve).  src/LCQProblem.cpp:444-560.
// ---------------------------------------------------------------------------------------------
// ROBUST selects the QP subsolver variant (qp_solve<NCH, ROBUST>): false in k_lcqp_run, true in k_lcqp_rerun, which repeats
// the instances that ended with SUBPROBLEM_SOLVER_ERROR.
template <int NCH, bool ROBUST>
__device__ __forceinline__ void lcqp_run(Ctx<NCH>& c)
{
    constexpr int np = 128 * NCH;
    const DevBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = threadIdx.x, n = c.n, nC = c.nC, nComp = c.nComp, mA = c.mA;
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
    const double phiConst = c.info->phiConst;
    uint64_t perturbCounter = 0;
    double* hist = c.info->hist;
    if (t == 0) {
        c.info->work[0] = c.info->work[1] = c.info->work[2] = c.info->work[3] = 0.0;
        if (db.traceCap > 0) db.traceLen[c.b] = 0;     // a run that records nothing leaves an empty trace, not the last run's
    }

    // xk = x0, g_tilde = g   (setInitialGuess .ipp:133-158, :966-967)
    for (int i = t; i < np; i += WG) { xk[i] = c.V(V_X0)[i]; gtil[i] = g[i]; }
    __syncthreads();

    auto getPhi = [&]() -> double {   // :1172-1185 with Cx = C*xk current
        double s = 0.0;
        for (int i = t; i < n; i += WG) s += (hasPhi ? gphi[i] * xk[i] : 0.0) + 0.5 * xk[i] * Cx[i];
        return phiConst + block_sum(s, c.lds);
    };
    auto updatePenalty = [&]() {      // :1199-1214 (Qk = Q + rho C is never materialised: Qk v = Qv + rho Cv)
        if (o.nDynamicPenalty > 0) histLen = 0;
        rho *= o.penaltyUpdateFactor;
        st.rhoOpt = rho;
        if (hasPhi) { for (int i = t; i < np; i += WG) gtil[i] = g[i] + rho * gphi[i]; __syncthreads(); }
    };
    auto solveQP = [&](int initial) -> int {   // :1115-1148
        const double* y0 = (initial && c.info->hasY0) ? db.y0 + (size_t)c.b * db.nd : nullptr;
        PROF(c, P_LCQP);
        const int ef = qp_solve<NCH, ROBUST, true>(c, initial, gk, y0, &qpIter);
        PROF(c, P_MISC);
        st.subproblemIter += qpIter;
        st.qpSolverExitFlag = ef;
        st.qpSolves++;
        if (ef != 0) return LCQP_SUBPROBLEM_SOLVER_ERROR;
        qp_export<NCH>(c, xnew, np, yk);
        for (int i = t; i < np; i += WG) pk[i] = xnew[i] - xk[i];
        __syncthreads();
        return 0;
    };

    // first QP (:452-467)
    if (o.solveZeroPenaltyFirst) {
        wg_copy(gk, g, np);
    } else {
        wg_symv<NCH>(c.C, nullptr, n, xk, nullptr, Cx, nullptr, nullptr, nullptr, c.lds);
        for (int i = t; i < np; i += WG) gk[i] = rho * Cx[i] + gtil[i];
        __syncthreads();
    }
    // One call site for the QP subsolver and one for the Q/C sweep (the kernel carries a single copy of each): the pass
    // below starts with the QP whose linear term gk is current -- the first QP of :452-467, then the hot starts of :545 --
    // and continues with the top of the reference's loop.
    // One sweep over Q and C per iterate: Q*[pk, xk] and C*[pk, xk] (the sweep getOptimalStepLength needs,
    // :1217-1237).  Q*(xk + alpha pk) in updateStep follows by linearity from these two direct products (no
    // recurrence over iterates), and A'yk_A + yk_box is taken from the verified KKT residual of the subproblem
    // (V_ATY), so updateStationarity needs no sweep of its own.
    {
        int initial = 1;
        for (;;) {
            rc = solveQP(initial);
            if (rc != 0) break;
            if (initial) {
                st.rhoOpt = rho;   // :473
            } else if (o.perturbStep) {
                // perturbStep :1353-1362 (seeded SplitMix64 instead of time-seeded rand())
                for (int i = t; i < n; i += WG) {
                    uint64_t z = o.perturbSeed + (perturbCounter + (uint64_t)i + 1ULL) * 0x9E3779B97F4A7C15ULL;
                    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                    z = z ^ (z >> 31);
                    xk[i] += ((int)(z % 3ULL) - 1) * 2.221e-16;
                }
                perturbCounter += (uint64_t)n;
                __syncthreads();
            }
            wg_symv<NCH>(c.Q, c.C, n, pk, xk, Qp, Cp, Qx, Cx, c.lds);
            if (!initial) {
                // getOptimalStepLength :1217-1237: qk = pk'Qk pk, lk = pk'(Qk xk + g_tilde)
                double sq = 0.0, sl = 0.0;
                for (int i = t; i < n; i += WG) {
                    sq += pk[i] * (Qp[i] + rho * Cp[i]);
                    sl += pk[i] * ((Qx[i] + rho * Cx[i]) + gtil[i]);
                }
                const double qk = block_sum(sq, c.lds), lk = block_sum(sl, c.lds);
                alphak = 1.0;
                if (qk > 0 && lk < 0) alphak = fmin(-lk / qk, 1.0);
            }
            initial = 0;
            // updateStep :1240-1243
            for (int i = t; i < np; i += WG) {
                xk[i] = xk[i] + alphak * pk[i];
                Qx[i] = Qx[i] + alphak * Qp[i];
                Cx[i] = Cx[i] + alphak * Cp[i];
            }
            __syncthreads();
            // updateStationarity :1246-1272: statk = Qk xk + g_tilde - A' yk_A - yk_box
            {
                const double* aty = c.V(V_ATY);
                for (int i = t; i < np; i += WG) statk[i] = (i < n) ? (Qx[i] + rho * Cx[i]) + gtil[i] - aty[i] : 0.0;
                __syncthreads();
            }
            const double statInf = wg_maxabs(statk, n, c.lds);
            if (db.traceCap > 0 && totalIter < db.traceCap) {   // storeSteps :488-490
                const double phiNow = getPhi();
                // getObj :1161-1169, getMerit :1188-1196 and the step size of updateTrackingVectors (src/OutputStatistics.cpp:131-164)
                double so = 0.0, sm = 0.0;
                for (int i = t; i < n; i += WG) { so += g[i] * xk[i] + 0.5 * xk[i] * Qx[i]; sm += 0.5 * rho * xk[i] * Cx[i]; }
                const double objNow = block_sum(so, c.lds), meritNow = objNow + block_sum(sm, c.lds);
                const double stepNow = wg_maxabs(pk, n, c.lds);
                double* ts = db.traceS + ((size_t)c.b * db.traceCap + totalIter) * 8;
                double* tx = db.traceX + ((size_t)c.b * db.traceCap + totalIter) * n;
                if (t == 0) {
                    ts[0] = statInf; ts[1] = phiNow; ts[2] = rho; ts[3] = alphak;
                    ts[4] = objNow; ts[5] = meritNow; ts[6] = stepNow; ts[7] = (double)qpIter;
                    db.traceLen[c.b] = totalIter + 1;
                }
                for (int i = t; i < n; i += WG) tx[i] = xk[i];
            }
            totalIter++; st.iterTotal++;
            // leyfferCheckPositive :1275-1313
            bool leyffer = false;
            {
                const int nd = o.nDynamicPenalty;
                if (nd > 0) {
                    const double cur = getPhi();
                    if (histLen < nd) { if (t == 0) hist[histLen] = cur; histLen++; __syncthreads(); }
                    else if (cur < o.complementarityTolerance) {
                        __syncthreads();
                        if (t == 0) { for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1]; hist[nd - 1] = cur; }
                        __syncthreads();
                    } else {
                        leyffer = true;
                        for (int i = 0; i < nd; i++) if (cur < o.etaDynamicPenalty * hist[i]) { leyffer = false; break; }
                        __syncthreads();
                        if (t == 0) { for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1]; hist[nd - 1] = cur; }
                        __syncthreads();
                    }
                }
            }
            if (leyffer) { updatePenalty(); st.iterOuter++; }
            // stationarity / complementarity checks :511-534
            if (statInf < o.stationarityTolerance) {
                if (getPhi() < o.complementarityTolerance) {
                    // transformDuals :1381-1409 (rows of L, R are rows nC.., nC+nComp.. of E)
                    double* lx = c.M(M_EX);
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
                    rc = 0;
                    break;
                } else {
                    updatePenalty(); st.iterOuter++;
                }
            }
            if (totalIter > o.maxIterations) { rc = LCQP_MAX_ITERATIONS_REACHED; break; }
            if (rho > o.maxPenaltyParameter) { rc = LCQP_MAX_PENALTY_REACHED; break; }
            // updateLinearization :1105-1112: gk = rho C xk + g_tilde
            for (int i = t; i < np; i += WG) gk[i] = rho * Cx[i] + gtil[i];
            __syncthreads();
        }
    }
    st.status = algoStat;
    st.returnValue = rc;
    st.admmIter = c.cAdmm; st.trials = c.cTrials; st.factorizations = c.cFact; st.corrections = c.cCorr; st.reserved = c.cSweeps;
    for (int i = t; i < n; i += WG) db.xout[(size_t)c.b * n + i] = xk[i];
    for (int i = t; i < db.nd; i += WG) db.yout[(size_t)c.b * db.nd + i] = yk[i];
    if (t == 0) db.stats[c.b] = st;
    PROF(c, P_LCQP);
#ifdef LCQP_PROFILE
    if (t == 0) for (int k = 0; k < 16; k++) db.prof[(size_t)c.b * 16 + k] = c.prof[k];
#endif
    __syncthreads();
}

}  // namespace lcqp
