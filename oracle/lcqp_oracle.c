/*
 * lcqp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See lcqp_oracle.h.
 * Every function cites the reference file:line (under /root/reference) it restates.
 */
#define _GNU_SOURCE      /* pthread_setaffinity_np, CPU_SET (orc_synth_bench) */
#include "lcqp_oracle.h"
#include "../include/lcqp_synth.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORC_EPS 2.221e-16 /* include/Utilities.hpp:350 */

/* ------------------------------------------------------------------------------------------------
 * Options defaults: src/Options.cpp:296-333
 * ---------------------------------------------------------------------------------------------- */
void orc_options_default(orc_options_t* o)
{
    memset(o, 0, sizeof(*o));
    o->complementarityTolerance = 1.0e3 * ORC_EPS;
    o->stationarityTolerance = 1.0e6 * ORC_EPS;
    o->initialPenaltyParameter = 0.01;
    o->penaltyUpdateFactor = 2.0;
    o->maxPenaltyParameter = 1e8;
    o->etaDynamicPenalty = 0.9;
    o->solveZeroPenaltyFirst = 1;
    o->perturbStep = 1;
    o->maxIterations = 1000;
    o->nDynamicPenalty = 3;
    o->printLevel = 2;
    o->storeSteps = 0;
    o->perturbSeed = 0x5EEDULL;
    o->admmRho = 0.1;
    o->admmSigma = 1e-6;
    o->admmAlpha = 1.6;
    o->rhoEqMult = 1e3;
    o->proxSmall = 1e-12;
    o->proxBig = 1e-8;
    o->pivotThreshold = 1e-7;
    o->depTau = 1e-12;
    o->feasTol = 1e-9;
    o->resTol = 1e-12;
    o->admmFirst = 0;
    o->admmHot = 0;
    o->maxTrials = 16;   /* 12 until round 3: cold starts of the synthetic workload need up to 14 trials (lcqp_hip_options_default) */
    o->maxRounds = 40;
}

/* ------------------------------------------------------------------------------------------------
 * Utilities restated loop-for-loop (row-major dense): src/Utilities.cpp
 * ---------------------------------------------------------------------------------------------- */
void orc_util_matmul(const double* A, const double* B, double* C, int m, int n, int p)
{ /* :38-47 */
    for (int i = 0; i < m; i++)
        for (int j = 0; j < p; j++) {
            C[i * p + j] = 0;
            for (int k = 0; k < n; k++) C[i * p + j] += A[i * n + k] * B[k * p + j];
        }
}

void orc_util_matmul_t(const double* A, const double* B, double* C, int m, int n, int p)
{ /* :62-72  C = A' * B, A is m x n */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) {
            C[i * p + j] = 0;
            for (int k = 0; k < m; k++) C[i * p + j] += A[k * n + i] * B[k * p + j];
        }
}

void orc_util_add_matmul_t(const double* A, const double* B, double* C, int m, int n, int p)
{ /* :85-93  C += A' * B */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++)
            for (int k = 0; k < m; k++) C[i * p + j] += A[k * n + i] * B[k * p + j];
}

void orc_util_symm_product(const double* A, const double* B, double* C, int m, int n)
{ /* :104-116  C = A'B + B'A */
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            C[i * n + j] = 0;
            for (int k = 0; k < m; k++) C[i * n + j] += A[k * n + i] * B[k * n + j] + B[k * n + i] * A[k * n + j];
            C[j * n + i] = C[i * n + j];
        }
}

void orc_util_affine(double alpha, const double* A, const double* b, const double* c, double* d, int m, int n)
{ /* :176-186  d = alpha*A*b + c */
    for (int i = 0; i < m; i++) {
        double tmp = 0;
        for (int k = 0; k < n; k++) tmp += A[i * n + k] * b[k];
        d[i] = alpha * tmp + c[i];
    }
}

void orc_util_weighted_matadd(double alpha, const double* A, double beta, const double* B, double* C, int m, int n)
{ /* :202-206 */
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) C[i * n + j] = alpha * A[i * n + j] + beta * B[i * n + j];
}

void orc_util_weighted_vecadd(double alpha, const double* a, double beta, const double* b, double* c, int m)
{ /* :209-211 */
    orc_util_weighted_matadd(alpha, a, beta, b, c, m, 1);
}

double orc_util_quadform(const double* Q, const double* p, int m)
{ /* :214-225 */
    double ret = 0;
    for (int i = 0; i < m; i++) {
        double tmp = 0;
        for (int j = 0; j < m; j++) tmp += Q[i * m + j] * p[j];
        ret += tmp * p[i];
    }
    return ret;
}

double orc_util_dot(const double* a, const double* b, int m)
{ /* :244-250 */
    double ret = 0;
    for (int i = 0; i < m; i++) ret += a[i] * b[i];
    return ret;
}

double orc_util_maxabs(const double* a, int m)
{ /* :253-265 */
    double mx = 0, mn = 0;
    for (int i = 0; i < m; i++) {
        if (a[i] > mx) mx = a[i];
        else if (a[i] < mn) mn = a[i];
    }
    return mx > -mn ? mx : -mn;
}

/* ------------------------------------------------------------------------------------------------
 * Small dense kernels for the QP subsolver (no reference counterpart: this arithmetic lives in
 * qpOASES in the reference and is replaced, not ported).
 * ---------------------------------------------------------------------------------------------- */
static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

/* in-place lower Cholesky of the n x n row-major matrix M (ld = n); reports the smallest pivot
 * (before sqrt). Returns -1 when a pivot is not positive. The strict upper triangle is zeroed. */
static int chol_lower(double* M, int n, double* minpiv)
{
    double mp = INFINITY;
    for (int j = 0; j < n; j++) {
        double d = M[j * n + j];
        for (int k = 0; k < j; k++) d -= M[j * n + k] * M[j * n + k];
        if (d < mp) mp = d;
        if (!(d > 0)) { if (minpiv) *minpiv = mp; return -1; }
        double ljj = sqrt(d);
        M[j * n + j] = ljj;
        for (int i = j + 1; i < n; i++) {
            double s = M[i * n + j];
            const double* ri = M + (size_t)i * n;
            const double* rj = M + (size_t)j * n;
            for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
            M[i * n + j] = s / ljj;
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) M[i * n + j] = 0.0;
    if (minpiv) *minpiv = mp;
    return 0;
}

static void trsv_lower(const double* L, int n, double* b) /* solves L y = b in place */
{
    for (int i = 0; i < n; i++) {
        double s = b[i];
        const double* r = L + (size_t)i * n;
        for (int k = 0; k < i; k++) s -= r[k] * b[k];
        b[i] = s / r[i];
    }
}

static void trsv_lower_t(const double* L, int n, double* b) /* solves L' x = b in place */
{
    for (int i = n - 1; i >= 0; i--) {
        double xi = b[i] / L[(size_t)i * n + i];
        b[i] = xi;
        const double* r = L + (size_t)i * n;
        for (int k = 0; k < i; k++) b[k] -= r[k] * xi;
    }
}

/* ------------------------------------------------------------------------------------------------
 * QP subsolver:  min 1/2 x'Qx + g'x  s.t.  lbA <= A x <= ubA,  lb <= x <= ub
 * in the place of qpOASES behind SubsolverQPOASES (src/SubsolverQPOASES.cpp:32-46,134-181).
 * Internally E = [A ; rows of I for variables with a finite bound], duals in OSQP sign
 * (Qx + g + E'y = 0); getSolution converts to the qpOASES layout/sign:
 * y[0:nV] box duals, y[nV:] row duals, Qx + g - A'y_A - y_box = 0.
 * ---------------------------------------------------------------------------------------------- */
enum { ST_INACT = 0, ST_LOWER = 1, ST_UPPER = 2, ST_EQ = 3 };

struct orc_qp {
    int nV, nC;
    double *Q, *A;
    orc_options_t opt;
    /* setup */
    int is_setup, mE, nfin;
    int k_ready;      /* LK exists (qp_build_K): the ADMM factor is built when the first ADMM iteration needs it, as on the device (lcqp_dev.hpp: qp_build_K) */
    int* boxidx;
    double *E, *Et, *l, *u, *rhov;
    double *rn;       /* |E_r|_2 of every row: scale of the rounding floor of a computed E_r x (qp_polish, the active-row test) */
    double scale, sigma, spv, rho;
    double *L1, *LK;
    /* persistent solver state (hot start) */
    double *x, *y; /* last solution (y: OSQP sign, length mE) */
    int* st;       /* last active set */
    int have_solution;
    /* ADMM state */
    double *xa, *ya, *za;
    /* scratch */
    double *w_n1, *w_n2, *w_n3, *w_n4, *w_m1, *w_m2, *w_a1, *w_a2, *w_a3, *w_a4;
    /* The working-set system S dy = t, S = Et_W Et_W', is solved with an inverse factor that is UPDATED when rows enter or
     * leave the working set W (what qpOASES does with its factors on a hot start, src/SubsolverQPOASES.cpp:158):
     *   Ti (nT rows x ns slots) with Ti'Ti = inv(S_W);  slot_row[s] = row of E held by slot s (-1: free), row_slot = its inverse,
     *   crow[s] = the row of Ti that was appended together with slot s (column s is zero in the rows above it);
     *   the entries of S are dot products of rows of Et (the device reads them from M = Et Et', built once at setup by k_build_M). */
    double *Ti;
    int *slot_row, *row_slot, *crow;
    int nT, ns;
    double upd_bytes;   /* bytes of Ti read or written by updates (device byte accounting) */
    double *r1_last, *ex_last, *g_last; /* residual (of the QP as given, without the proximal term), E x and linear term of the last verified solution */
    double *xref;       /* anchor of the proximal term: the point the solve started from (x0, or the previous solution on a hot start) */
    int *newst;
    double *dy_last, *dx_last; /* change of (ya, xa) in the last ADMM iteration: OSQP's infeasibility certificates */
    /* dependent-row rules of the single-QP path (SubsolverHIP / k_qp_solve); the batched homotopy kernel runs without them */
    int robust;
    int *dep;      /* per row: 1 = flagged linearly dependent by the last factorisation of S, 2 = left because of it */
    int *prio;     /* per row: 0, or the stamp of the trial that promoted the row to the front of the active list */
    int prio_ctr;
    int cap_na;
    /* outputs */
    double *xsol, *ysol;
    /* counters */
    int c_admm, c_trials, c_fact, c_corr, c_sweeps;
    int c_pred, c_trsv;            /* predicted corrections; triangular solves with L1 */
    double rows_swept, rows_corr;  /* rows of E read by the residual sweeps; rows of Et read by the corrections */
};

orc_qp_t* orc_qp_create(int nV, int nC, const double* Q, const double* A, const orc_options_t* opt)
{
    orc_qp_t* q = (orc_qp_t*)calloc(1, sizeof(*q));
    q->nV = nV; q->nC = nC;
    q->Q = dalloc((size_t)nV * nV);
    q->A = dalloc((size_t)nC * nV);
    memcpy(q->Q, Q, sizeof(double) * nV * nV);           /* deep copy: SubsolverQPOASES.cpp:41-45 */
    if (nC > 0) memcpy(q->A, A, sizeof(double) * nC * nV);
    if (opt) q->opt = *opt; else orc_options_default(&q->opt);
    q->robust = 1;
    q->xsol = dalloc(nV);
    q->ysol = dalloc((size_t)nV + nC);
    return q;
}

static void qp_free_setup(orc_qp_t* q)
{
    free(q->boxidx); free(q->E); free(q->Et); free(q->l); free(q->u); free(q->rhov); free(q->rn); q->rn = NULL;
    free(q->L1); free(q->LK); free(q->x); free(q->y); free(q->st); free(q->xa); free(q->ya); free(q->za);
    free(q->xref); free(q->w_n1); free(q->w_n2); free(q->w_n3); free(q->w_n4); free(q->w_m1); free(q->w_m2); free(q->w_a1); free(q->w_a2); free(q->w_a3); free(q->w_a4);
    free(q->Ti); free(q->slot_row); free(q->row_slot); free(q->crow);
    q->Ti = NULL; q->slot_row = q->row_slot = q->crow = NULL;
    free(q->dy_last); free(q->dx_last); q->dy_last = q->dx_last = NULL;
    free(q->newst); free(q->dep); free(q->prio); q->dep = q->prio = NULL; free(q->r1_last); free(q->ex_last); free(q->g_last);
    q->r1_last = q->ex_last = q->g_last = NULL;
    q->boxidx = NULL; q->E = q->Et = q->l = q->u = q->rhov = q->L1 = q->LK = q->x = q->y = NULL;
    q->st = NULL; q->xa = q->ya = q->za = q->w_n1 = q->w_n2 = q->w_n3 = q->w_n4 = q->w_m1 = q->w_m2 = q->w_a1 = q->w_a2 = q->w_a3 = q->w_a4 = NULL;
    q->newst = NULL;
    q->is_setup = 0;
}

void orc_qp_destroy(orc_qp_t* q)
{
    if (!q) return;
    qp_free_setup(q);
    free(q->Q); free(q->A); free(q->xsol); free(q->ysol);
    free(q);
}

void orc_qp_get_counters(orc_qp_t* q, int* admm, int* trials, int* facts, int* corrections)
{
    if (admm) *admm = q->c_admm;
    if (trials) *trials = q->c_trials;
    if (facts) *facts = q->c_fact;
    if (corrections) *corrections = q->c_corr;
}

int orc_qp_get_sweeps(orc_qp_t* q) { return q->c_sweeps; }

static double bound_or(const double* b, int i, double dflt) { return b ? b[i] : dflt; }

/* One-time setup for a bound pattern: E, rho vector, the two constant factorisations
 * (L1 of Q + sp I for the polish, LK of Q + sigma I + E' diag(rho) E for ADMM), Et = E L1^-T.
 * Returns 0, or 3 when Q + sp I is not positive definite (non-convex QP). */
static int qp_setup(orc_qp_t* q, const double* lbA, const double* ubA, const double* lb, const double* ub)
{
    const int n = q->nV, nC = q->nC;
    const orc_options_t* o = &q->opt;
    qp_free_setup(q);
    q->boxidx = (int*)calloc(n ? n : 1, sizeof(int));
    int nfin = 0;
    for (int i = 0; i < n; i++) {
        double lo = bound_or(lb, i, -INFINITY), hi = bound_or(ub, i, INFINITY);
        if (isfinite(lo) || isfinite(hi)) q->boxidx[nfin++] = i;
    }
    q->nfin = nfin;
    const int mE = nC + nfin;
    q->mE = mE;
    q->E = dalloc((size_t)mE * n);
    q->Et = dalloc((size_t)mE * n);
    q->l = dalloc(mE); q->u = dalloc(mE); q->rhov = dalloc(mE);
    if (nC > 0) memcpy(q->E, q->A, sizeof(double) * nC * n);
    for (int k = 0; k < nfin; k++) q->E[(size_t)(nC + k) * n + q->boxidx[k]] = 1.0;
    for (int i = 0; i < nC; i++) { q->l[i] = bound_or(lbA, i, -INFINITY); q->u[i] = bound_or(ubA, i, INFINITY); }
    for (int k = 0; k < nfin; k++) {
        q->l[nC + k] = bound_or(lb, q->boxidx[k], -INFINITY);
        q->u[nC + k] = bound_or(ub, q->boxidx[k], INFINITY);
    }
    q->rn = dalloc(mE);
    for (int r = 0; r < mE; r++) { double s2 = 0; for (int k = 0; k < n; k++) s2 += q->E[(size_t)r * n + k] * q->E[(size_t)r * n + k]; q->rn[r] = sqrt(s2); }
    double scale = 0;
    for (int i = 0; i < n; i++) { double d = fabs(q->Q[(size_t)i * n + i]); if (d > scale) scale = d; }
    if (!(scale > 1e-300)) scale = 1.0;
    q->scale = scale;
    q->sigma = o->admmSigma * scale;
    q->rho = o->admmRho * scale;
    for (int i = 0; i < mE; i++) {
        if (isinf(q->l[i]) && isinf(q->u[i])) q->rhov[i] = 0.0;
        else if (q->l[i] == q->u[i]) q->rhov[i] = q->rho * o->rhoEqMult;
        else q->rhov[i] = q->rho;
    }
    /* L1 = chol(Q + sp I): try the small prox weight, fall back to the big one for PSD Hessians */
    q->L1 = dalloc((size_t)n * n);
    for (int pass = 0; pass < 2; pass++) {
        q->spv = (pass == 0 ? o->proxSmall : o->proxBig) * scale;
        memcpy(q->L1, q->Q, sizeof(double) * n * n);
        for (int i = 0; i < n; i++) q->L1[(size_t)i * n + i] += q->spv;
        double minpiv;
        int rc = chol_lower(q->L1, n, &minpiv);
        if (rc == 0 && (pass == 1 || minpiv >= o->pivotThreshold * scale)) break;
        if (pass == 1) return 3;
    }
    /* Et = E L1^-T : row r solves L1 t = e_r' */
    for (int r = 0; r < mE; r++) {
        double* t = q->Et + (size_t)r * n;
        memcpy(t, q->E + (size_t)r * n, sizeof(double) * n);
        trsv_lower(q->L1, n, t);
    }
    /* LK = chol(Q + sigma I + E' diag(rhov) E) is only needed when the active-set iteration fails (one instance in a hundred of the synthetic
     * workload): built on demand by qp_build_K, like the device does -- an eager build was half of a solve's time for nothing */
    q->LK = dalloc((size_t)n * n);
    q->k_ready = 0;

    q->x = dalloc(n); q->y = dalloc(mE); q->st = (int*)calloc(mE ? mE : 1, sizeof(int));
    q->xa = dalloc(n); q->ya = dalloc(mE); q->za = dalloc(mE);
    q->w_n1 = dalloc(n); q->w_n2 = dalloc(n); q->w_n3 = dalloc(n); q->w_n4 = dalloc(n); q->w_m1 = dalloc(mE); q->w_m2 = dalloc(mE);
    q->cap_na = (2 * n > 64) ? 2 * n : 64;   /* room for the degenerate vertices of small problems (many rows, few variables) */
    if (q->cap_na > mE) q->cap_na = mE;
    { const int lim = n > 2048 ? 3264 : (n > 1024 ? 2432 : (n > 512 ? 1216 : 896));   /* the device keeps the active-row solves in LDS (max_active(NCH)); binds for nV > 448 only */
      if (q->cap_na > lim) q->cap_na = lim; }
    q->w_a1 = dalloc(q->cap_na); q->w_a2 = dalloc(q->cap_na); q->w_a3 = dalloc(q->cap_na); q->w_a4 = dalloc(q->cap_na);
    q->newst = (int*)calloc(mE ? mE : 1, sizeof(int));
    q->dy_last = dalloc(mE); q->dx_last = dalloc(n);
    q->dep = (int*)calloc(mE ? mE : 1, sizeof(int));
    q->prio = (int*)calloc(mE ? mE : 1, sizeof(int));
    q->prio_ctr = 0;
    q->Ti = dalloc((size_t)q->cap_na * q->cap_na);
    q->slot_row = (int*)calloc(q->cap_na ? q->cap_na : 1, sizeof(int));
    q->crow = (int*)calloc(q->cap_na ? q->cap_na : 1, sizeof(int));
    q->row_slot = (int*)calloc(mE ? mE : 1, sizeof(int));
    for (int r = 0; r < mE; r++) q->row_slot[r] = -1;
    q->nT = q->ns = 0;
    q->r1_last = dalloc(n); q->ex_last = dalloc(mE); q->g_last = dalloc(n); q->xref = dalloc(n);
    q->have_solution = 0;
    q->is_setup = 1;
    return 0;
}

/* true when the new bounds keep the finite/equality pattern the factorisations were built for */
static int qp_bounds_compatible(orc_qp_t* q, const double* lbA, const double* ubA, const double* lb, const double* ub)
{
    const int n = q->nV, nC = q->nC;
    int k = 0;
    for (int i = 0; i < n; i++) {
        double lo = bound_or(lb, i, -INFINITY), hi = bound_or(ub, i, INFINITY);
        int fin = isfinite(lo) || isfinite(hi);
        int was = (k < q->nfin && q->boxidx[k] == i);
        if (fin != was) return 0;
        if (was) k++;
    }
    for (int i = 0; i < q->mE; i++) {
        double lo, hi;
        if (i < nC) { lo = bound_or(lbA, i, -INFINITY); hi = bound_or(ubA, i, INFINITY); }
        else { lo = bound_or(lb, q->boxidx[i - nC], -INFINITY); hi = bound_or(ub, q->boxidx[i - nC], INFINITY); }
        int free_new = isinf(lo) && isinf(hi), free_old = (q->rhov[i] == 0.0);
        int eq_new = (lo == hi), eq_old = (q->l[i] == q->u[i]);
        if (free_new != free_old || eq_new != eq_old) return 0;
    }
    return 1;
}

static void qp_update_bounds(orc_qp_t* q, const double* lbA, const double* ubA, const double* lb, const double* ub)
{
    const int nC = q->nC;
    for (int i = 0; i < q->mE; i++) {
        if (i < nC) { q->l[i] = bound_or(lbA, i, -INFINITY); q->u[i] = bound_or(ubA, i, INFINITY); }
        else { q->l[i] = bound_or(lb, q->boxidx[i - nC], -INFINITY); q->u[i] = bound_or(ub, q->boxidx[i - nC], INFINITY); }
    }
}

static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* LK = chol(Q + sigma I + E' diag(rhov) E); returns non-zero when a pivot is not positive (device: qp_build_K) */
static int qp_build_K(orc_qp_t* q)
{
    const int n = q->nV, mE = q->mE;
    memcpy(q->LK, q->Q, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) q->LK[(size_t)i * n + i] += q->sigma;
    for (int r = 0; r < mE; r++) {
        const double rv = q->rhov[r];
        if (rv == 0.0) continue;
        const double* e = q->E + (size_t)r * n;
        for (int i = 0; i < n; i++) {
            const double ei = rv * e[i];
            if (ei == 0.0) continue;
            double* row = q->LK + (size_t)i * n;
            for (int j = 0; j < n; j++) row[j] += ei * e[j];
        }
    }
    const int rc = chol_lower(q->LK, n, NULL);
    q->k_ready = (rc == 0);
    return rc;
}

/* n_it ADMM iterations on (xa, za, ya) with the constant factor LK (OSQP iteration, reduced KKT form):
 *   xt = K^-1 (sigma x - g + E'(rho.z - y));  zt = E xt;  relaxation alpha;  z = clip;  y += rho (zr - z). */
static void qp_admm(orc_qp_t* q, const double* g, int n_it)
{
    const int n = q->nV, mE = q->mE;
    const double alpha = q->opt.admmAlpha, sigma = q->sigma;
    double* rhs = q->w_n1;
    for (int it = 0; it < n_it; it++) {
        for (int i = 0; i < n; i++) rhs[i] = sigma * q->xa[i] - g[i];
        for (int r = 0; r < mE; r++) {
            double v = q->rhov[r] * q->za[r] - q->ya[r];
            if (v == 0.0) continue;
            const double* e = q->E + (size_t)r * n;
            for (int i = 0; i < n; i++) rhs[i] += e[i] * v;
        }
        trsv_lower(q->LK, n, rhs);
        trsv_lower_t(q->LK, n, rhs); /* rhs = xt */
        for (int r = 0; r < mE; r++) {
            const double* e = q->E + (size_t)r * n;
            double zt = 0;
            for (int i = 0; i < n; i++) zt += e[i] * rhs[i];
            double zr = alpha * zt + (1.0 - alpha) * q->za[r];
            double rv = q->rhov[r];
            const double yold = q->ya[r];
            if (rv > 0.0) {
                double zn = clipd(zr + q->ya[r] / rv, q->l[r], q->u[r]);
                q->ya[r] += rv * (zr - zn);
                q->za[r] = zn;
            } else {
                q->za[r] = zr;
                q->ya[r] = 0.0;
            }
            if (it == n_it - 1) q->dy_last[r] = q->ya[r] - yold;
        }
        for (int i = 0; i < n; i++) {
            const double xold = q->xa[i];
            q->xa[i] = alpha * rhs[i] + (1.0 - alpha) * q->xa[i];
            if (it == n_it - 1) q->dx_last[i] = q->xa[i] - xold;
        }
        q->c_admm++;
    }
}

/* OSQP's certificates from the last ADMM step (Stellato et al., Math. Prog. Comp. 12, 2020, section 3.4), relative
 * tolerance 1e-4 (OSQP's default): primal infeasibility from dy = y_k - y_{k-1}, unboundedness from dx = x_k - x_{k-1}.
 * Returns 4 (infeasible), 5 (unbounded) -- the exit flags -- or 0. */
#define ORC_CERT_EPS 1e-4
static int qp_certificate(orc_qp_t* q, const double* g)
{
    const int n = q->nV, mE = q->mE;
    const double eps = ORC_CERT_EPS;
    double ny = 0;
    for (int r = 0; r < mE; r++) if (fabs(q->dy_last[r]) > ny) ny = fabs(q->dy_last[r]);
    if (ny > 1e-30) {
        double sup = 0; int bad = 0;
        for (int r = 0; r < mE; r++) {
            const double d = q->dy_last[r];
            if (d > 0) { if (!isfinite(q->u[r])) { if (d > eps * ny) bad = 1; } else sup += q->u[r] * d; }
            else if (d < 0) { if (!isfinite(q->l[r])) { if (-d > eps * ny) bad = 1; } else sup += q->l[r] * d; }
        }
        if (!bad && sup <= -eps * ny) {
            double* t = q->w_n2;
            for (int i = 0; i < n; i++) t[i] = 0;
            for (int r = 0; r < mE; r++) {
                const double d = q->dy_last[r];
                if (d == 0) continue;
                const double* e = q->E + (size_t)r * n;
                for (int i = 0; i < n; i++) t[i] += e[i] * d;
            }
            double na = 0;
            for (int i = 0; i < n; i++) if (fabs(t[i]) > na) na = fabs(t[i]);
            if (na <= eps * ny) return 4;
        }
    }
    double nx = 0;
    for (int i = 0; i < n; i++) if (fabs(q->dx_last[i]) > nx) nx = fabs(q->dx_last[i]);
    if (nx > 1e-30) {
        double gd = 0;
        for (int i = 0; i < n; i++) gd += g[i] * q->dx_last[i];
        if (gd <= -eps * nx) {
            double nq = 0;
            for (int i = 0; i < n; i++) {
                const double* qr = q->Q + (size_t)i * n;
                double sdot = 0;
                for (int k = 0; k < n; k++) sdot += qr[k] * q->dx_last[k];
                if (fabs(sdot) > nq) nq = fabs(sdot);
            }
            if (nq <= eps * nx) {
                int ok = 1;
                for (int r = 0; r < mE; r++) {
                    const double* e = q->E + (size_t)r * n;
                    double sdot = 0;
                    for (int k = 0; k < n; k++) sdot += e[k] * q->dx_last[k];
                    if (isfinite(q->u[r]) && sdot > eps * nx) ok = 0;
                    if (isfinite(q->l[r]) && sdot < -eps * nx) ok = 0;
                }
                if (ok) return 5;
            }
        }
    }
    return 0;
}

/* OSQP's rho adaptation (Stellato et al. 2020, section 5.2) for the fallback rounds: after a failed round, scale all rho_i by
 *   sqrt( (|E xa - za| / max(|E xa|, |za|)) / (|Q xa + g + E'ya| / max(|Q xa|, |E'ya|, |g|)) )      (infinity norms, clipped to
 * [1e-3, 1e3]) when that factor is above 5 or below 1/5, and refactorise K = Q + sigma I + E' diag(rho) E.  A fixed rho makes
 * ADMM crawl on badly scaled QPs (large penalty parameters, degenerate vertices) and the polish then never gets a usable
 * working-set guess.  Returns 1 when rho changed. */
static int qp_adapt_rho(orc_qp_t* q, const double* g)
{
    const int n = q->nV, mE = q->mE;
    double *t = q->w_n2, *t2 = q->w_n3;
    double rp = 0, rd = 0, nz = 0, nax = 0, nq = 0, naty = 0, gm = 0;
    for (int i = 0; i < n; i++) {
        const double* qr = q->Q + (size_t)i * n;
        double sd = 0;
        for (int k = 0; k < n; k++) sd += qr[k] * q->xa[k];
        t[i] = sd; t2[i] = 0;
        if (fabs(sd) > nq) nq = fabs(sd);
        if (fabs(g[i]) > gm) gm = fabs(g[i]);
    }
    for (int r = 0; r < mE; r++) {
        const double* e = q->E + (size_t)r * n;
        double sd = 0;
        for (int k = 0; k < n; k++) sd += e[k] * q->xa[k];
        if (fabs(sd - q->za[r]) > rp) rp = fabs(sd - q->za[r]);
        if (fabs(sd) > nax) nax = fabs(sd);
        if (fabs(q->za[r]) > nz) nz = fabs(q->za[r]);
        const double yr = q->ya[r];
        if (yr != 0.0) for (int k = 0; k < n; k++) t2[k] += e[k] * yr;
    }
    for (int i = 0; i < n; i++) {
        const double v = t[i] + g[i] + t2[i];
        if (fabs(v) > rd) rd = fabs(v);
        if (fabs(t2[i]) > naty) naty = fabs(t2[i]);
    }
    const double num = rp / fmax(fmax(nax, nz), 1e-30), den = rd / fmax(fmax(fmax(nq, naty), gm), 1e-30);
    double fac = sqrt(num / fmax(den, 1e-30));
    if (fac > 1e3) fac = 1e3;
    if (fac < 1e-3) fac = 1e-3;
    if (!(fac > 5.0 || fac < 0.2)) return 0;
    for (int r = 0; r < mE; r++) q->rhov[r] *= fac;
    q->rho *= fac;
    (void)qp_build_K(q);
    q->c_fact++;
    return 1;
}

/* active-set guess from an ADMM iterate (the rule OSQP's polish uses) */
static void qp_guess_from_admm(orc_qp_t* q, int* st)
{
    for (int r = 0; r < q->mE; r++) {
        double l = q->l[r], u = q->u[r], z = q->za[r], y = q->ya[r];
        int s = ST_INACT;
        if (isfinite(l) && (z - l < -y)) s = ST_LOWER;
        if (isfinite(u) && (u - z < y)) s = ST_UPPER;
        if (l == u) s = ST_EQ;
        st[r] = s;
    }
}

/* ---- inverse factor of the working-set matrix (see struct orc_qp) --------------------------------------------------------- */
static void ti_reset(orc_qp_t* q)
{
    for (int s = 0; s < q->ns; s++) if (q->slot_row[s] >= 0) q->row_slot[q->slot_row[s]] = -1;
    q->nT = q->ns = 0;
}

/* dy = Ti' (Ti t) over the slots in use (t is zero on free slots) */
static void ti_apply(const orc_qp_t* q, const double* t, double* dy)
{
    const int ld = q->cap_na, ns = q->ns;
    for (int s = 0; s < ns; s++) dy[s] = 0.0;
    for (int j = 0; j < q->nT; j++) {
        const double* row = q->Ti + (size_t)j * ld;
        double u = 0;
        for (int s = 0; s < ns; s++) u += row[s] * t[s];
        for (int s = 0; s < ns; s++) dy[s] += row[s] * u;
    }
}

/* Row r enters: with s = S_W,r (entries of M), w = inv(S_W) s and delta^2 = S_rr - s'w (the Schur complement = the squared
 * Cholesky pivot of the row behind the rows of W), the new last row of Ti is [-w'/delta, 1/delta].  delta^2 <= tau S_rr: the row
 * is linearly dependent on W -- it stays out of the factor, its multiplier is not moved and its equation not enforced
 * (returns 0).  Returns 1 when appended, -1 when the factor is full. */
static int ti_append(orc_qp_t* q, int r, double tau)
{
    const int ld = q->cap_na, n = q->nV;
    double *sv = q->w_a1, *w = q->w_a3;
    const double* tr = q->Et + (size_t)r * n;
    for (int s = 0; s < q->ns; s++) {
        double sdot = 0;
        if (q->slot_row[s] >= 0) { const double* ts = q->Et + (size_t)q->slot_row[s] * n; for (int k = 0; k < n; k++) sdot += tr[k] * ts[k]; }
        sv[s] = sdot;
    }
    ti_apply(q, sv, w);
    double srr = 0;
    for (int k = 0; k < n; k++) srr += tr[k] * tr[k];
    double d2 = srr;
    for (int s = 0; s < q->ns; s++) d2 -= sv[s] * w[s];
    q->upd_bytes += 8.0 * ((double)q->nT * q->ns + 2.0 * q->ns);
    if (!(d2 > tau * srr) || !(d2 > 0.0)) return 0;
    int snew = -1;
    for (int s = 0; s < q->ns; s++) if (q->slot_row[s] < 0) { snew = s; break; }
    if (snew < 0) {
        if (q->ns >= q->cap_na) return -1;
        snew = q->ns++;
        for (int j = 0; j < q->nT; j++) q->Ti[(size_t)j * ld + snew] = 0.0;      /* a fresh column */
    }
    const double delta = sqrt(d2);
    double* row = q->Ti + (size_t)q->nT * ld;
    for (int s = 0; s < q->ns; s++) row[s] = (q->slot_row[s] >= 0) ? -w[s] / delta : 0.0;
    row[snew] = 1.0 / delta;
    q->slot_row[snew] = r; q->row_slot[r] = snew; q->crow[snew] = q->nT;
    q->nT++;
    return 1;
}

/* The row held by slot p leaves: rotations of neighbouring rows of Ti, from the row that created the slot downwards, move
 * column p into the last row, which is dropped (Ti'Ti then is the inverse of S without row and column p).  The rotations
 * follow from column p alone: rho_j = hypot(rho_{j-1}, Ti[j][p]). */
static void ti_delete(orc_qp_t* q, int p)
{
    const int ld = q->cap_na, ns = q->ns, i0 = q->crow[p], nT = q->nT;
    double* carry = q->w_a1;
    memcpy(carry, q->Ti + (size_t)i0 * ld, sizeof(double) * ns);
    double rho = carry[p];
    for (int j = i0 + 1; j < nT; j++) {
        double* rj = q->Ti + (size_t)j * ld;
        const double b = rj[p], rn = hypot(rho, b);
        double cs = 1.0, sn = 0.0;
        if (rn > 0.0) { cs = b / rn; sn = -rho / rn; }
        double* out = q->Ti + (size_t)(j - 1) * ld;
        for (int s = 0; s < ns; s++) {
            const double cv = carry[s], rv = rj[s];
            out[s] = cs * cv + sn * rv;
            carry[s] = cs * rv - sn * cv;
        }
        out[p] = 0.0;
        rho = rn;
    }
    q->upd_bytes += 8.0 * 2.0 * (double)(nT - i0) * ns;
    const int r = q->slot_row[p];
    q->slot_row[p] = -1; q->row_slot[r] = -1;
    q->nT = nT - 1;
    for (int s = 0; s < ns; s++) if (q->slot_row[s] >= 0 && q->crow[s] > i0) q->crow[s]--;
    while (q->ns > 0 && q->slot_row[q->ns - 1] < 0) q->ns--;      /* free slots at the end are given back */
}


/* Dot product in the summation order of the device's row sweeps (wg_rows, lcqp_wg.hpp): 64 lanes, lane l sums the columns
 * 128k + 2l, 128k + 2l + 1 of every 128-column chunk k, then a butterfly over the lanes (offsets 32, 16, ... 1).
 * Why the oracle's QP solver sums E x this way (round 3): at the end of every inner loop of the homotopy the step p = xnew - xk
 * is ~1e-8 and getOptimalStepLength (src/LCQProblem.cpp:1217-1237) divides lk = p'(Qk xk + g~) ~ -1e-16 by qk = p'Qk p ~ 1e-16.
 * lk contains (E_a p)'y, which is zero in exact arithmetic and in floating point is y' times the rounding noise of the
 * feasibility residual b - E_a x that the last correction removed, plus the rounding of x itself (~4e-17, unavoidable with x
 * stored in doubles -- any subsolver has it, qpOASES included).  So whether alpha comes out as 1 or as 0.8 there is a coin flip,
 * and one flip moves a penalty update by one cycle of nDynamicPenalty + 1 iterates.  With a left-to-right sum the noise of E x is
 * about four times that of the device's tree sum: the oracle's lk then lands far outside (-qk, 0) and alpha = 1 almost always,
 * the device's lands inside it a quarter of the time -- a 4:1 bias in the iterate counts (1990 instances +4, 484 -4 of 8192)
 * although both end in the same point to 2e-14.  With the same summation order the noise has the same size on both sides and the
 * histogram is symmetric (tools/cmp_iters.py).  orc_qp_set_sum_order(0) restores the left-to-right sum
 * (tests/test_oracle_solver.py uses it to show that the oracle differs from ITSELF in the same way when only this order changes). */
static int g_sum_order = 1;
void orc_qp_set_sum_order(int device_order) { g_sum_order = device_order; }
/* divisor of the cap on entering rows of a cold polish (qp_polish): max(n / div, 16) rows per trial; 0 switches the cap off (test hook:
 * tests/test_oracle_solver.py::test_capped_cold_start_reaches_the_same_qp_solutions).  The device uses 8. */
static int g_enter_cap_div = 8;
void orc_qp_set_enter_cap(int div) { g_enter_cap_div = div; }
static double dot_lanes(const double* a, const double* x, int n)
{
    double v[64];
    for (int l = 0; l < 64; l++) {
        double s = 0;
        for (int k0 = 0; k0 < n; k0 += 128) {
            const int c = k0 + 2 * l;
            if (c < n) s += a[c] * x[c];
            if (c + 1 < n) s += a[c + 1] * x[c + 1];
        }
        v[l] = s;
    }
    for (int o = 32; o > 0; o >>= 1)
        for (int l = 0; l < o; l++) v[l] += v[l + o];
    return v[0];
}

/* Primal-dual active-set polish in correction (iterative refinement) form.
 * Start: x, yfull (OSQP sign, zero on inactive rows), active set st.  Returns 1 on a verified KKT point.
 *
 * Round 3: a trial no longer evaluates the whole KKT residual before it knows whether it needs it.  After a correction the state
 * is known up to rounding: the stationarity residual is sigma_p * dx (~1e-12 relative, taken as 0), and the rows in the factor sit
 * on their bounds.  So a trial is
 *   (a)  rows whose multiplier has the wrong sign leave (their multipliers are remembered in ylv);
 *   (b)  STAGE 1: E x of the inactive rows only (and of active rows flagged dependent) -- on the device only the rows the
 *        screening cannot rule out; violated rows enter;
 *   (c)  nothing changed: STAGE 2, the true residual -- one pass over Q and the active rows of E; the point is accepted on
 *        these true residuals only (so accuracy is what it was), else a full correction with them follows;
 *   (d)  the set changed: the factor follows, then a PREDICTED correction: r1 = sum over the rows that left of y_r e_r, hence
 *        c = L1^-1 r1 = sum y_r Et_r (no forward solve), Et_W c from entries of M = Et Et' on the device, r2 = 0 on the rows that
 *        were in the factor and b - E x on the rows that entered; dx = L1^-T (c - Et_W' dy): one pass over Et_W, one backward solve.
 * Intermediate trials read neither Q nor the active rows of E.
 * What is solved is the PROXIMAL QP  min 1/2 x'Qx + g'x + sigma_p/2 |x - xref|^2  with xref = the point the solve started from and
 * sigma_p the weight the constant factor L1 = chol(Q + sigma_p I) carries anyway (1e-12 max|Q_ii|; 1e-8 max|Q_ii| when Q is only PSD):
 * the corrections are then exact Newton steps of the problem whose residual is tested (the predicted residual after a correction is
 * exactly zero), and the QP has ONE solution also when Q has flat directions -- the minimiser nearest xref up to O(sigma_p) --,
 * whatever path (working sets, ADMM rounds) leads there.  A solution of the proximal QP is returned when it also satisfies the QP as
 * given to the tolerance (sigma_p |x - xref| <= resTol (1 + |g|): always so for sigma_p = 1e-12 max|Q_ii|); otherwise xref moves to
 * it and the iteration continues -- the proximal-point method, every step of which has a unique solution.
 * q->robust: rows the safeguarded factor update flags as linearly dependent get two extra rules -- strictly inside the bound:
 * the row leaves; violated: the row is promoted to the front of the ordered active list, so that another row becomes the
 * dependent one.
 * reuse != 0 (hot start from the last verified solution): the first trial needs no sweep --
 * r1 = r1_last + (g_last - g) and E x = ex_last hold exactly for an unchanged (x, y). */
/* how far E_r x lies outside [l, u] beyond the feasibility tolerance (0: not violated) */
static double row_violation(double e, double l, double u, double feasTol)
{
    const double ftol = feasTol * (1.0 + fabs(e));
    if (e < l - ftol) return l - e;
    if (e > u + ftol) return e - u;
    return 0.0;
}

/* Round 5, two additions (device: qp_polish in lcqp_dev.hpp, the same arithmetic):
 *  (1) ACTIVE ROWS TO THEIR ROUNDING FLOOR.  The homotopy ends on phi < complementarityTolerance = 1e3 eps (src/LCQProblem.cpp:511-534,
 *      src/Options.cpp:297), a sum of products (L_i x - lbL_i)(R_i x - lbR_i) in which one factor is the residual of an ACTIVE row of this QP.
 *      An active-set solver like the reference's holds its active rows to rounding; a residual test at resTol (1 + |b|) = 1e-12 does not.
 *      A point that passes the residual tests is therefore accepted only when every row of the factor also holds to
 *      16 eps (|b_r| + |E_r| |x|) -- the rounding of a computed E_r x --, else one more correction (iterative refinement, at most two per
 *      polish) is taken first.  On well-conditioned QPs the rows are at that floor after every correction and nothing changes.
 *  (2) DAMPED TRIALS.  damp != 0 (the rounds after g_damp_round failed ones): the primal-dual update -- all wrong-signed rows out, all
 *      violated rows in -- thrashes on LP-like QPs (singular Hessian, |g| ~ 1e7 at the end of a penalty homotopy); a damped polish
 *      changes ONE row per trial: the row with the largest wrong-signed multiplier leaves, else the most violated row enters, and
 *      it may take 4 n + 32 more trials. */
static int g_damp_round = 3;
void orc_qp_set_damp_round(int r) { g_damp_round = r; }
/* diagnostic: one line per round / trial on stderr (the device prints the same lines in a -DLCQP_TRACE_QP build; tools/fuzz_case.py --trace) */
static int g_trace_qp = 0;
void orc_qp_set_trace(int on) { g_trace_qp = on; }
static int qp_polish(orc_qp_t* q, const double* g, double* x, double* yfull, int* st, int reuse, int damp)
{
    const int n = q->nV, mE = q->mE, robust = q->robust;
    const orc_options_t* o = &q->opt;
    double gmax = 0;
    for (int i = 0; i < n; i++) { double a = fabs(g[i]); if (a > gmax) gmax = a; }
    const double gs = 1.0 + gmax;
    const double ytol = o->feasTol * gs;
    double *r1 = q->w_n1, *c = q->w_n2, *du = q->w_n3, *Ex = q->w_m1, *ylv = q->w_m2;
    double* dy = q->w_a4;
    int fact_valid = 0;
    const double spv = q->spv;
    for (int r = 0; r < mE; r++) ylv[r] = 0.0;
    /* the cap on entering rows (below) is for polishes that start from an EMPTY working set: with a guess in hand -- the previous QP's set,
     * the one ADMM proposes -- the full primal-dual update is the better step (and degenerate problems such as example_data, where rows
     * flagged dependent are re-tried every trial, need it) */
    int cap_on = 0;
    if (!reuse && g_enter_cap_div > 0) { cap_on = 1; for (int r = 0; r < mE; r++) if (st[r] != ST_INACT) cap_on = 0; }

    const int maxTrials = damp ? o->maxTrials + 4 * n + 32 : o->maxTrials;
    int nrefine = 0;
    for (int trial = 0; trial < maxTrials; trial++) {
        q->c_trials++;
        int changed = 0, promoted = 0, nlv = 0, have_true = 0;
        if (trial == 0) {
            if (reuse) {
                for (int i = 0; i < n; i++) r1[i] = (q->r1_last[i] + (q->g_last[i] - g[i])) - spv * (x[i] - q->xref[i]);
                memcpy(Ex, q->ex_last, sizeof(double) * mE);
            } else {
                /* cold: the whole residual, every row of E */
                q->c_sweeps++; q->rows_swept += mE;
                if (robust) for (int r = 0; r < mE; r++) if (st[r] == ST_INACT) yfull[r] = 0.0;
                for (int i = 0; i < n; i++) {
                    const double* qr = q->Q + (size_t)i * n;
                    double s = 0;
                    for (int k = 0; k < n; k++) s += qr[k] * x[k];
                    r1[i] = -g[i] - s;
                }
                for (int r = 0; r < mE; r++) {
                    const double* e = q->E + (size_t)r * n;
                    if (g_sum_order) Ex[r] = dot_lanes(e, x, n);
                    else { double s = 0; for (int k = 0; k < n; k++) s += e[k] * x[k]; Ex[r] = s; }
                    double yr = yfull[r];
                    if (yr != 0.0)
                        for (int k = 0; k < n; k++) r1[k] -= e[k] * yr;
                }
                for (int i = 0; i < n; i++) r1[i] -= spv * (x[i] - q->xref[i]);
            }
            have_true = 1;      /* the first trial only corrects: the working set it was handed stays */
        } else {
            /* (a) leaving rows (damped: the one with the largest wrong-signed multiplier, the lowest index among equals) */
            double ylvmax = 0.0;
            if (damp)
                for (int r = 0; r < mE; r++) {
                    const int s = st[r];
                    if (((s == ST_LOWER && yfull[r] > ytol) || (s == ST_UPPER && yfull[r] < -ytol)) && fabs(yfull[r]) > ylvmax) ylvmax = fabs(yfull[r]);
                }
            for (int r = 0; r < mE; r++) {
                const int s = st[r];
                if ((s == ST_LOWER && yfull[r] > ytol) || (s == ST_UPPER && yfull[r] < -ytol)) {
                    if (damp && (nlv > 0 || fabs(yfull[r]) < ylvmax)) continue;
                    ylv[r] = yfull[r]; yfull[r] = 0.0; st[r] = ST_INACT; nlv++; changed = 1;
                }
            }
            /* (b) stage 1: E x of the inactive rows and of the active rows flagged dependent */
            for (int r = 0; r < mE; r++) {
                if (!(st[r] == ST_INACT || (robust && q->dep[r]))) continue;
                const double* e = q->E + (size_t)r * n;
                if (g_sum_order) Ex[r] = dot_lanes(e, x, n);
                else { double s = 0; for (int k = 0; k < n; k++) s += e[k] * x[k]; Ex[r] = s; }
                q->rows_swept++;
            }
            /* Entering rows are capped (round 3): when more than max(n/8, 16) inactive rows are violated -- a cold start, where the
             * primal-dual update would put every violated row into the working set at once, overshoot, and oscillate for eight to ten
             * trials with a factor rebuild each -- only the most violated ones enter: those at or above a cut found by twelve bisection
             * steps on [0, largest violation] (the same arithmetic on the device: qp_polish in lcqp_dev.hpp).  The others stay
             * inactive and are looked at again in the next trial.  On the synthetic workload the first QP of a homotopy then takes fewer
             * trials (8.4 -> 6.4 with n/4) and fewer, smaller rebuilds; n/8 is the device's optimum (same-box A/B of n/3 ... n/16). */
            double vcut = 0.0;
            if (cap_on) {
                const int cap = (n / g_enter_cap_div > 16) ? n / g_enter_cap_div : 16;
                int nviol = 0; double vmax = 0.0;
                for (int r = 0; r < mE; r++) if (st[r] == ST_INACT) {
                    const double v = row_violation(Ex[r], q->l[r], q->u[r], o->feasTol);
                    if (v > 0.0) { nviol++; if (v > vmax) vmax = v; }
                }
                if (nviol > cap) {
                    double lo = 0.0, hi = vmax;
                    for (int it = 0; it < 12; it++) {
                        const double mid = 0.5 * (lo + hi);
                        int cnt = 0;
                        for (int r = 0; r < mE; r++) if (st[r] == ST_INACT) cnt += (row_violation(Ex[r], q->l[r], q->u[r], o->feasTol) >= mid);
                        if (cnt > cap) lo = mid; else hi = mid;
                    }
                    vcut = lo;      /* the lower end: a few more than cap rows -- with the upper end a tie of many equally violated rows would never enter */
                }
            }
            int nent = 0;
            if (damp) {      /* one change per trial: the most violated row enters (the lowest index among equals), and only when no row left */
                vcut = INFINITY;
                if (!changed) {
                    double vm = 0.0;
                    for (int r = 0; r < mE; r++) if (st[r] == ST_INACT) { const double v = row_violation(Ex[r], q->l[r], q->u[r], o->feasTol); if (v > vm) vm = v; }
                    if (vm > 0.0) vcut = vm;
                }
            }
            for (int r = 0; r < mE; r++) {
                const int s = st[r];
                const double ftol = o->feasTol * (1.0 + fabs(Ex[r]));
                if (s == ST_INACT) {
                    if (damp && nent > 0) continue;
                    if (Ex[r] < q->l[r] - ftol) { if (q->l[r] - Ex[r] >= vcut) { st[r] = ST_LOWER; changed = 1; nent++; } }
                    else if (Ex[r] > q->u[r] + ftol) { if (Ex[r] - q->u[r] >= vcut) { st[r] = ST_UPPER; changed = 1; nent++; } }
                } else if (robust && q->dep[r]) {
                    /* the last factor update flagged this row as dependent on the rows before it, so the correction left its
                     * multiplier alone and did not enforce its equation.  Strictly inside its bound: the row is not active.
                     * Violated: it must be active, so it moves to the front of the list and a different row becomes the
                     * dependent one. */
                    int viol, inside = 0;
                    if (s == ST_LOWER) { viol = Ex[r] < q->l[r] - ftol; inside = Ex[r] > q->l[r] + ftol; }
                    else if (s == ST_UPPER) { viol = Ex[r] > q->u[r] + ftol; inside = Ex[r] < q->u[r] - ftol; }
                    else viol = fabs(Ex[r] - q->l[r]) > ftol;
                    if (inside) { ylv[r] = yfull[r]; yfull[r] = 0.0; st[r] = ST_INACT; nlv += (ylv[r] != 0.0); changed = 1; }
                    else if (viol) { q->prio[r] = q->prio_ctr + 1; promoted = 1; }
                }
            }
            if (promoted) { q->prio_ctr++; changed = 1; }
            if (!changed) {
                /* (c) stage 2: the true residual -- Q and the active rows of E */
                q->c_sweeps++;
                for (int i = 0; i < n; i++) {
                    const double* qr = q->Q + (size_t)i * n;
                    double s = 0;
                    for (int k = 0; k < n; k++) s += qr[k] * x[k];
                    q->w_n4[i] = s;
                    r1[i] = -g[i] - s;
                }
                double res_stat = 0, res_eq = 0, bmax = 0, xn = 0, sc = 0;
                int nloose = 0;      /* rows of the factor that are not at the rounding floor of E_r x */
                for (int i = 0; i < n; i++) xn += x[i] * x[i];
                xn = sqrt(xn);
                for (int r = 0; r < mE; r++) {
                    if (st[r] == ST_INACT) continue;
                    const double* e = q->E + (size_t)r * n;
                    if (g_sum_order) Ex[r] = dot_lanes(e, x, n);
                    else { double s = 0; for (int k = 0; k < n; k++) s += e[k] * x[k]; Ex[r] = s; }
                    q->rows_swept++;
                    const double yr = yfull[r];
                    if (yr != 0.0)
                        for (int k = 0; k < n; k++) r1[k] -= e[k] * yr;
                    const double b = (st[r] == ST_UPPER) ? q->u[r] : q->l[r];
                    /* (round 6) a row at the rounding floor of its computed E_r x cannot be held more exactly: it does not count as a residual */
                    const int above = fabs(b - Ex[r]) > 16.0 * ORC_EPS * (fabs(b) + q->rn[r] * xn);
                    if (above && fabs(b - Ex[r]) > res_eq) res_eq = fabs(b - Ex[r]);
                    if (fabs(b) > bmax) bmax = fabs(b);
                    if (q->row_slot[r] >= 0 && above) nloose++;
                }
                for (int i = 0; i < n; i++) {
                    du[i] = r1[i];                         /* residual of the QP as given: the next hot start needs it without the proximal term */
                    r1[i] -= spv * (x[i] - q->xref[i]);
                    double a = fabs(r1[i]); if (a > res_stat) res_stat = a;
                }
                /* (round 6) THE RESIDUAL'S OWN ROUNDING FLOOR.  r1 is a sum of three vectors, g, Qx and E'y; it cannot be evaluated -- let alone
                 * reduced by a correction -- below a few dozen roundings of the largest of them.  On a QP whose solution lies far out along a flat
                 * direction of Q (fuzz seed 8 id 370: |g| = 2, |Qx| = |E'y| = 1e3, |x| = 4e5) the test res_stat <= resTol (1 + |g|) = 3e-12 asks
                 * for 3e-15 relative to the terms: the refinement stagnated at 7e-12 ... 1.5e-11 with the RIGHT working set, one side of a
                 * comparison passed by luck after sixty trials, the other never.  The tolerance is therefore at least 64 eps (|g_i| + |Qx|_i + |E'y|_i),
                 * largest over i.  Well-scaled QPs never see it (64 eps = 1.4e-14 against 1e-12). */
                for (int i = 0; i < n; i++) { const double t3 = fabs(g[i]) + fabs(q->w_n4[i]) + fabs(-g[i] - q->w_n4[i] - du[i]); if (t3 > sc) sc = t3; }
                const double rtolS = fmax(o->resTol * gs, 64.0 * ORC_EPS * sc);
                if (g_trace_qp) fprintf(stderr, "  orc trial %d stage 2: res_stat %.3e (tol %.3e) res_eq %.3e (tol %.3e) loose %d scale %.2e |x| %.2e\n", trial, res_stat, rtolS, res_eq, o->resTol * (1.0 + bmax), nloose, sc, xn);
                if (res_stat <= rtolS && res_eq <= o->resTol * (1.0 + bmax) && nloose > 0 && nrefine < 2 && trial + 1 < maxTrials) {
                    nrefine++;      /* solved to the residual tolerance, but the active rows can be held more exactly: one more correction */
                } else
                if (res_stat <= rtolS && res_eq <= o->resTol * (1.0 + bmax)) {
                    /* the proximal QP is solved.  Is it the QP as given, i.e. is sigma_p |x - xref| below the tolerance too? */
                    double res_orig = 0;
                    for (int i = 0; i < n; i++) { double a = fabs(du[i]); if (a > res_orig) res_orig = a; }
                    if (res_orig <= rtolS) {
                        memcpy(q->r1_last, du, sizeof(double) * n);
                        memcpy(q->ex_last, Ex, sizeof(double) * mE);
                        memcpy(q->g_last, g, sizeof(double) * n);
                        return 1;
                    }
                    /* no (PSD Hessians, sigma_p = 1e-8 max|Q_ii|, far from xref): next step of the proximal-point iteration, anchored here */
                    memcpy(q->xref, x, sizeof(double) * n);
                    memcpy(r1, du, sizeof(double) * n);
                }
                have_true = 1;
            }
        }
        if (changed) fact_valid = 0;
        if (!fact_valid) {
            /* bring the inverse factor to the working set st[]: rows that left are rotated out, rows that entered (and rows
             * flagged dependent earlier, which may have become independent) are appended in ascending row order */
            int touched = 0, ndel = 0, nadd = 0;
            for (int sl = 0; sl < q->ns; sl++) if (q->slot_row[sl] >= 0 && st[q->slot_row[sl]] == ST_INACT) ndel++;
            for (int r = 0; r < mE; r++) if (st[r] != ST_INACT && q->row_slot[r] < 0) nadd++;
            /* more active rows than variables while the set still changes by more than max(n/2, 32) rows per trial: the primal-dual update has
             * overshot (a cold start far from the solution, where every violated row enters at once) and more trials only thrash
             * with factors at full rank -- give up and let ADMM produce a working set */
            if (trial >= 2 && q->nT - ndel + nadd > n && ndel + nadd > (n / 2 > 32 ? n / 2 : 32)) {
                for (int r = 0; r < mE; r++) ylv[r] = 0.0;
                return 0;
            }
            /* from scratch when the factor is empty, when most of it would change, or when promotions dictate the order (the
             * device builds the factor in one piece then: blocked Cholesky and blocked triangular inverse, ti_bulk) */
            const int bulk = (robust && q->prio_ctr > 0) || (q->nT == 0 && nadd > 0) || (ndel > 0 && ndel >= (q->nT / 2 > 8 ? q->nT / 2 : 8))
                             || nadd >= 16 || (3 * ndel + 7 * nadd >= 96);      /* the last: the device's cost model of updates against a rebuild */
            int full = 0;
            if (bulk) {
                ti_reset(q); touched = 1;
                /* promotions: latest first (ascending row index within one), so that a row OTHER than the promoted one is
                 * found dependent; then the rest in ascending order (the loop below) */
                for (int stamp = robust ? q->prio_ctr : 0; stamp >= 1 && !full; stamp--)
                    for (int r = 0; r < mE && !full; r++)
                        if (st[r] != ST_INACT && q->prio[r] == stamp) {
                            const int rc = ti_append(q, r, o->depTau);
                            if (rc < 0) { full = 1; break; }
                            q->dep[r] = (rc == 0);
                        }
            } else {
                for (int sl = 0; sl < q->ns; sl++)
                    if (q->slot_row[sl] >= 0 && st[q->slot_row[sl]] == ST_INACT) { ti_delete(q, sl); touched = 1; }
            }
            for (int r = 0; r < mE && !full; r++) {
                if (st[r] == ST_INACT) { q->dep[r] = 0; continue; }
                if (q->row_slot[r] >= 0) continue;
                const int rc = ti_append(q, r, o->depTau);
                if (rc < 0) { full = 1; break; }
                q->dep[r] = (rc == 0);
                touched = 1;
            }
            if (full) { for (int r = 0; r < mE; r++) ylv[r] = 0.0; return 0; }
            if (touched) q->c_fact++;
            fact_valid = 1;
        }
        const int nsl = q->ns;
        double* tv = q->w_a2;
        if (g_trace_qp) { int nd = 0; for (int r = 0; r < mE; r++) nd += (st[r] != ST_INACT && q->dep[r]); fprintf(stderr, "  orc trial %d damp %d: left %d changed %d true %d na %d ns %d dep %d\n", trial, damp, nlv, changed, have_true, q->nT, nsl, nd); }
        if (have_true) {
            /* full correction:  c = L1^-1 r1 ;  S dy = T c - r2 ;  dx = L1^-T (c - T' dy)   (T = rows of Et in the slots of the factor) */
            memcpy(c, r1, sizeof(double) * n);
            trsv_lower(q->L1, n, c);
            q->c_trsv++;
            for (int sl = 0; sl < nsl; sl++) {
                const int r = q->slot_row[sl];
                if (r < 0) { tv[sl] = 0.0; continue; }
                const double bb = (st[r] == ST_UPPER) ? q->u[r] : q->l[r];
                const double* ta = q->Et + (size_t)r * n;
                double sdot = 0;
                for (int k = 0; k < n; k++) sdot += ta[k] * c[k];
                tv[sl] = sdot - (bb - Ex[r]);
                q->rows_corr++;
            }
        } else {
            /* predicted correction: c = sum over the rows that left of y_r Et_r (= L1^-1 of their share of the residual) */
            for (int i = 0; i < n; i++) c[i] = 0.0;
            if (nlv > 0)
                for (int r = 0; r < mE; r++) {
                    const double yr = ylv[r];
                    if (yr == 0.0) continue;
                    const double* tr = q->Et + (size_t)r * n;
                    for (int k = 0; k < n; k++) c[k] += yr * tr[k];
                    q->rows_corr++;
                }
            for (int sl = 0; sl < nsl; sl++) {
                const int r = q->slot_row[sl];
                if (r < 0) { tv[sl] = 0.0; continue; }
                const double bb = (st[r] == ST_UPPER) ? q->u[r] : q->l[r];
                double sdot = 0;
                if (nlv > 0) { const double* ta = q->Et + (size_t)r * n; for (int k = 0; k < n; k++) sdot += ta[k] * c[k]; }   /* device: sum_r y_r M[r][row] */
                tv[sl] = sdot - (bb - Ex[r]);
            }
            q->c_pred++;
        }
        for (int r = 0; r < mE; r++) ylv[r] = 0.0;
        ti_apply(q, tv, dy);
        memcpy(du, c, sizeof(double) * n);
        for (int sl = 0; sl < nsl; sl++) {
            const int r = q->slot_row[sl];
            if (r < 0) continue;
            q->rows_corr++;
            if (dy[sl] == 0.0) continue;
            const double* ta = q->Et + (size_t)r * n;
            const double v = dy[sl];
            for (int k = 0; k < n; k++) du[k] -= ta[k] * v;
        }
        trsv_lower_t(q->L1, n, du);
        q->c_trsv++;
        for (int i = 0; i < n; i++) x[i] += du[i];
        for (int sl = 0; sl < nsl; sl++) {
            const int r = q->slot_row[sl];
            if (r < 0) continue;
            yfull[r] += dy[sl];
            Ex[r] = (st[r] == ST_UPPER) ? q->u[r] : q->l[r];      /* the rows of the factor now sit on their bounds (up to rounding) */
        }
        q->c_corr++;
    }
    for (int r = 0; r < mE; r++) ylv[r] = 0.0;
    return 0;
}

int orc_qp_solve(orc_qp_t* q, int initialSolve, int* iterations, int* exit_flag, const double* g,
                 const double* lbA, const double* ubA, const double* x0, const double* y0,
                 const double* lb, const double* ub)
{
    const int n = q->nV, nC = q->nC;
    const orc_options_t* o = &q->opt;
    int trials0 = q->c_trials, admm0 = q->c_admm;
    *iterations = 0;
    *exit_flag = 0;
    if (initialSolve || !q->is_setup || !qp_bounds_compatible(q, lbA, ubA, lb, ub)) {
        double *xkeep = NULL, *ykeep = NULL; /* (a pattern change on a hot start restarts cold) */
        (void)xkeep; (void)ykeep;
        int rc = qp_setup(q, lbA, ubA, lb, ub);
        if (rc != 0) { *exit_flag = rc; return ORC_SUBPROBLEM_SOLVER_ERROR; }
        initialSolve = 1;
    } else {
        qp_update_bounds(q, lbA, ubA, lb, ub);
    }
    const int mE = q->mE;
    for (int r = 0; r < mE; r++)
        if (q->l[r] > q->u[r]) { *exit_flag = 2; return ORC_SUBPROBLEM_SOLVER_ERROR; } /* infeasible bounds */
    if (q->prio_ctr) { memset(q->prio, 0, sizeof(int) * mE); q->prio_ctr = 0; }   /* promotions last for one solve */

    /* starting point: initial solve uses (x0, y0) like qp.init (SubsolverQPOASES.cpp:152); a hot start
     * continues from the previous solution and working set like qp.hotstart (:158) */
    if (initialSolve) {
        for (int i = 0; i < n; i++) q->x[i] = x0 ? x0[i] : 0.0;
        for (int r = 0; r < mE; r++) {
            double yr = 0.0;
            if (y0) yr = (r < nC) ? -y0[n + r] : -y0[q->boxidx[r - nC]];
            q->y[r] = yr;
        }
    }
    memcpy(q->xa, q->x, sizeof(double) * n);
    memcpy(q->xref, q->x, sizeof(double) * n);      /* anchor of the proximal term (qp_polish) */
    memcpy(q->ya, q->y, sizeof(double) * mE);
    int n_admm = initialSolve ? o->admmFirst : o->admmHot;
    int use_stored_set = (!initialSolve && q->have_solution && n_admm == 0);
    /* Rows flagged dependent keep their multiplier while a solve runs (their equations are not in the factor).  ACROSS the solves of a
     * homotopy that let the multipliers of two parallel rows -- duplicated or redundant equalities -- drift apart without bound (1e11 against
     * -1e11 after eight penalty updates: their sum is what the QP determines), until the cancellation error of A'y alone exceeded the
     * stationarity tolerance and the homotopy ran into maxIterations at a point that IS stationary (fuzz seed 22 id 283).  A hot start
     * therefore hands a flagged row's multiplier back: it starts at zero, the stored residual no longer belongs to the stored point, and the
     * polish takes its cold entry (the true residual, every row), on the stored working set.  (round 5; device: qp_solve) */
    int reuse_stored = use_stored_set;
    if (use_stored_set && q->robust)
        for (int r = 0; r < q->mE; r++)
            if (q->dep[r] && q->st[r] != ST_INACT && q->y[r] != 0.0) { q->y[r] = 0.0; reuse_stored = 0; }
    if (!reuse_stored) memcpy(q->ya, q->y, sizeof(double) * q->mE);
    int admm_ready = 0; /* za = clip(E xa) is only needed once ADMM runs */
    double* xt = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* yt = (double*)malloc(sizeof(double) * (mE ? mE : 1));
    int* stt = (int*)malloc(sizeof(int) * (mE ? mE : 1));
    int solved = 0, certificate = 0;
    for (int round = 0; round < o->maxRounds && !solved; round++) {
        if (!admm_ready && (n_admm > 0 || !(round == 0 && use_stored_set))) {
            for (int r = 0; r < mE; r++) {
                const double* e = q->E + (size_t)r * n;
                double s = 0;
                for (int k = 0; k < n; k++) s += e[k] * q->xa[k];
                q->za[r] = clipd(s, q->l[r], q->u[r]);
                if (q->rhov[r] == 0.0) q->ya[r] = 0.0;
            }
            admm_ready = 1;
        }
        if (n_admm > 0) {
            if (!q->k_ready && qp_build_K(q) != 0) { free(xt); free(yt); free(stt); *exit_flag = 3; return ORC_SUBPROBLEM_SOLVER_ERROR; }   /* first ADMM iteration: LK is built now */
            qp_admm(q, g, n_admm);
        }
        if (round == 0 && use_stored_set) {
            memcpy(stt, q->st, sizeof(int) * mE);
            for (int r = 0; r < mE; r++) if (q->l[r] == q->u[r]) stt[r] = ST_EQ;
        } else {
            qp_guess_from_admm(q, stt);
        }
        memcpy(xt, q->xa, sizeof(double) * n);
        for (int r = 0; r < mE; r++) yt[r] = (stt[r] != ST_INACT) ? q->ya[r] : 0.0;
        if (g_trace_qp) fprintf(stderr, " orc round %d: admm %d stored %d reuse %d trials so far %d\n", round, n_admm, use_stored_set, round == 0 && reuse_stored, q->c_trials - trials0);
        if (qp_polish(q, g, xt, yt, stt, round == 0 && reuse_stored, round >= g_damp_round)) { solved = 1; break; }
        if (round >= 1 && n_admm > 0) qp_adapt_rho(q, g);
        if (round >= 2) {    /* at least 20 ADMM iterations behind us: is the QP infeasible or unbounded? */
            certificate = qp_certificate(q, g);
            if (certificate) break;
        }
        n_admm = 2 * n_admm;
        if (n_admm < 10) n_admm = 10;
        if (n_admm > 400) n_admm = 400;
    }
    *iterations = (q->c_trials - trials0) + (q->c_admm - admm0);
    if (!solved) { free(xt); free(yt); free(stt); *exit_flag = certificate ? certificate : 1; return ORC_SUBPROBLEM_SOLVER_ERROR; }
    memcpy(q->x, xt, sizeof(double) * n);
    memcpy(q->y, yt, sizeof(double) * mE);
    memcpy(q->st, stt, sizeof(int) * mE);
    q->have_solution = 1;
    free(xt); free(yt); free(stt);
    /* qpOASES layout/sign */
    memcpy(q->xsol, q->x, sizeof(double) * n);
    for (int i = 0; i < n + nC; i++) q->ysol[i] = 0.0;
    for (int r = 0; r < nC; r++) q->ysol[n + r] = -q->y[r];
    for (int k = 0; k < q->nfin; k++) q->ysol[q->boxidx[k]] = -q->y[nC + k];
    return ORC_SUCCESSFUL_RETURN;
}

void orc_qp_get_solution(orc_qp_t* q, double* x, double* y)
{ /* SubsolverQPOASES.cpp:172-181 */
    memcpy(x, q->xsol, sizeof(double) * q->nV);
    memcpy(y, q->ysol, sizeof(double) * ((size_t)q->nV + q->nC));
}

/* ------------------------------------------------------------------------------------------------
 * LCQP: loadLCQP (dense) + runSolver.   src/LCQProblem.cpp
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int nV, nC, nComp, nDuals, boxDualOffset;
    double *Q, *g, *L, *R, *A, *lbA, *ubA, *lb, *ub, *C, *Qk;
    double *lbL, *lbR; /* NULL when not given */
    double *g_tilde, *g_phi;
    double phi_const;
    double *xk, *yk, *yk_A, *gk, *xnew, *pk, *statk, *constr_statk, *lk_tmp;
    int have_yk;
    double alphak, rho;
    int outerIter, innerIter, totalIter, algoStat;
    double* hist; int histLen;
    const orc_options_t* opt;
    orc_stats_t* stats;
    orc_qp_t* qp;
    int qpIterk, qpExit;
    uint64_t perturbCounter;
} lcqp_t;

static double lcqp_getPhi(lcqp_t* p)
{ /* :1172-1185 */
    double phi_lin = 0;
    if (p->g_phi) phi_lin += orc_util_dot(p->g_phi, p->xk, p->nV);
    return p->phi_const + phi_lin + orc_util_quadform(p->C, p->xk, p->nV) / 2.0;
}

static void lcqp_updateLinearization(lcqp_t* p)
{ /* :1105-1112  gk = rho*C*xk + g_tilde */
    orc_util_affine(p->rho, p->C, p->xk, p->g_tilde, p->gk, p->nV, p->nV);
}

static int lcqp_solveQPSubproblem(lcqp_t* p, int initialSolve)
{ /* :1115-1148 */
    int ret = orc_qp_solve(p->qp, initialSolve, &p->qpIterk, &p->qpExit, p->gk, p->lbA, p->ubA, p->xk,
                           p->have_yk ? p->yk : NULL, p->lb, p->ub);
    p->stats->subproblemIter += p->qpIterk;
    p->stats->qpSolverExitFlag = p->qpExit;
    p->stats->qpSolves++;
    p->have_yk = 1; /* :1129-1131 allocates yk when it was NULL */
    if (ret != ORC_SUCCESSFUL_RETURN) return ret;
    orc_qp_get_solution(p->qp, p->xnew, p->yk);
    for (int i = 0; i < p->nC + 2 * p->nComp; i++) p->yk_A[i] = p->yk[p->boxDualOffset + i];
    orc_util_weighted_vecadd(1, p->xnew, -1, p->xk, p->pk, p->nV);
    return ORC_SUCCESSFUL_RETURN;
}

static void lcqp_updateStationarity(lcqp_t* p)
{ /* :1246-1272  statk = Qk*xk + g_tilde - A'*yk_A - yk[0:nV] */
    const int nV = p->nV, m = p->nC + 2 * p->nComp;
    orc_util_affine(1, p->Qk, p->xk, p->g_tilde, p->statk, nV, nV);
    orc_util_matmul_t(p->A, p->yk_A, p->constr_statk, m, nV, 1);
    orc_util_weighted_vecadd(1, p->statk, -1, p->constr_statk, p->statk, nV);
    /* box term: lb/ub are always allocated on the dense arm (:896-904), so it is always applied */
    orc_util_weighted_vecadd(1, p->statk, -1, p->yk, p->statk, nV);
}

static void lcqp_updatePenalty(lcqp_t* p)
{ /* :1199-1214 */
    if (p->opt->nDynamicPenalty > 0) p->histLen = 0;
    p->rho *= p->opt->penaltyUpdateFactor;
    p->stats->rhoOpt = p->rho;
    orc_util_weighted_matadd(1, p->Q, p->rho, p->C, p->Qk, p->nV, p->nV); /* updateQk :1316-1326 */
    if (p->g_phi) orc_util_weighted_vecadd(1.0, p->g, p->rho, p->g_phi, p->g_tilde, p->nV);
}

static int lcqp_leyfferCheckPositive(lcqp_t* p)
{ /* :1275-1313 */
    int n = p->opt->nDynamicPenalty;
    if (n <= 0) return 0;
    double complCur = lcqp_getPhi(p);
    if (p->histLen < n) { p->hist[p->histLen++] = complCur; return 0; }
    if (lcqp_getPhi(p) < p->opt->complementarityTolerance) { /* complementarityCheck :1156-1158 */
        memmove(p->hist, p->hist + 1, sizeof(double) * (n - 1));
        p->hist[n - 1] = complCur;
        return 0;
    }
    int retFlag = 1;
    for (int i = 0; i < n; i++)
        if (complCur < p->opt->etaDynamicPenalty * p->hist[i]) { retFlag = 0; break; }
    memmove(p->hist, p->hist + 1, sizeof(double) * (n - 1));
    p->hist[n - 1] = complCur;
    return retFlag;
}

static void lcqp_getOptimalStepLength(lcqp_t* p)
{ /* :1217-1237 */
    double qk = orc_util_quadform(p->Qk, p->pk, p->nV);
    orc_util_affine(1, p->Qk, p->xk, p->g_tilde, p->lk_tmp, p->nV, p->nV);
    double lk = orc_util_dot(p->pk, p->lk_tmp, p->nV);
    p->alphak = 1;
    if (qk > 0 && lk < 0) { double a = -lk / qk; p->alphak = a < 1.0 ? a : 1.0; }
}

static void lcqp_perturbStep(lcqp_t* p)
{ /* :1353-1362; the reference draws rand()%3-1 from a time-seeded generator, here a seeded SplitMix64 */
    for (int i = 0; i < p->nV; i++) {
        int randNum = (int)(lcqp_sm64(p->opt->perturbSeed, p->perturbCounter++) % 3ULL) - 1;
        p->xk[i] += randNum * ORC_EPS;
    }
}

static void lcqp_transformDuals(lcqp_t* p)
{ /* :1381-1409 */
    const int nV = p->nV, nC = p->nC, nComp = p->nComp, off = p->boxDualOffset;
    double* tmp = dalloc(nComp);
    orc_util_matmul(p->R, p->xk, tmp, nComp, nV, 1);
    for (int i = 0; i < nComp; i++) p->yk[off + nC + i] = p->yk[off + nC + i] - p->rho * tmp[i];
    orc_util_matmul(p->L, p->xk, tmp, nComp, nV, 1);
    for (int i = 0; i < nComp; i++) p->yk[off + nC + nComp + i] = p->yk[off + nC + nComp + i] - p->rho * tmp[i];
    free(tmp);
}

static void lcqp_determineStationarityType(lcqp_t* p)
{ /* :1412-1453 with getWeakComplementarities :1456-1482 */
    const int nV = p->nV, nC = p->nC, nComp = p->nComp;
    const double ctol = p->opt->complementarityTolerance;
    double *Lx = dalloc(nComp), *Rx = dalloc(nComp);
    orc_util_matmul(p->L, p->xk, Lx, nComp, nV, 1);
    orc_util_matmul(p->R, p->xk, Rx, nComp, nV, 1);
    int s_stat = 1, m_stat = 1, done = 0;
    for (int i = 0; i < nComp && !done; i++) {
        if (!(Lx[i] <= ctol && Rx[i] <= ctol)) continue;
        double a = p->yk_A[nC + i], b = p->yk_A[nC + nComp + i];
        double dualProd = a * b, dualMin = a < b ? a : b;
        if (dualMin < 0) s_stat = 0;
        if (fabs(dualProd) >= ctol && dualMin <= 0) {
            if (dualProd <= ctol) { p->algoStat = ORC_W_STATIONARY; done = 1; break; }
            m_stat = 0;
        }
    }
    free(Lx); free(Rx);
    if (done) return;
    if (s_stat) { p->algoStat = ORC_S_STATIONARY; return; }
    if (m_stat) { p->algoStat = ORC_M_STATIONARY; return; }
    p->algoStat = ORC_C_STATIONARY;
}

static int g_lcqp_robust = 1;   /* every kernel carries the dependent-row rules since round 2 (k_lcqp_run, k_qp_solve) */
void orc_lcqp_set_robust(int on) { g_lcqp_robust = on; }

int orc_lcqp_solve(int nV, int nC, int nComp,
                   const double* Q, const double* g, const double* L, const double* R,
                   const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                   const double* A, const double* lbA, const double* ubA,
                   const double* lb, const double* ub, const double* x0, const double* y0,
                   const orc_options_t* opt, double* xOpt, double* yOpt, orc_stats_t* stats,
                   int traceCap, double* traceScalars, double* traceX, int* traceLen)
{
    orc_options_t dflt;
    orc_stats_t lstats;
    if (!opt) { orc_options_default(&dflt); opt = &dflt; }
    if (!stats) stats = &lstats;
    memset(stats, 0, sizeof(*stats));
    if (traceLen) *traceLen = 0;
    if (nV <= 0 || nComp <= 0) return (stats->returnValue = ORC_LCQPOBJECT_NOT_SETUP);   /* :98-99 */
    if (!g) return (stats->returnValue = ORC_INVALID_OBJECTIVE_LINEAR_TERM);               /* .ipp:44-45 */
    if (!A && nC > 0) return (stats->returnValue = ORC_INVALID_CONSTRAINT_MATRIX);         /* :569-570 */
    if (!L || !R) return (stats->returnValue = ORC_INVALID_COMPLEMENTARITY_MATRIX);        /* :611-612 */
    if (nC < 0) nC = 0;

    lcqp_t P; memset(&P, 0, sizeof(P));
    lcqp_t* p = &P;
    const int m = nC + 2 * nComp;
    p->nV = nV; p->nC = nC; p->nComp = nComp; p->opt = opt; p->stats = stats;
    p->Q = dalloc((size_t)nV * nV); memcpy(p->Q, Q, sizeof(double) * nV * nV);   /* setQ .ipp:27-36 */
    p->g = dalloc(nV); memcpy(p->g, g, sizeof(double) * nV);
    /* setConstraints :563-626: stack [A; L; R], default bounds */
    p->A = dalloc((size_t)m * nV);
    for (int i = 0; i < nC * nV; i++) p->A[i] = A[i];
    for (int i = 0; i < nComp * nV; i++) p->A[i + nC * nV] = L[i];
    for (int i = 0; i < nComp * nV; i++) p->A[i + nC * nV + nComp * nV] = R[i];
    p->lbA = dalloc(m); p->ubA = dalloc(m);
    for (int i = 0; i < nC; i++) { p->lbA[i] = lbA ? lbA[i] : -INFINITY; p->ubA[i] = ubA ? ubA[i] : INFINITY; }
    p->L = dalloc((size_t)nComp * nV); p->R = dalloc((size_t)nComp * nV);
    memcpy(p->L, L, sizeof(double) * nComp * nV); memcpy(p->R, R, sizeof(double) * nComp * nV);
    p->C = dalloc((size_t)nV * nV);
    orc_util_symm_product(p->L, p->R, p->C, nComp, nV);
    int rc = ORC_SUCCESSFUL_RETURN;
    /* setComplementarityBounds :726-785 */
    if (lbL) { p->lbL = dalloc(nComp); }
    if (lbR) { p->lbR = dalloc(nComp); }
    for (int i = 0; i < nComp && rc == 0; i++) {
        if (lbL) { if (lbL[i] <= -INFINITY) { rc = ORC_INVALID_LOWER_COMPLEMENTARITY_BOUND; break; } p->lbL[i] = lbL[i]; p->lbA[nC + i] = lbL[i]; }
        else p->lbA[nC + i] = 0;
        p->ubA[nC + i] = ubL ? ubL[i] : INFINITY;
    }
    for (int i = 0; i < nComp && rc == 0; i++) {
        if (lbR) { if (lbR[i] <= -INFINITY) { rc = ORC_INVALID_LOWER_COMPLEMENTARITY_BOUND; break; } p->lbR[i] = lbR[i]; p->lbA[nC + nComp + i] = lbR[i]; }
        else p->lbA[nC + nComp + i] = 0;
        p->ubA[nC + nComp + i] = ubR ? ubR[i] : INFINITY;
    }
    /* setInitialGuess .ipp:133-158 */
    p->nDuals = nV + m; p->boxDualOffset = nV;                                   /* :889-890 */
    p->xk = dalloc(nV); if (x0) memcpy(p->xk, x0, sizeof(double) * nV);
    p->yk = dalloc(p->nDuals);
    if (y0) { memcpy(p->yk, y0, sizeof(double) * p->nDuals); p->have_yk = 1; }
    /* initializeSolver :885-1034 (dense arm) */
    p->lb = dalloc(nV); p->ub = dalloc(nV);
    for (int i = 0; i < nV; i++) { p->lb[i] = lb ? lb[i] : -INFINITY; p->ub[i] = ub ? ub[i] : INFINITY; }  /* setLB/setUB .ipp:54-112 */
    p->Qk = dalloc((size_t)nV * nV); p->gk = dalloc(nV); p->xnew = dalloc(nV); p->yk_A = dalloc(m);
    p->pk = dalloc(nV); p->statk = dalloc(nV); p->constr_statk = dalloc(nV); p->lk_tmp = dalloc(nV);
    p->g_tilde = dalloc(nV); memcpy(p->g_tilde, p->g, sizeof(double) * nV);     /* :966-967 */
    p->hist = dalloc(opt->nDynamicPenalty > 0 ? opt->nDynamicPenalty : 1);
    if (rc == 0) {
        /* phi expressions :969-996.  (The reference dereferences both lbL and lbR when either is given;
         * a missing one is treated as zeros here.) */
        if (p->lbL || p->lbR) {
            double* zero = dalloc(nComp);
            const double* a = p->lbL ? p->lbL : zero; const double* b = p->lbR ? p->lbR : zero;
            p->phi_const = orc_util_dot(a, b, nComp);
            p->g_phi = dalloc(nV);
            if (p->lbL) orc_util_add_matmul_t(p->R, p->lbL, p->g_phi, nComp, nV, 1);
            if (p->lbR) orc_util_add_matmul_t(p->L, p->lbR, p->g_phi, nComp, nV, 1);
            for (int i = 0; i < nV; i++) p->g_phi[i] = -p->g_phi[i];
            free(zero);
        }
        p->alphak = 1; p->rho = opt->initialPenaltyParameter;                  /* :999-1004 */
        p->algoStat = ORC_PROBLEM_NOT_SOLVED;
        p->qp = orc_qp_create(nV, m, p->Q, p->A, opt);                          /* :906-907 */
        /* the dependent-row rules every device kernel carries since round 2 (k_lcqp_run, k_qp_solve); orc_lcqp_set_robust(0) selects the
         * plain polish of round 1 */
        p->qp->robust = g_lcqp_robust != 0;

        /* runSolver :444-560 */
        if (opt->solveZeroPenaltyFirst) memcpy(p->gk, p->g, sizeof(double) * nV);
        else lcqp_updateLinearization(p);
        rc = lcqp_solveQPSubproblem(p, 1);
        if (rc == 0) {
            orc_util_weighted_matadd(1, p->Q, p->rho, p->C, p->Qk, nV, nV);     /* setQk :880 */
            stats->rhoOpt = p->rho;
            for (;;) {
                orc_util_weighted_vecadd(1, p->xk, p->alphak, p->pk, p->xk, nV); /* updateStep :1240-1243 */
                lcqp_updateStationarity(p);
                if (traceCap > 0 && traceLen && *traceLen < traceCap) {
                    int t = *traceLen;
                    if (traceScalars) {
                        traceScalars[8 * t + 0] = orc_util_maxabs(p->statk, nV);
                        traceScalars[8 * t + 1] = lcqp_getPhi(p);
                        traceScalars[8 * t + 2] = p->rho;
                        traceScalars[8 * t + 3] = p->alphak;
                        /* getObj :1161-1169, getMerit :1188-1196 (Qk = Q + rho C), step size and QP iterations of
                         * updateTrackingVectors (src/OutputStatistics.cpp:131-164) */
                        traceScalars[8 * t + 4] = orc_util_dot(p->g, p->xk, nV) + 0.5 * orc_util_quadform(p->Q, p->xk, nV);
                        traceScalars[8 * t + 5] = traceScalars[8 * t + 4] + 0.5 * p->rho * orc_util_quadform(p->C, p->xk, nV);
                        traceScalars[8 * t + 6] = orc_util_maxabs(p->pk, nV);
                        traceScalars[8 * t + 7] = (double)p->qpIterk;
                    }
                    if (traceX) memcpy(traceX + (size_t)t * nV, p->xk, sizeof(double) * nV);
                    *traceLen = t + 1;
                }
                p->totalIter++; stats->iterTotal++;                             /* :1335-1338 */
                p->innerIter++;
                if (lcqp_leyfferCheckPositive(p)) {
                    lcqp_updatePenalty(p);
                    p->outerIter++; stats->iterOuter++; p->innerIter = 0;
                }
                lcqp_updateLinearization(p);
                if (orc_util_maxabs(p->statk, nV) < opt->stationarityTolerance) {      /* :1151-1153 */
                    if (lcqp_getPhi(p) < opt->complementarityTolerance) {               /* :1156-1158 */
                        lcqp_transformDuals(p);
                        lcqp_determineStationarityType(p);
                        stats->status = p->algoStat;
                        rc = ORC_SUCCESSFUL_RETURN;
                        break;
                    } else {
                        lcqp_updatePenalty(p);
                        p->outerIter++; stats->iterOuter++; p->innerIter = 0;
                    }
                }
                if (p->totalIter > opt->maxIterations) { rc = ORC_MAX_ITERATIONS_REACHED; break; }
                if (p->rho > opt->maxPenaltyParameter) { rc = ORC_MAX_PENALTY_REACHED; break; }
                lcqp_updateLinearization(p);
                rc = lcqp_solveQPSubproblem(p, 0);
                if (rc != 0) break;
                if (opt->perturbStep) lcqp_perturbStep(p);
                lcqp_getOptimalStepLength(p);
            }
        }
    }
    if (xOpt) memcpy(xOpt, p->xk, sizeof(double) * nV);                           /* :1485-1493 */
    if (yOpt) memcpy(yOpt, p->yk, sizeof(double) * p->nDuals);                    /* :1496-1504 */
    stats->status = p->algoStat;
    stats->returnValue = rc;
    if (p->qp) {
        orc_qp_get_counters(p->qp, &stats->admmIter, &stats->trials, &stats->factorizations, &stats->corrections);
        stats->reserved = orc_qp_get_sweeps(p->qp);
        orc_qp_destroy(p->qp);
    }
    free(p->Q); free(p->g); free(p->A); free(p->lbA); free(p->ubA); free(p->L); free(p->R); free(p->C);
    free(p->lbL); free(p->lbR); free(p->xk); free(p->yk); free(p->lb); free(p->ub); free(p->Qk); free(p->gk);
    free(p->xnew); free(p->yk_A); free(p->pk); free(p->statk); free(p->constr_statk); free(p->lk_tmp);
    free(p->g_tilde); free(p->g_phi); free(p->hist);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic instances (include/lcqp_synth.h) and the threaded CPU-baseline driver.
 * ---------------------------------------------------------------------------------------------- */
void orc_synth_generate(uint64_t seed0, uint64_t instance, int n, int nC, int nComp,
                        double* Q, double* g, double* L, double* R, double* A, double* lbA, double* ubA)
{
    const uint64_t st = lcqp_synth_state(seed0, instance);
    double* M = dalloc((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) M[(size_t)i * n + j] = lcqp_synth_M(st, n, i, j);
    /* Q = M'M/n + I, summed over k ascending (the device generator uses the same order) */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Q[(size_t)i * n + j] = 0.0;
    for (int k = 0; k < n; k++) {
        const double* mk = M + (size_t)k * n;
        for (int i = 0; i < n; i++) {
            double a = mk[i];
            double* qi = Q + (size_t)i * n;
            for (int j = 0; j < n; j++) qi[j] += a * mk[j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Q[(size_t)i * n + j] = Q[(size_t)i * n + j] / (double)n + (i == j ? 1.0 : 0.0);
    free(M);
    for (int i = 0; i < n; i++) g[i] = lcqp_synth_g(st, n, nC, nComp, i);
    memset(L, 0, sizeof(double) * nComp * n);
    memset(R, 0, sizeof(double) * nComp * n);
    for (int i = 0; i < nComp; i++) { L[(size_t)i * n + i] = 1.0; R[(size_t)i * n + nComp + i] = 1.0; }
    const double sn = sqrt((double)n);
    for (int r = 0; r < nC; r++) {
        double ax = 0;
        for (int c = 0; c < n; c++) {
            double a = lcqp_synth_Araw(st, n, nC, nComp, r, c) / sn;
            A[(size_t)r * n + c] = a;
            ax += a * lcqp_synth_xstar(st, n, nC, nComp, c);
        }
        lbA[r] = ax - lcqp_synth_slo(st, n, nC, nComp, r);
        ubA[r] = ax + lcqp_synth_shi(st, n, nC, nComp, r);
    }
}

typedef struct {
    uint64_t seed0; int first, count, n, nC, nComp; const orc_options_t* opt;
    double *xOut, *yOut; orc_stats_t* statsOut;
    int next; int ok; pthread_mutex_t mu;
} batch_ctx_t;

static void* batch_worker(void* arg)
{
    batch_ctx_t* c = (batch_ctx_t*)arg;
    const int n = c->n, nC = c->nC, nComp = c->nComp, nd = n + nC + 2 * nComp;
    double *Q = dalloc((size_t)n * n), *g = dalloc(n), *L = dalloc((size_t)nComp * n), *R = dalloc((size_t)nComp * n);
    double *A = dalloc((size_t)nC * n), *lbA = dalloc(nC), *ubA = dalloc(nC), *x = dalloc(n), *y = dalloc(nd);
    for (;;) {
        pthread_mutex_lock(&c->mu);
        int k = c->next++;
        pthread_mutex_unlock(&c->mu);
        if (k >= c->count) break;
        orc_stats_t st;
        orc_synth_generate(c->seed0, (uint64_t)(c->first + k), n, nC, nComp, Q, g, L, R, A, lbA, ubA);
        int rc = orc_lcqp_solve(n, nC, nComp, Q, g, L, R, NULL, NULL, NULL, NULL, A, lbA, ubA, NULL, NULL, NULL, NULL,
                                c->opt, x, y, &st, 0, NULL, NULL, NULL);
        if (c->xOut) memcpy(c->xOut + (size_t)k * n, x, sizeof(double) * n);
        if (c->yOut) memcpy(c->yOut + (size_t)k * nd, y, sizeof(double) * nd);
        if (c->statsOut) c->statsOut[k] = st;
        if (rc == 0) { pthread_mutex_lock(&c->mu); c->ok++; pthread_mutex_unlock(&c->mu); }
    }
    free(Q); free(g); free(L); free(R); free(A); free(lbA); free(ubA); free(x); free(y);
    return NULL;
}

int orc_synth_batch_solve(uint64_t seed0, int first, int count, int n, int nC, int nComp, const orc_options_t* opt,
                          int threads, double* xOut, double* yOut, orc_stats_t* statsOut)
{
    batch_ctx_t c;
    memset(&c, 0, sizeof(c));
    c.seed0 = seed0; c.first = first; c.count = count; c.n = n; c.nC = nC; c.nComp = nComp; c.opt = opt;
    c.xOut = xOut; c.yOut = yOut; c.statsOut = statsOut;
    pthread_mutex_init(&c.mu, NULL);
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, batch_worker, &c);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&c.mu);
    return c.ok;
}

/* Steady-state CPU baseline (bench.py's cpu_baseline leg; what the reference's own timer brackets is load + solve,
 * interfaces/matlab/LCQPow.cpp:882,915-916).  `threads` workers, worker t pinned to cpus[t] (cpus may be NULL: no pinning), each with
 * its own `perThread` instances first + t*perThread ... generated BEFORE the clock starts; the allocator is told to keep freed blocks
 * (no mmap / munmap per solve) and every worker runs one untimed warm-up solve, so that the timed solves reuse the worker's buffers
 * instead of faulting fresh pages in.  All workers start the timed part together (barrier); secondsOut = start to the last
 * worker's finish.  xOut / yOut / statsOut (may be NULL) are indexed by instance - first.  Returns the number of timed instances that
 * returned SUCCESSFUL_RETURN. */
#include <malloc.h>
#include <sched.h>
#include <time.h>
typedef struct {
    uint64_t seed0; int first, perThread, n, nC, nComp; const orc_options_t* opt;
    double *xOut, *yOut; orc_stats_t* statsOut;
    pthread_barrier_t* bar; int tid, cpu, ok; double tEnd;
} bench_worker_t;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void* bench_worker(void* arg)
{
    bench_worker_t* w = (bench_worker_t*)arg;
    if (w->cpu >= 0) {
        cpu_set_t set; CPU_ZERO(&set); CPU_SET(w->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);      /* (best effort: a refused mask leaves the worker unpinned) */
    }
    const int n = w->n, nC = w->nC, nComp = w->nComp, nd = n + nC + 2 * nComp, K = w->perThread;
    const size_t szQ = (size_t)n * n, szL = (size_t)nComp * n, szA = (size_t)nC * n;
    double *Q = dalloc(szQ * K), *g = dalloc((size_t)n * K), *L = dalloc(szL * K), *R = dalloc(szL * K);
    double *A = dalloc(szA * K), *lbA = dalloc((size_t)nC * K), *ubA = dalloc((size_t)nC * K), *x = dalloc(n), *y = dalloc(nd);
    for (int k = 0; k < K; k++)
        orc_synth_generate(w->seed0, (uint64_t)(w->first + w->tid * K + k), n, nC, nComp, Q + szQ * k, g + (size_t)n * k, L + szL * k, R + szL * k,
                           A + szA * k, lbA + (size_t)nC * k, ubA + (size_t)nC * k);
    orc_stats_t st;
    (void)orc_lcqp_solve(n, nC, nComp, Q, g, L, R, NULL, NULL, NULL, NULL, A, lbA, ubA, NULL, NULL, NULL, NULL, w->opt, x, y, &st, 0, NULL, NULL, NULL);   /* warm-up */
    pthread_barrier_wait(w->bar);      /* the clock starts here (taken by the caller between two barriers) */
    pthread_barrier_wait(w->bar);
    for (int k = 0; k < K; k++) {
        const int rc = orc_lcqp_solve(n, nC, nComp, Q + szQ * k, g + (size_t)n * k, L + szL * k, R + szL * k, NULL, NULL, NULL, NULL,
                                      A + szA * k, lbA + (size_t)nC * k, ubA + (size_t)nC * k, NULL, NULL, NULL, NULL, w->opt, x, y, &st, 0, NULL, NULL, NULL);
        const size_t id = (size_t)w->tid * K + k;
        if (w->xOut) memcpy(w->xOut + id * n, x, sizeof(double) * n);
        if (w->yOut) memcpy(w->yOut + id * nd, y, sizeof(double) * nd);
        if (w->statsOut) w->statsOut[id] = st;
        w->ok += (rc == 0);
    }
    w->tEnd = now_s();
    free(Q); free(g); free(L); free(R); free(A); free(lbA); free(ubA); free(x); free(y);
    return NULL;
}

int orc_synth_bench(uint64_t seed0, int first, int threads, int perThread, const int* cpus, int n, int nC, int nComp, const orc_options_t* opt,
                    double* xOut, double* yOut, orc_stats_t* statsOut, double* secondsOut)
{
    if (threads < 1 || perThread < 1) return -1;
    /* matrices come from the worker's heap arena and go back to it: no mmap / munmap / page faults per solve (they would serialise the
     * workers on the address-space lock of the process: 128 workers then reach what 20 do).  32 MiB is the largest threshold glibc takes. */
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1);
    bench_worker_t* w = (bench_worker_t*)calloc((size_t)threads, sizeof(bench_worker_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        w[t].seed0 = seed0; w[t].first = first; w[t].perThread = perThread; w[t].n = n; w[t].nC = nC; w[t].nComp = nComp; w[t].opt = opt;
        w[t].xOut = xOut; w[t].yOut = yOut; w[t].statsOut = statsOut; w[t].bar = &bar; w[t].tid = t; w[t].cpu = cpus ? cpus[t] : -1;
        pthread_create(&th[t], NULL, bench_worker, &w[t]);
    }
    pthread_barrier_wait(&bar);      /* every worker has generated its instances and finished its warm-up solve */
    const double t0 = now_s();
    pthread_barrier_wait(&bar);      /* release */
    int ok = 0; double tEnd = t0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); ok += w[t].ok; if (w[t].tEnd > tEnd) tEnd = w[t].tEnd; }
    if (secondsOut) *secondsOut = tEnd - t0;
    pthread_barrier_destroy(&bar);
    free(w); free(th);
    return ok;
}

/* Read bandwidth of the host memory seen by `threads` pinned workers streaming `bytesPerThread` each (private arrays, first touched by their
 * worker), `reps` passes: GB/s.  bench.py prints it beside the cpu_baseline: with every core busy the oracle is bound by this, not by its
 * arithmetic (its working set per instance, ~15 MB, exceeds a core's share of the L3). */
typedef struct { pthread_barrier_t* bar; int cpu, reps; size_t n; double sum, tEnd; } stream_worker_t;
static void* stream_worker(void* arg)
{
    stream_worker_t* w = (stream_worker_t*)arg;
    if (w->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(w->cpu, &set); pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
    double* a = (double*)malloc(sizeof(double) * w->n);
    for (size_t i = 0; i < w->n; i++) a[i] = (double)(i & 7);
    pthread_barrier_wait(w->bar);
    pthread_barrier_wait(w->bar);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int r = 0; r < w->reps; r++)
        for (size_t i = 0; i + 3 < w->n; i += 4) { s0 += a[i]; s1 += a[i + 1]; s2 += a[i + 2]; s3 += a[i + 3]; }
    w->sum = (s0 + s1) + (s2 + s3);
    w->tEnd = now_s();
    free(a);
    return NULL;
}
double orc_host_stream_gbps(int threads, const int* cpus, size_t bytesPerThread, int reps)
{
    if (threads < 1 || reps < 1) return 0.0;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1);
    stream_worker_t* w = (stream_worker_t*)calloc((size_t)threads, sizeof(stream_worker_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) { w[t].bar = &bar; w[t].cpu = cpus ? cpus[t] : -1; w[t].reps = reps; w[t].n = bytesPerThread / sizeof(double); pthread_create(&th[t], NULL, stream_worker, &w[t]); }
    pthread_barrier_wait(&bar);
    const double t0 = now_s();
    pthread_barrier_wait(&bar);
    double tEnd = t0, chk = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (w[t].tEnd > tEnd) tEnd = w[t].tEnd; chk += w[t].sum; }
    const double gbps = (chk >= 0.0 && tEnd > t0) ? (double)threads * (double)(bytesPerThread / sizeof(double)) * sizeof(double) * reps / (tEnd - t0) / 1e9 : 0.0;
    pthread_barrier_destroy(&bar);
    free(w); free(th);
    return gbps;
}

/* ------------------------------------------------------------------------------------------------
 * CSC utilities restated (src/Utilities.cpp:49-59,75-82,96-102,118-173,189-199,228-241,469-650)
 * ---------------------------------------------------------------------------------------------- */
orc_csc_t* orc_csc_create(int m, int n, int nnz, const double* x, const int* i, const int* p)
{ /* copyCSC(int,int,int,...) :487-513 */
    orc_csc_t* M = (orc_csc_t*)malloc(sizeof(orc_csc_t));
    M->m = m; M->n = n; M->nzmax = nnz; M->nz = -1;
    M->p = (int*)malloc(sizeof(int) * (n + 1));
    M->i = (int*)malloc(sizeof(int) * (nnz ? nnz : 1));
    M->x = (double*)malloc(sizeof(double) * (nnz ? nnz : 1));
    memcpy(M->p, p, sizeof(int) * (n + 1));
    if (nnz) { memcpy(M->i, i, sizeof(int) * nnz); memcpy(M->x, x, sizeof(double) * nnz); }
    return M;
}

void orc_csc_free(orc_csc_t* M)
{
    if (!M) return;
    free(M->p); free(M->i); free(M->x); free(M);
}

orc_csc_t* orc_csc_upper(const orc_csc_t* A)
{ /* copyCSC(M, toUpperTriangular = true) :516-560: keep entries on or above the diagonal */
    int cnt = 0;
    for (int j = 0; j < A->n; j++)
        for (int k = A->p[j]; k < A->p[j + 1]; k++) cnt += (A->i[k] <= j);
    orc_csc_t* M = (orc_csc_t*)malloc(sizeof(orc_csc_t));
    M->m = A->m; M->n = A->n; M->nzmax = cnt; M->nz = -1;
    M->p = (int*)malloc(sizeof(int) * (A->n + 1));
    M->i = (int*)malloc(sizeof(int) * (cnt ? cnt : 1));
    M->x = (double*)malloc(sizeof(double) * (cnt ? cnt : 1));
    M->p[0] = 0;
    int w = 0;
    for (int j = 0; j < A->n; j++) {
        for (int k = A->p[j]; k < A->p[j + 1]; k++) {
            if (A->i[k] > j) continue;
            M->i[w] = A->i[k]; M->x[w] = A->x[k]; w++;
        }
        M->p[j + 1] = w;
    }
    return M;
}

double* orc_csc_to_dns(const orc_csc_t* S)
{ /* :593-617 */
    const int m = S->m, n = S->n;
    const size_t mn = (size_t)m * (size_t)n;
    double* full = (double*)calloc(mn > 0 ? mn : 1, sizeof(double));
    for (int j = 0; j < n; j++)
        for (int k = S->p[j]; k < S->p[j + 1]; k++) {
            if (k == S->nzmax) return full;
            if ((long)S->i[k] * n + j >= (long)m * n || S->i[k] < 0) { free(full); return NULL; }
            full[(size_t)S->i[k] * n + j] = S->x[k];
        }
    return full;
}

orc_csc_t* orc_dns_to_csc(const double* full, int m, int n)
{ /* :620-650: column by column, every entry that is > 0 or < 0 */
    int nnz = 0;
    for (size_t e = 0; e < (size_t)m * n; e++) nnz += (full[e] > 0 || full[e] < 0);
    orc_csc_t* M = (orc_csc_t*)malloc(sizeof(orc_csc_t));
    M->m = m; M->n = n; M->nzmax = nnz; M->nz = -1;
    M->p = (int*)malloc(sizeof(int) * (n + 1));
    M->i = (int*)malloc(sizeof(int) * (nnz ? nnz : 1));
    M->x = (double*)malloc(sizeof(double) * (nnz ? nnz : 1));
    M->p[0] = 0;
    int w = 0;
    for (int c = 0; c < n; c++) {
        for (int r = 0; r < m; r++) {
            const double v = full[(size_t)r * n + c];
            if (v > 0 || v < 0) { M->i[w] = r; M->x[w] = v; w++; }
        }
        M->p[c + 1] = w;
    }
    return M;
}

void orc_csc_matmul(const orc_csc_t* A, const double* b, double* c)
{ /* :49-59  c = A b */
    for (int i = 0; i < A->m; i++) c[i] = 0;
    for (int j = 0; j < A->n; j++)
        for (int k = A->p[j]; k < A->p[j + 1]; k++) c[A->i[k]] += A->x[k] * b[j];
}

void orc_csc_matmul_t(const orc_csc_t* A, const double* b, double* c)
{ /* :75-82  c = A' b */
    for (int j = 0; j < A->n; j++) {
        c[j] = 0;
        for (int k = A->p[j]; k < A->p[j + 1]; k++) c[j] += b[A->i[k]] * A->x[k];
    }
}

void orc_csc_add_matmul_t(const orc_csc_t* A, const double* b, double* c)
{ /* :96-102  c += A' b */
    for (int j = 0; j < A->n; j++)
        for (int k = A->p[j]; k < A->p[j + 1]; k++) c[j] += b[A->i[k]] * A->x[k];
}

static int csc_index_of(int val, const int* sorted, int beg, int end)
{ /* getIndexOfIn :727-737 */
    for (int k = beg; k < end; k++) {
        if (sorted[k] == val) return k;
        if (sorted[k] > val) break;
    }
    return -1;
}

orc_csc_t* orc_csc_symm_product(const orc_csc_t* L, const orc_csc_t* R)
{ /* :118-173  C = L'R + R'L, entries with |value| <= ZERO (1e-25) dropped, NULL when empty */
    const int n = L->n;
    int cap = 16, w = 0;
    int* ci = (int*)malloc(sizeof(int) * cap);
    double* cx = (double*)malloc(sizeof(double) * cap);
    int* cp = (int*)malloc(sizeof(int) * (n + 1));
    cp[0] = 0;
    for (int j = 0; j < n; j++) {
        for (int i = 0; i < n; i++) {
            double tmp = 0;
            for (int k = L->p[i]; k < L->p[i + 1]; k++) {
                int s2 = csc_index_of(L->i[k], R->i, R->p[j], R->p[j + 1]);
                if (s2 != -1) tmp += L->x[k] * R->x[s2];
            }
            for (int k = R->p[i]; k < R->p[i + 1]; k++) {
                int s1 = csc_index_of(R->i[k], L->i, L->p[j], L->p[j + 1]);
                if (s1 != -1) tmp += R->x[k] * L->x[s1];
            }
            if (!(fabs(tmp) <= 1.0e-25)) {
                if (w == cap) { cap *= 2; ci = (int*)realloc(ci, sizeof(int) * cap); cx = (double*)realloc(cx, sizeof(double) * cap); }
                ci[w] = i; cx[w] = tmp; w++;
            }
        }
        cp[j + 1] = w;
    }
    if (w == 0) { free(ci); free(cx); free(cp); return NULL; }
    orc_csc_t* M = (orc_csc_t*)malloc(sizeof(orc_csc_t));
    M->m = n; M->n = n; M->nzmax = w; M->nz = -1; M->p = cp; M->i = ci; M->x = cx;
    return M;
}

void orc_csc_affine(double alpha, const orc_csc_t* S, const double* b, const double* c, double* d, int m)
{ /* :189-199  d = alpha * S' b + c  (the reference relies on S being symmetric) */
    for (int j = 0; j < m; j++) {
        double tmp = 0;
        for (int k = S->p[j]; k < S->p[j + 1]; k++) tmp += S->x[k] * b[S->i[k]];
        d[j] = alpha * tmp + c[j];
    }
}

double orc_csc_quadform(const orc_csc_t* S, const double* p, int m)
{ /* :228-241 */
    double ret = 0;
    for (int j = 0; j < m; j++) {
        double tmp = 0;
        for (int k = S->p[j]; k < S->p[j + 1]; k++) tmp += S->x[k] * p[S->i[k]];
        ret += p[j] * tmp;
    }
    return ret;
}
