/*
 * lcqp_oracle_sparse.c -- CPU restatement of the SPARSE arm of the hot path (TEST INFRASTRUCTURE ONLY, like lcqp_oracle.c:
 * used by tests/, bench.py's cpu_baseline leg and nothing else).
 *
 * What it restates:
 *   - LCQProblem::runSolver (src/LCQProblem.cpp:444-560) on the OSQP_SPARSE arm of initializeSolver (:929-960): no box
 *     constraints, nDuals = nC + 2 nComp, boxDualOffset = 0, statk = Qk xk + g_tilde - A'yk_A without a box term (:1246-1272 with
 *     lb = ub = NULL), the sparse stacking [A; L; R] (:629-723), C xk applied as L'(R xk) + R'(L xk) (C = L'R + R'L, :622-623);
 *   - the subsolver behind SubsolverOSQP (src/SubsolverOSQP.cpp:124-200): OSQP itself is an un-vendored submodule (external/osqp
 *     is empty, version unknown) -- PARITY UNPINNED at that boundary.  What stands in its place is OSQP's published algorithm
 *     (Stellato et al., Math. Prog. Comp. 12, 2020): ADMM on the quasi-definite KKT matrix [Q + sigma I, E'; E, -diag(1/rho)]
 *     factorised once (LDL'), and a polish on the guessed active set with the regularised KKT matrix
 *     [Q + delta I, Ea'; Ea, -delta2 I] in iterative-refinement form, run as a primal-dual active-set loop exactly like the dense
 *     restatement (qp_polish in lcqp_oracle.c) so that LCQPow's tolerances (complementarity 2.2e-13) are met.  Duals are returned
 *     with the sign flip of src/SubsolverOSQP.cpp:196-199.
 * The KKT matrices are factorised as band matrices in an ordering handed in by the caller (perm, half bandwidth w); the result
 * does not depend on the ordering beyond rounding.
 */
#include "lcqp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { SP_INACT = 0, SP_LOWER = 1, SP_UPPER = 2, SP_EQ = 3 };

typedef struct {
    int N, w;      /* N: nodes inside the band (the border nodes come after them in the ordering) */
    double* B;     /* N x (w+1): B[i*(w+1) + k] = K[i][i-w+k]; after factorisation the unit lower factor, B[i][w] = D_i */
    /* bordered band (round 3): K = [B U'; U C] with kb border nodes -- nodes whose rows are too dense for any band, e.g. the coupling
     * constraint and the two shared variables of examples/OptimizeOnCircle.cpp.  B is factorised as a band, the border is carried by
     * W = U inv(B) and the Schur complement S = C - U inv(B) U' (kb x kb, LDL' without pivoting: a Schur complement of a quasi-definite
     * matrix is quasi-definite). */
    int kb;
    double *U, *W, *S;   /* kb x N (U, then W), kb x kb */
    double* Uf;          /* N x (w+1): the rows in upper form while band_factor runs (allocated once at setup, beside B) */
    struct gen_s* G;     /* general sparse LDL' (round 6; selected by a negative half bandwidth): patterns that are neither banded nor bordered */
} band_t;

/* General sparse LDL' of the permuted KKT matrix, K = L D L' without pivoting (quasi-definite for delta, delta2 > 0: any symmetric
 * permutation factorises).  The up-looking algorithm of T. A. Davis, "Algorithm 849: a concise sparse Cholesky factorization package",
 * ACM TOMS 31 (2005) -- the one OSQP's QDLDL restates (the reference's OSQP arm factorises its KKT matrix with it: src/SubsolverOSQP.cpp:152,
 * osqp_setup; OSQP itself is an absent submodule) -- elimination tree and column counts once per pattern, numeric factorisation per working set.
 * A: upper triangle of K in the ordering perm, compressed columns (row index <= column), pattern fixed (every row of E in), values
 * re-assembled per factorisation: an inactive row keeps explicit zeros and the diagonal -1. */
typedef struct gen_s {
    int N, nnzA, nnzL;
    int *Ap, *Ai; double* Ax;
    int *posQ, *posE, *posD;        /* A-position of every non-zero of Q (-1: strictly lower in this ordering, its mirror is taken) and of E; of the diagonals */
    int *parent, *Lp, *Li, *Lnz, *flag, *pattern;
    double *Lx, *D, *Y;
} gen_t;

typedef struct {
    int n, m, nC, nComp, N, w, kb;      /* kb: border nodes, the last kb positions of perm */
    const int *Qp, *Qi; const double* Qx;      /* CSR of the symmetric Q */
    const int *Ep, *Ei; const double* Ex;      /* CSR of E = [A; L; R] */
    const int *perm; int* iperm;
    double *l, *u, *rhov;
    double scale, sigma, delta, delta2;        /* regularisation of the polish KKT matrix, safe level (PSD Hessians, dependent active rows) */
    double delta_s, delta2_s; int big_reg;     /* the light level tried first (round 3), and whether this problem has needed the safe one */
    band_t Ka, Kp;
    int* stf; int stf_valid;                   /* working set the polish factor was built for */
    double *x, *y; int* st; int have_solution;
    double *xa, *ya, *za;
    double *r1, *ex, *wN, *r1_last, *ex_last, *g_last;
    int* newst;
    orc_options_t opt;
    int c_admm, c_trials, c_fact, c_corr, c_sweeps;
} sqp_t;

static double* dal(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }
static int* ial(size_t n) { return (int*)calloc(n ? n : 1, sizeof(int)); }
static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- sparse products (CSR) ------------------------------------------------------------------------------------------- */
static void sp_Qx(const sqp_t* q, const double* x, double* out)
{
    for (int i = 0; i < q->n; i++) { double s = 0; for (int k = q->Qp[i]; k < q->Qp[i + 1]; k++) s += q->Qx[k] * x[q->Qi[k]]; out[i] = s; }
}
static void sp_Ex(const sqp_t* q, const double* x, double* out)
{
    for (int r = 0; r < q->m; r++) { double s = 0; for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) s += q->Ex[k] * x[q->Ei[k]]; out[r] = s; }
}
static void sp_ETy_sub(const sqp_t* q, const double* y, double* out)   /* out -= E'y */
{
    for (int r = 0; r < q->m; r++) { const double yr = y[r]; if (yr == 0.0) continue; for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) out[q->Ei[k]] -= q->Ex[k] * yr; }
}

/* ---- band LDL' (quasi-definite matrix, no pivoting) ---------------------------------------------------------------------- */
static void band_factor(band_t* f)
{
    /* Right-looking, every row in UPPER form relative to its diagonal, U[r][k] = K[r][r + k] (by symmetry the entries of column r below the
     * diagonal) -- the arithmetic of the device's sp_factor_reg (round 5) in the same order: at step j the row i = j + a forms its
     * multiplier L[i][j] = U[j][a] * (1 / d_j) and loses L[i][j] * U[j][a + k], k = 0 .. w - a.  (Until round 4: left-looking, L[i][j] =
     * s / d_j with the products (L[i][k] d_k) L[j][k] -- the same sums in another association.)  Result in f->B as before: L below the
     * diagonal (unit diagonal implied), D on it. */
    const int N = f->N, w = f->w, ld = w + 1;
    double* U = f->Uf;      /* workspace of the object: no allocation inside the timed CPU path (ADVICE, round 5) */
    for (int r = 0; r < N; r++)
        for (int k = 0; k <= w; k++) U[(size_t)r * ld + k] = (r + k < N) ? f->B[(size_t)(r + k) * ld + (w - k)] : 0.0;
    for (int j = 0; j < N; j++) {
        const double* uj = U + (size_t)j * ld;
        const double d = uj[0], rinv = 1.0 / d;
        for (int a = 1; a <= w && j + a < N; a++) {
            double* ui = U + (size_t)(j + a) * ld;
            const double la = uj[a] * rinv;
            for (int k = 0; k + a <= w; k++) ui[k] -= la * uj[a + k];
            f->B[(size_t)(j + a) * ld + (w - a)] = la;
        }
        f->B[(size_t)j * ld + w] = d;
    }
}
static void band_solve(const band_t* f, double* b)
{
    const int N = f->N, w = f->w, ld = w + 1;
    for (int i = 0; i < N; i++) {
        const double* ri = f->B + (size_t)i * ld;
        const int j0 = i - w > 0 ? i - w : 0;
        double s = b[i];
        for (int j = j0; j < i; j++) s -= ri[w - (i - j)] * b[j];
        b[i] = s;
    }
    for (int i = 0; i < N; i++) b[i] /= f->B[(size_t)i * ld + w];
    for (int i = N - 1; i >= 0; i--) {
        const double* ri = f->B + (size_t)i * ld;
        const int j0 = i - w > 0 ? i - w : 0;
        const double xi = b[i];
        for (int j = j0; j < i; j++) b[j] -= ri[w - (i - j)] * xi;
    }
}
/* ---- general sparse LDL' (see gen_t) ------------------------------------------------------------------------------------------------------ */
static void gen_free(gen_t* G)
{
    if (!G) return;
    free(G->Ap); free(G->Ai); free(G->Ax); free(G->posQ); free(G->posE); free(G->posD); free(G->parent); free(G->Lp); free(G->Li); free(G->Lnz);
    free(G->flag); free(G->pattern); free(G->Lx); free(G->D); free(G->Y); free(G);
}
static int gen_cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }
/* pattern of the upper triangle of the permuted K, the maps from the non-zeros of Q and E into it, elimination tree, column counts */
static gen_t* gen_setup(int n, int m, const int* Qp, const int* Qi, const int* Ep, const int* Ei, const int* iperm)
{
    const int N = n + m;
    gen_t* G = (gen_t*)calloc(1, sizeof(gen_t));
    if (!G) return NULL;
    G->N = N;
    int* cnt = (int*)calloc((size_t)N + 1, sizeof(int));
    /* column counts: one diagonal entry per node, one entry per Q pair (taken once: from the non-zero whose first index comes later), one per E non-zero */
    for (int p = 0; p < N; p++) cnt[p + 1] = 1;
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) { const int pi = iperm[i], pj = iperm[Qi[k]]; if (pj < pi) cnt[pi + 1]++; }
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) { const int pr = iperm[n + r], pj = iperm[Ei[k]]; cnt[(pr > pj ? pr : pj) + 1]++; }
    G->Ap = (int*)malloc(sizeof(int) * ((size_t)N + 1));
    G->Ap[0] = 0;
    for (int p = 0; p < N; p++) G->Ap[p + 1] = G->Ap[p] + cnt[p + 1];
    G->nnzA = G->Ap[N];
    G->Ai = (int*)malloc(sizeof(int) * (size_t)(G->nnzA ? G->nnzA : 1));
    G->Ax = dal((size_t)G->nnzA);
    int* fill = (int*)calloc((size_t)N, sizeof(int));
    for (int p = 0; p < N; p++) G->Ai[G->Ap[p] + fill[p]++] = p;
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) { const int pi = iperm[i], pj = iperm[Qi[k]]; if (pj < pi) G->Ai[G->Ap[pi] + fill[pi]++] = pj; }
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) {
        const int pr = iperm[n + r], pj = iperm[Ei[k]], hi = pr > pj ? pr : pj, lo = pr > pj ? pj : pr;
        G->Ai[G->Ap[hi] + fill[hi]++] = lo;
    }
    for (int p = 0; p < N; p++) qsort(G->Ai + G->Ap[p], (size_t)(G->Ap[p + 1] - G->Ap[p]), sizeof(int), gen_cmp_int);
    /* maps (binary search in the sorted columns) */
    G->posQ = (int*)malloc(sizeof(int) * (size_t)(Qp[n] ? Qp[n] : 1)); G->posE = (int*)malloc(sizeof(int) * (size_t)(Ep[m] ? Ep[m] : 1)); G->posD = (int*)malloc(sizeof(int) * (size_t)N);
#define GEN_FIND(col, row, out) do { int lo_ = G->Ap[col], hi_ = G->Ap[(col) + 1] - 1; (out) = -1; while (lo_ <= hi_) { const int md_ = (lo_ + hi_) / 2; \
        if (G->Ai[md_] == (row)) { (out) = md_; break; } if (G->Ai[md_] < (row)) lo_ = md_ + 1; else hi_ = md_ - 1; } } while (0)
    for (int p = 0; p < N; p++) GEN_FIND(p, p, G->posD[p]);
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) {
        const int pi = iperm[i], pj = iperm[Qi[k]];
        if (pj < pi) GEN_FIND(pi, pj, G->posQ[k]); else if (pj == pi) G->posQ[k] = G->posD[pi]; else G->posQ[k] = -1;      /* the mirror entry carries the pair */
    }
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) {
        const int pr = iperm[n + r], pj = iperm[Ei[k]], hi = pr > pj ? pr : pj, lo = pr > pj ? pj : pr;
        GEN_FIND(hi, lo, G->posE[k]);
    }
#undef GEN_FIND
    free(cnt); free(fill);
    /* elimination tree and column counts of L */
    G->parent = (int*)malloc(sizeof(int) * (size_t)N); G->Lnz = (int*)calloc((size_t)N, sizeof(int)); G->flag = (int*)malloc(sizeof(int) * (size_t)N);
    G->pattern = (int*)malloc(sizeof(int) * (size_t)N); G->Lp = (int*)malloc(sizeof(int) * ((size_t)N + 1));
    for (int k = 0; k < N; k++) {
        G->parent[k] = -1; G->flag[k] = k;
        for (int p = G->Ap[k]; p < G->Ap[k + 1]; p++) {
            int i = G->Ai[p];
            for (; i < k && G->flag[i] != k; i = G->parent[i]) {
                if (G->parent[i] == -1) G->parent[i] = k;
                G->Lnz[i]++;
                G->flag[i] = k;
            }
        }
    }
    G->Lp[0] = 0;
    for (int k = 0; k < N; k++) G->Lp[k + 1] = G->Lp[k] + G->Lnz[k];
    G->nnzL = G->Lp[N];
    G->Li = (int*)malloc(sizeof(int) * (size_t)(G->nnzL ? G->nnzL : 1)); G->Lx = dal((size_t)G->nnzL); G->D = dal((size_t)N); G->Y = dal((size_t)N);
    if (!G->Li || !G->Lx || !G->D || !G->Y) { gen_free(G); return NULL; }
    return G;
}
static void gen_factor(gen_t* G)
{
    const int N = G->N;
    for (int k = 0; k < N; k++) {
        int top = N;
        G->Y[k] = 0.0; G->flag[k] = k; G->Lnz[k] = 0;
        for (int p = G->Ap[k]; p < G->Ap[k + 1]; p++) {
            int i = G->Ai[p];
            G->Y[i] += G->Ax[p];
            int len = 0;
            for (; G->flag[i] != k; i = G->parent[i]) { G->pattern[len++] = i; G->flag[i] = k; }
            while (len > 0) G->pattern[--top] = G->pattern[--len];
        }
        G->D[k] = G->Y[k]; G->Y[k] = 0.0;
        for (; top < N; top++) {
            const int i = G->pattern[top];
            const double yi = G->Y[i];
            G->Y[i] = 0.0;
            const int p2 = G->Lp[i] + G->Lnz[i];
            for (int p = G->Lp[i]; p < p2; p++) G->Y[G->Li[p]] -= G->Lx[p] * yi;
            const double lki = yi / G->D[i];
            G->D[k] -= lki * yi;
            G->Li[p2] = k; G->Lx[p2] = lki; G->Lnz[i]++;
        }
    }
}
static void gen_solve(const gen_t* G, double* b)
{
    const int N = G->N;
    for (int j = 0; j < N; j++) { const double bj = b[j]; for (int p = G->Lp[j]; p < G->Lp[j + 1]; p++) b[G->Li[p]] -= G->Lx[p] * bj; }
    for (int j = 0; j < N; j++) b[j] /= G->D[j];
    for (int j = N - 1; j >= 0; j--) { double s = b[j]; for (int p = G->Lp[j]; p < G->Lp[j + 1]; p++) s -= G->Lx[p] * b[G->Li[p]]; b[j] = s; }
}

/* K = [Q + dprim I, Ea'; Ea, -diag(ddual)] in the ordering perm; rows with use[r] == 0 are decoupled (diagonal -1).  Positions >= f->N
 * are border nodes: their couplings go to U (with band nodes) and S (among themselves). */
static void kkt_put(band_t* f, int pa, int pb, double v)
{
    const int Nb = f->N, w = f->w, ld = w + 1, kb = f->kb;
    const int hi = pa > pb ? pa : pb, lo = pa > pb ? pb : pa;
    if (hi < Nb) f->B[(size_t)hi * ld + w - (hi - lo)] += v;
    else if (lo < Nb) f->U[(size_t)(hi - Nb) * Nb + lo] += v;
    else { f->S[(size_t)(hi - Nb) * kb + (lo - Nb)] += v; if (hi != lo) f->S[(size_t)(lo - Nb) * kb + (hi - Nb)] += v; }
}
static void kkt_assemble(const sqp_t* q, band_t* f, double dprim, const double* ddual, const int* use)
{
    if (f->G) {
        gen_t* G = f->G;
        const int n = q->n, m = q->m;
        memset(G->Ax, 0, sizeof(double) * (size_t)G->nnzA);
        for (int i = 0; i < n; i++) {
            for (int k = q->Qp[i]; k < q->Qp[i + 1]; k++) if (G->posQ[k] >= 0) G->Ax[G->posQ[k]] += q->Qx[k];
            G->Ax[G->posD[q->iperm[i]]] += dprim;
        }
        for (int r = 0; r < m; r++) {
            const int pd = G->posD[q->iperm[n + r]];
            if (use && !use[r]) { G->Ax[pd] = -1.0; continue; }
            G->Ax[pd] = -ddual[r];
            for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) G->Ax[G->posE[k]] += q->Ex[k];
        }
        return;
    }
    const int n = q->n, m = q->m, w = q->w, ld = w + 1, kb = f->kb;
    memset(f->B, 0, sizeof(double) * (size_t)f->N * ld);
    if (kb > 0) { memset(f->U, 0, sizeof(double) * (size_t)kb * f->N); memset(f->S, 0, sizeof(double) * (size_t)kb * kb); }
    for (int i = 0; i < n; i++) {
        const int pi = q->iperm[i];
        for (int k = q->Qp[i]; k < q->Qp[i + 1]; k++) {
            const int pj = q->iperm[q->Qi[k]];
            if (pj <= pi) kkt_put(f, pi, pj, q->Qx[k]);
        }
        kkt_put(f, pi, pi, dprim);
    }
    for (int r = 0; r < m; r++) {
        const int pr = q->iperm[n + r];
        if (use && !use[r]) { kkt_put(f, pr, pr, -1.0); continue; }
        kkt_put(f, pr, pr, -ddual[r]);
        for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) kkt_put(f, pr, q->iperm[q->Ei[k]], q->Ex[k]);
    }
}
/* the factorisation of the bordered matrix: band LDL' of B, W = U inv(B) row by row, S = C - W U', dense LDL' of S in place */
static void kkt_factor(band_t* f)
{
    if (f->G) { gen_factor(f->G); return; }
    const int Nb = f->N, kb = f->kb;
    band_factor(f);
    for (int b = 0; b < kb; b++) {
        double* wb = f->W + (size_t)b * Nb;
        memcpy(wb, f->U + (size_t)b * Nb, sizeof(double) * Nb);
        band_solve(f, wb);
    }
    for (int a = 0; a < kb; a++)
        for (int b = 0; b < kb; b++) {
            double s = 0;
            const double *ua = f->U + (size_t)a * Nb, *wb = f->W + (size_t)b * Nb;
            for (int p = 0; p < Nb; p++) if (ua[p] != 0.0) s += ua[p] * wb[p];
            f->S[(size_t)a * kb + b] -= s;
        }
    for (int j = 0; j < kb; j++) {       /* S = L D L': L below the diagonal, D on it */
        double d = f->S[(size_t)j * kb + j];
        for (int k = 0; k < j; k++) d -= f->S[(size_t)j * kb + k] * f->S[(size_t)j * kb + k] * f->S[(size_t)k * kb + k];
        f->S[(size_t)j * kb + j] = d;
        for (int i = j + 1; i < kb; i++) {
            double v = f->S[(size_t)i * kb + j];
            for (int k = 0; k < j; k++) v -= f->S[(size_t)i * kb + k] * f->S[(size_t)j * kb + k] * f->S[(size_t)k * kb + k];
            f->S[(size_t)i * kb + j] = v / d;
        }
    }
}
/* K x = b in place (b in the ordering perm: band nodes first, then the border) */
static void kkt_solve(const band_t* f, double* b)
{
    if (f->G) { gen_solve(f->G, b); return; }
    const int Nb = f->N, kb = f->kb;
    band_solve(f, b);
    if (kb == 0) return;
    double t[64];
    for (int a = 0; a < kb; a++) {
        double s = b[Nb + a];
        const double* ua = f->U + (size_t)a * Nb;
        for (int p = 0; p < Nb; p++) if (ua[p] != 0.0) s -= ua[p] * b[p];
        t[a] = s;
    }
    for (int i = 0; i < kb; i++) for (int k = 0; k < i; k++) t[i] -= f->S[(size_t)i * kb + k] * t[k];
    for (int i = 0; i < kb; i++) t[i] /= f->S[(size_t)i * kb + i];
    for (int i = kb - 1; i >= 0; i--) for (int k = i + 1; k < kb; k++) t[i] -= f->S[(size_t)k * kb + i] * t[k];
    for (int a = 0; a < kb; a++) {
        const double* wa = f->W + (size_t)a * Nb;
        for (int p = 0; p < Nb; p++) b[p] -= wa[p] * t[a];
        b[Nb + a] = t[a];
    }
}

/* ---- subsolver ---------------------------------------------------------------------------------------------------------- */
static void sqp_free(sqp_t* q)
{
    free(q->iperm); free(q->l); free(q->u); free(q->rhov); free(q->Ka.B); free(q->Kp.B); free(q->Ka.Uf); free(q->Kp.Uf); gen_free(q->Ka.G); gen_free(q->Kp.G); free(q->Ka.U); free(q->Ka.W); free(q->Ka.S); free(q->Kp.U); free(q->Kp.W); free(q->Kp.S); free(q->stf); free(q->x); free(q->y);
    free(q->st); free(q->xa); free(q->ya); free(q->za); free(q->r1); free(q->ex); free(q->wN); free(q->r1_last); free(q->ex_last);
    free(q->g_last); free(q->newst);
}

static int sqp_setup(sqp_t* q, const double* lbE, const double* ubE)
{
    const int n = q->n, m = q->m, N = n + m;
    const orc_options_t* o = &q->opt;
    q->N = N;
    q->iperm = ial(N);
    for (int p = 0; p < N; p++) q->iperm[q->perm[p]] = p;
    q->l = dal(m); q->u = dal(m); q->rhov = dal(m);
    memcpy(q->l, lbE, sizeof(double) * m); memcpy(q->u, ubE, sizeof(double) * m);
    double scale = 0;
    for (int i = 0; i < n; i++) for (int k = q->Qp[i]; k < q->Qp[i + 1]; k++) if (q->Qi[k] == i && fabs(q->Qx[k]) > scale) scale = fabs(q->Qx[k]);
    if (!(scale > 1e-300)) scale = 1.0;
    q->scale = scale;
    q->sigma = o->admmSigma * scale;
    q->delta = o->proxBig * scale;          /* primal regularisation of the polish KKT matrix */
    q->delta2 = 1e-9 / scale;               /* dual regularisation: makes the matrix quasi-definite whatever the rank of Ea */
    /* Round 3: two levels, as on the dense path (proxSmall / proxBig).  A correction with the safe level leaves delta * dx and delta2 * dy
     * (1e-8, 1e-9 relative) in the true residuals, so every QP paid one refinement trial -- a sweep, a band solve and the vector passes --
     * for the regularisation alone.  The light level leaves 1e-12 / 1e-14 and the first correction is accepted.  The band LDL' is not
     * pivoted, so the light level is only kept when every pivot has the sign its node prescribes (variables +, active rows -) and
     * a safe size; otherwise this factorisation uses the safe level -- and every later one of the problem when a variable's pivot was the
     * reason (a Hessian that is only semidefinite). */
    q->delta_s = o->proxSmall * scale; q->delta2_s = 1e-14 / scale; q->big_reg = 0;
    /* The light level is not even tried (big_reg from the start) unless the Hessian is safely definite by its diagonal
     * (min Q_ii >= 1e-6 max Q_ii) and the ordering puts every band row behind one of its variables -- a row eliminated before all of them
     * has the bare -delta2 as its pivot.  (lcqp_hip_sparse.hip: sp_choose_ordering makes the same test, there over the whole batch.) */
    {
        double dmin = INFINITY;
        for (int i = 0; i < n; i++) {
            double qii = 0.0;
            for (int k = q->Qp[i]; k < q->Qp[i + 1]; k++) if (q->Qi[k] == i) qii = q->Qx[k];
            if (qii < dmin) dmin = qii;
        }
        int follow = 1;
        for (int r = 0; r < m; r++) {
            if (q->iperm[n + r] >= N - q->kb) continue;
            int f = 0;
            for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) f |= (q->iperm[q->Ei[k]] < q->iperm[n + r]);
            follow &= f;
        }
        if (!(dmin >= 1e-6 * scale) || !follow) q->big_reg = 1;
    }
    const double rho = o->admmRho * scale;
    double* dd = dal(m);
    for (int r = 0; r < m; r++) {
        if (isinf(q->l[r]) && isinf(q->u[r])) q->rhov[r] = 1e-6 * rho;          /* OSQP's RHO_MIN for unconstrained rows */
        else if (q->l[r] == q->u[r]) q->rhov[r] = rho * o->rhoEqMult;
        else q->rhov[r] = rho;
        dd[r] = 1.0 / q->rhov[r];
    }
    const int kb = q->kb, Nb = N - kb;
    q->Ka.N = q->Kp.N = Nb; q->Ka.w = q->Kp.w = q->w; q->Ka.kb = q->Kp.kb = kb;
    if (q->w < 0) {      /* general sparse LDL' (no band, no border): one symbolic analysis per factor object */
        q->Ka.G = gen_setup(n, m, q->Qp, q->Qi, q->Ep, q->Ei, q->iperm);
        q->Kp.G = gen_setup(n, m, q->Qp, q->Qi, q->Ep, q->Ei, q->iperm);
        if (!q->Ka.G || !q->Kp.G) { free(dd); return 3; }
        q->Ka.w = q->Kp.w = 0;
    }
    const int wst = q->w < 0 ? 0 : q->w;
    q->Ka.B = dal((size_t)Nb * (wst + 1)); q->Kp.B = dal((size_t)Nb * (wst + 1));
    q->Ka.Uf = dal((size_t)Nb * (wst + 1)); q->Kp.Uf = dal((size_t)Nb * (wst + 1));
    q->Ka.U = dal((size_t)kb * Nb); q->Ka.W = dal((size_t)kb * Nb); q->Ka.S = dal((size_t)kb * kb);
    q->Kp.U = dal((size_t)kb * Nb); q->Kp.W = dal((size_t)kb * Nb); q->Kp.S = dal((size_t)kb * kb);
    if (!q->Ka.B || !q->Kp.B || !q->Ka.Uf || !q->Kp.Uf || !q->Ka.U || !q->Ka.W || !q->Ka.S || !q->Kp.U || !q->Kp.W || !q->Kp.S) { free(dd); return 3; }   /* out of memory: a setup failure, not a NULL dereference */
    kkt_assemble(q, &q->Ka, q->sigma, dd, NULL);
    kkt_factor(&q->Ka);
    free(dd);
    q->stf = ial(m); q->stf_valid = 0;
    q->x = dal(n); q->y = dal(m); q->st = ial(m);
    q->xa = dal(n); q->ya = dal(m); q->za = dal(m);
    q->r1 = dal(n); q->ex = dal(m); q->wN = dal(N); q->r1_last = dal(n); q->ex_last = dal(m); q->g_last = dal(n);
    q->newst = ial(m);
    return 0;
}

/* n_it ADMM iterations (OSQP, KKT form): [Q + sigma I, E'; E, -1/rho][xt; nu] = [sigma x - g; z - y/rho], zt = z + (nu - y)/rho */
static void sqp_admm(sqp_t* q, const double* g, int n_it)
{
    const int n = q->n, m = q->m;
    const double alpha = q->opt.admmAlpha;
    double* b = q->wN;
    for (int it = 0; it < n_it; it++) {
        for (int i = 0; i < n; i++) b[q->iperm[i]] = q->sigma * q->xa[i] - g[i];
        for (int r = 0; r < m; r++) b[q->iperm[n + r]] = q->za[r] - q->ya[r] / q->rhov[r];
        kkt_solve(&q->Ka, b);
        for (int r = 0; r < m; r++) {
            const double rv = q->rhov[r];
            const double zt = q->za[r] + (b[q->iperm[n + r]] - q->ya[r]) / rv;
            const double zr = alpha * zt + (1.0 - alpha) * q->za[r];
            if (isinf(q->l[r]) && isinf(q->u[r])) { q->za[r] = zr; q->ya[r] = 0.0; continue; }
            const double zn = clipd(zr + q->ya[r] / rv, q->l[r], q->u[r]);
            q->ya[r] += rv * (zr - zn);
            q->za[r] = zn;
        }
        for (int i = 0; i < n; i++) q->xa[i] = alpha * b[q->iperm[i]] + (1.0 - alpha) * q->xa[i];
        q->c_admm++;
    }
}

/* primal-dual active-set polish in correction form (the sparse twin of qp_polish in lcqp_oracle.c) */
static int sqp_polish(sqp_t* q, const double* g, double* x, double* y, int* st, int reuse)
{
    const int n = q->n, m = q->m;
    const orc_options_t* o = &q->opt;
    double gmax = 0;
    for (int i = 0; i < n; i++) if (fabs(g[i]) > gmax) gmax = fabs(g[i]);
    const double gs = 1.0 + gmax;
    double *r1 = q->r1, *Ex = q->ex, *b = q->wN;
    int fact_valid = 0, nrefine = 0;
    double xinf = 0.0;      /* |x|_inf behind the last correction */
    double e1max = 0.0;     /* largest row 1-norm of E: with |x|_inf the scale of the rounding of a computed E_r x */
    for (int r = 0; r < m; r++) { double s1 = 0.0; for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) s1 += fabs(q->Ex[k]); if (s1 > e1max) e1max = s1; }
    for (int trial = 0; trial < o->maxTrials; trial++) {
        q->c_trials++;
        /* Two stages, as on the dense path (round 3).  Stage 1 is what every trial needs: E x, for the status test.  Stage 2 -- the true
         * residual r1 = -g - Q x - E'y, one pass over Q and one over E' -- runs only when stage 1 changed nothing: after a correction the
         * residual is zero on the old working set up to rounding and regularisation, so when the set changes the next right-hand side is
         * known without it (the multipliers of the leaving rows, below); the trial that accepts always has the true residual. */
        int have_r1 = 0;
        if (trial == 0 && reuse) {
            for (int i = 0; i < n; i++) r1[i] = q->r1_last[i] + (q->g_last[i] - g[i]);
            memcpy(Ex, q->ex_last, sizeof(double) * m);
            have_r1 = 1;
        } else {
            sp_Ex(q, x, Ex);
        }
        double res_stat = 0, res_eq = 0, bmax = 0;
        int changed = 0, nact = 0, nloose = 0;
        const double ytol = o->feasTol * gs;
        /* active rows are held to the rounding floor of a computed E_r x before a point is accepted (round 5; the dense twin is qp_polish in
         * lcqp_oracle.c, the device's sp_ph_trial): runSolver ends on phi < 1e3 eps (src/LCQProblem.cpp:511-534, src/Options.cpp:297) */
        const double exScale = e1max * xinf;
        for (int r = 0; r < m; r++) {
            int s = st[r], ns = s;
            if (s == SP_INACT) {
                const double ftol = o->feasTol * (1.0 + fabs(Ex[r]));
                if (Ex[r] < q->l[r] - ftol) ns = SP_LOWER;
                else if (Ex[r] > q->u[r] + ftol) ns = SP_UPPER;
            } else {
                const double bb = (s == SP_UPPER) ? q->u[r] : q->l[r];
                const int above = !(fabs(bb - Ex[r]) <= 16.0 * 2.221e-16 * (fabs(bb) + exScale));      /* (a NaN counts) a row at the rounding floor of its computed E_r x is not a residual (round 6) */
                if (above && (fabs(bb - Ex[r]) > res_eq || Ex[r] != Ex[r])) res_eq = fabs(bb - Ex[r]);
                if (fabs(bb) > bmax) bmax = fabs(bb);
                if (above) nloose++;
                if (s == SP_LOWER && y[r] > ytol) ns = SP_INACT;
                if (s == SP_UPPER && y[r] < -ytol) ns = SP_INACT;
            }
            q->newst[r] = ns;
            if (ns != s) changed++;
            nact += (ns != SP_INACT);
        }
        double rscale = 0.0;      /* max_i(|g_i| + |Q x|_i + |E'y|_i): 64 eps of it is the residual's own rounding floor (round 6, as in qp_polish of lcqp_oracle.c) */
        if (!have_r1 && (trial == 0 || !changed)) {
            q->c_sweeps++;
            sp_Qx(q, x, r1);
            for (int i = 0; i < n; i++) { b[i] = r1[i]; r1[i] = -g[i] - r1[i]; }      /* b: scratch (the correction's right-hand side is formed later) */
            sp_ETy_sub(q, y, r1);
            for (int i = 0; i < n; i++) { const double t3 = fabs(g[i]) + fabs(b[i]) + fabs(-g[i] - b[i] - r1[i]); if (t3 > rscale) rscale = t3; }
            have_r1 = 1;
        }
        const double rtolS = fmax(o->resTol * gs, 64.0 * 2.221e-16 * rscale);
        if (have_r1) for (int i = 0; i < n; i++) if (fabs(r1[i]) > res_stat || r1[i] != r1[i]) res_stat = fabs(r1[i]);      /* a NaN stays (and is never accepted) */
        if (trial > 0 && !changed && res_stat <= rtolS && res_eq <= o->resTol * (1.0 + bmax) && nloose > 0 && nrefine < 2 && trial + 1 < o->maxTrials) {
            nrefine++;      /* a verified KKT point whose active rows can be held more exactly: one more correction */
        } else
        if (trial > 0 && !changed && res_stat <= rtolS && res_eq <= o->resTol * (1.0 + bmax)) {
            memcpy(q->r1_last, r1, sizeof(double) * n); memcpy(q->ex_last, Ex, sizeof(double) * m); memcpy(q->g_last, g, sizeof(double) * n);
            return 1;
        }
        if (changed && trial > 0) {
            if (trial >= 2 && nact > n && changed > (n / 2 > 32 ? n / 2 : 32)) return 0;       /* overshooting cold start: hand over to ADMM */
            if (!have_r1) for (int i = 0; i < n; i++) r1[i] = 0.0;        /* predicted: the last correction left nothing on the old working set */
            for (int r = 0; r < m; r++) {
                const int ns = q->newst[r];
                if (ns == SP_INACT && st[r] != SP_INACT && y[r] != 0.0) {
                    for (int k = q->Ep[r]; k < q->Ep[r + 1]; k++) r1[q->Ei[k]] += q->Ex[k] * y[r];     /* leaving row: its multiplier leaves the residual */
                    y[r] = 0.0;
                }
                st[r] = ns;
            }
            fact_valid = 0;
        }
        if (!fact_valid) {
            int same = q->stf_valid;
            for (int r = 0; r < m && same; r++) same = ((q->stf[r] != SP_INACT) == (st[r] != SP_INACT));
            if (!same) {
                int* use = q->newst;
                double* d2 = (double*)malloc(sizeof(double) * (m ? m : 1));
                for (int r = 0; r < m; r++) use[r] = (st[r] != SP_INACT);
                for (int level = q->big_reg; level < 2; level++) {
                    const double dp = level ? q->delta : q->delta_s, dd = level ? q->delta2 : q->delta2_s;
                    for (int r = 0; r < m; r++) d2[r] = dd;
                    kkt_assemble(q, &q->Kp, dp, d2, use);
                    kkt_factor(&q->Kp);
                    if (level == 1) break;
                    /* pivots of the band (the border is not examined): variables must be > 1e-8 scale, active rows < -1e-8 / scale */
                    int badVar = 0, badRow = 0;
                    const int Nb = q->Kp.N, ld = q->w + 1;
                    for (int pp = 0; pp < Nb; pp++) {
                        const double D = q->Kp.G ? q->Kp.G->D[pp] : q->Kp.B[(size_t)pp * ld + q->w];
                        const int node = q->perm[pp];
                        if (node < n) badVar |= !(D > 1e-8 * q->scale);
                        else if (use[node - n]) badRow |= !(D < -1e-8 / q->scale);
                    }
                    if (!badVar && !badRow) break;
                    if (badVar) q->big_reg = 1;      /* the Hessian is not safely definite: every later factorisation too; dependent rows come and go */
                }
                free(d2);
                memcpy(q->stf, st, sizeof(int) * m); q->stf_valid = 1;
                q->c_fact++;
            }
            fact_valid = 1;
        }
        /* correction: [Q + delta I, Ea'; Ea, -delta2 I][dx; dy] = [r1; ba - Ea x] */
        for (int i = 0; i < n; i++) b[q->iperm[i]] = r1[i];
        for (int r = 0; r < m; r++) {
            double v = 0.0;
            if (st[r] != SP_INACT) v = ((st[r] == SP_UPPER) ? q->u[r] : q->l[r]) - Ex[r];
            b[q->iperm[n + r]] = v;
        }
        kkt_solve(&q->Kp, b);
        xinf = 0.0;
        for (int i = 0; i < n; i++) { x[i] += b[q->iperm[i]]; if (fabs(x[i]) > xinf) xinf = fabs(x[i]); }
        for (int r = 0; r < m; r++) if (st[r] != SP_INACT) y[r] += b[q->iperm[n + r]];
        q->c_corr++;
    }
    return 0;
}

static int sqp_solve(sqp_t* q, int initial, const double* g, const double* x0, const double* y0 /* reference sign, m */, int* iterations, int* flag)
{
    const int n = q->n, m = q->m;
    const orc_options_t* o = &q->opt;
    const int trials0 = q->c_trials, admm0 = q->c_admm;
    *iterations = 0; *flag = 0;
    for (int r = 0; r < m; r++) if (q->l[r] > q->u[r]) { *flag = 2; return ORC_SUBPROBLEM_SOLVER_ERROR; }
    if (initial) {
        for (int i = 0; i < n; i++) q->x[i] = x0 ? x0[i] : 0.0;
        for (int r = 0; r < m; r++) q->y[r] = y0 ? -y0[r] : 0.0;
    }
    memcpy(q->xa, q->x, sizeof(double) * n); memcpy(q->ya, q->y, sizeof(double) * m);
    int n_admm = initial ? o->admmFirst : o->admmHot;
    const int use_stored = (!initial && q->have_solution && n_admm == 0);
    int admm_ready = 0, solved = 0;
    double* xt = dal(n); double* yt = dal(m); int* stt = ial(m);
    for (int round = 0; round < o->maxRounds && !solved; round++) {
        if (!admm_ready && (n_admm > 0 || !(round == 0 && use_stored))) {
            sp_Ex(q, q->xa, q->za);
            for (int r = 0; r < m; r++) { q->za[r] = clipd(q->za[r], q->l[r], q->u[r]); if (isinf(q->l[r]) && isinf(q->u[r])) q->ya[r] = 0.0; }
            admm_ready = 1;
        }
        if (n_admm > 0) sqp_admm(q, g, n_admm);
        for (int r = 0; r < m; r++) {
            int s;
            if (round == 0 && use_stored) { s = q->st[r]; if (q->l[r] == q->u[r]) s = SP_EQ; }
            else {
                const double lo = q->l[r], hi = q->u[r], z = q->za[r], yy = q->ya[r];
                s = SP_INACT;
                if (isfinite(lo) && (z - lo < -yy)) s = SP_LOWER;
                if (isfinite(hi) && (hi - z < yy)) s = SP_UPPER;
                if (lo == hi) s = SP_EQ;
            }
            stt[r] = s;
            yt[r] = (s != SP_INACT) ? q->ya[r] : 0.0;
        }
        memcpy(xt, q->xa, sizeof(double) * n);
        if (sqp_polish(q, g, xt, yt, stt, round == 0 && use_stored)) { solved = 1; break; }
        n_admm = 2 * n_admm;
        if (n_admm < 10) n_admm = 10;
        if (n_admm > 400) n_admm = 400;
    }
    *iterations = (q->c_trials - trials0) + (q->c_admm - admm0);
    if (solved) { memcpy(q->x, xt, sizeof(double) * n); memcpy(q->y, yt, sizeof(double) * m); memcpy(q->st, stt, sizeof(int) * m); q->have_solution = 1; }
    free(xt); free(yt); free(stt);
    if (!solved) { *flag = 1; return ORC_SUBPROBLEM_SOLVER_ERROR; }
    return ORC_SUCCESSFUL_RETURN;
}

/* ---- LCQProblem::runSolver on the OSQP_SPARSE arm ----------------------------------------------------------------------- */
int orc_sparse_lcqp_solve(int nV, int nC, int nComp,
                          const int* Qp, const int* Qi, const double* Qx, const double* g,
                          const int* Ep, const int* Ei, const double* Ex,
                          const double* lbA, const double* ubA, const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                          const double* x0, const double* y0, const int* perm, int w, int kb,
                          const orc_options_t* opt, double* xOpt, double* yOpt, orc_stats_t* stats)
{
    const int n = nV, m = nC + 2 * nComp;
    memset(stats, 0, sizeof(*stats));
    sqp_t Q_; memset(&Q_, 0, sizeof(Q_));
    sqp_t* q = &Q_;
    q->n = n; q->m = m; q->nC = nC; q->nComp = nComp; q->w = w; q->kb = kb; q->perm = perm;
    q->Qp = Qp; q->Qi = Qi; q->Qx = Qx; q->Ep = Ep; q->Ei = Ei; q->Ex = Ex; q->opt = *opt;
    /* stacked bounds: setConstraints / setComplementarityBounds (src/LCQProblem.cpp:585-608, 726-785) */
    double *lE = dal(m), *uE = dal(m);
    for (int i = 0; i < nC; i++) { lE[i] = lbA ? lbA[i] : -INFINITY; uE[i] = ubA ? ubA[i] : INFINITY; }
    for (int i = 0; i < nComp; i++) {
        lE[nC + i] = lbL ? lbL[i] : 0.0; uE[nC + i] = ubL ? ubL[i] : INFINITY;
        lE[nC + nComp + i] = lbR ? lbR[i] : 0.0; uE[nC + nComp + i] = ubR ? ubR[i] : INFINITY;
    }
    if (sqp_setup(q, lE, uE) != 0) { free(lE); free(uE); sqp_free(q); stats->qpSolverExitFlag = 3; stats->returnValue = ORC_SUBPROBLEM_SOLVER_ERROR; return ORC_SUBPROBLEM_SOLVER_ERROR; }   /* out of memory */
    double *xk = dal(n), *yk = dal(m), *pk = dal(n), *xnew = dal(n), *gk = dal(n), *gtil = dal(n), *gphi = dal(n), *statk = dal(n);
    double *Qxv = dal(n), *Cxv = dal(n), *Qpv = dal(n), *Cpv = dal(n), *lx = dal(m), *tmp = dal(n);
    const int hasPhi = (lbL != NULL) || (lbR != NULL);
    double phiConst = 0.0;
    if (hasPhi) {       /* :969-996: phi_const = lbL'lbR, g_phi = -(R'lbL + L'lbR) */
        for (int i = 0; i < nComp; i++) phiConst += (lbL ? lbL[i] : 0.0) * (lbR ? lbR[i] : 0.0);
        for (int i = 0; i < nComp; i++) {
            const double a = lbL ? lbL[i] : 0.0, bq = lbR ? lbR[i] : 0.0;
            for (int k = Ep[nC + nComp + i]; k < Ep[nC + nComp + i + 1]; k++) gphi[Ei[k]] -= Ex[k] * a;      /* R'lbL */
            for (int k = Ep[nC + i]; k < Ep[nC + i + 1]; k++) gphi[Ei[k]] -= Ex[k] * bq;                    /* L'lbR */
        }
    }
    if (x0) memcpy(xk, x0, sizeof(double) * n);
    memcpy(gtil, g, sizeof(double) * n);
    double rho = opt->initialPenaltyParameter, alphak = 1.0;
    double hist[64]; int histLen = 0, rc = 0, qpIter = 0, flag = 0, totalIter = 0, algoStat = 0;
    uint64_t perturbCounter = 0;
#define SP_CPROD(v, out) do { sp_Ex(q, (v), lx); for (int i_ = 0; i_ < n; i_++) (out)[i_] = 0.0;                                       \
        for (int i_ = 0; i_ < nComp; i_++) {                                                                                          \
            const double Lv = lx[nC + i_], Rv = lx[nC + nComp + i_];                                                                  \
            for (int k_ = Ep[nC + i_]; k_ < Ep[nC + i_ + 1]; k_++) (out)[Ei[k_]] += Ex[k_] * Rv;                  /* L'(R v) */       \
            for (int k_ = Ep[nC + nComp + i_]; k_ < Ep[nC + nComp + i_ + 1]; k_++) (out)[Ei[k_]] += Ex[k_] * Lv; /* R'(L v) */       \
        } } while (0)
#define SP_PHI(dst) do { double s_ = 0; for (int i_ = 0; i_ < n; i_++) s_ += (hasPhi ? gphi[i_] * xk[i_] : 0.0) + 0.5 * xk[i_] * Cxv[i_]; (dst) = phiConst + s_; } while (0)
    if (opt->solveZeroPenaltyFirst) memcpy(gk, g, sizeof(double) * n);
    else { SP_CPROD(xk, Cxv); for (int i = 0; i < n; i++) gk[i] = rho * Cxv[i] + gtil[i]; }
    int initial = 1;
    for (;;) {
        rc = sqp_solve(q, initial, gk, xk, initial ? y0 : NULL, &qpIter, &flag);            /* :1115-1148 */
        stats->subproblemIter += qpIter; stats->qpSolverExitFlag = flag; stats->qpSolves++;
        if (rc != 0) break;
        memcpy(xnew, q->x, sizeof(double) * n);
        for (int r = 0; r < m; r++) yk[r] = -q->y[r];                                        /* src/SubsolverOSQP.cpp:196-199 */
        for (int i = 0; i < n; i++) pk[i] = xnew[i] - xk[i];
        if (initial) stats->rhoOpt = rho;
        else if (opt->perturbStep) {
            for (int i = 0; i < n; i++) {
                uint64_t z = opt->perturbSeed + (perturbCounter + (uint64_t)i + 1ULL) * 0x9E3779B97F4A7C15ULL;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z = z ^ (z >> 31);
                xk[i] += ((int)(z % 3ULL) - 1) * 2.221e-16;
            }
            perturbCounter += (uint64_t)n;
        }
        sp_Qx(q, pk, Qpv); sp_Qx(q, xk, Qxv); SP_CPROD(pk, Cpv); SP_CPROD(xk, Cxv);
        if (!initial) {                                                                       /* getOptimalStepLength :1217-1237 */
            double qk = 0, lk = 0;
            for (int i = 0; i < n; i++) { qk += pk[i] * (Qpv[i] + rho * Cpv[i]); lk += pk[i] * ((Qxv[i] + rho * Cxv[i]) + gtil[i]); }
            alphak = 1.0;
            if (qk > 0 && lk < 0) alphak = fmin(-lk / qk, 1.0);
        }
        initial = 0;
        for (int i = 0; i < n; i++) { xk[i] += alphak * pk[i]; Qxv[i] += alphak * Qpv[i]; Cxv[i] += alphak * Cpv[i]; }      /* updateStep */
        /* updateStationarity :1246-1272 without the box term (lb = ub = NULL on this arm) */
        for (int i = 0; i < n; i++) tmp[i] = 0.0;
        for (int r = 0; r < m; r++) { const double yr = yk[r]; if (yr != 0.0) for (int k = Ep[r]; k < Ep[r + 1]; k++) tmp[Ei[k]] += Ex[k] * yr; }
        double statInf = 0;
        for (int i = 0; i < n; i++) { statk[i] = (Qxv[i] + rho * Cxv[i]) + gtil[i] - tmp[i]; if (fabs(statk[i]) > statInf) statInf = fabs(statk[i]); }
        totalIter++; stats->iterTotal++;
        int leyffer = 0;                                                                      /* :1275-1313 */
        const int nd = opt->nDynamicPenalty;
        if (nd > 0) {
            double cur; SP_PHI(cur);
            if (histLen < nd) hist[histLen++] = cur;
            else {
                if (!(cur < opt->complementarityTolerance)) { leyffer = 1; for (int i = 0; i < nd; i++) if (cur < opt->etaDynamicPenalty * hist[i]) { leyffer = 0; break; } }
                for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1];
                hist[nd - 1] = cur;
            }
        }
#define SP_PENALTY() do { if (nd > 0) histLen = 0; rho *= opt->penaltyUpdateFactor; stats->rhoOpt = rho;                               \
        if (hasPhi) for (int i_ = 0; i_ < n; i_++) gtil[i_] = g[i_] + rho * gphi[i_]; } while (0)
        if (leyffer) { SP_PENALTY(); stats->iterOuter++; }
        if (statInf < opt->stationarityTolerance) {
            double phi; SP_PHI(phi);
            if (phi < opt->complementarityTolerance) {
                sp_Ex(q, xk, lx);
                int sflag = 1, mflag = 1, wflag = 0;                                          /* :1412-1482 on the untransformed duals */
                const double ctol = opt->complementarityTolerance;
                for (int i = 0; i < nComp; i++) {
                    const double Lx = lx[nC + i], Rx = lx[nC + nComp + i];
                    if (!(Lx <= ctol && Rx <= ctol)) continue;
                    const double a = yk[nC + i], bq = yk[nC + nComp + i];
                    const double dualProd = a * bq, dualMin = fmin(a, bq);
                    if (dualMin < 0) sflag = 0;
                    if (fabs(dualProd) >= ctol && dualMin <= 0) { if (dualProd <= ctol) { wflag = 1; break; } mflag = 0; }
                }
                algoStat = wflag ? 1 : (sflag ? 4 : (mflag ? 3 : 2));
                for (int i = 0; i < nComp; i++) { yk[nC + i] -= rho * lx[nC + nComp + i]; yk[nC + nComp + i] -= rho * lx[nC + i]; }   /* transformDuals :1381-1409 */
                rc = 0;
                break;
            }
            SP_PENALTY(); stats->iterOuter++;
        }
        if (totalIter > opt->maxIterations) { rc = ORC_MAX_ITERATIONS_REACHED; break; }
        if (rho > opt->maxPenaltyParameter) { rc = ORC_MAX_PENALTY_REACHED; break; }
        for (int i = 0; i < n; i++) gk[i] = rho * Cxv[i] + gtil[i];                          /* updateLinearization :1105-1112 */
    }
    stats->status = algoStat; stats->returnValue = rc;
    stats->admmIter = q->c_admm; stats->trials = q->c_trials; stats->factorizations = q->c_fact; stats->corrections = q->c_corr; stats->reserved = q->c_sweeps;
    if (xOpt) memcpy(xOpt, xk, sizeof(double) * n);
    if (yOpt) memcpy(yOpt, yk, sizeof(double) * m);
    free(xk); free(yk); free(pk); free(xnew); free(gk); free(gtil); free(gphi); free(statk); free(Qxv); free(Cxv); free(Qpv); free(Cpv);
    free(lx); free(tmp); free(lE); free(uE);
    sqp_free(q);
    return rc;
}
