// lcqp_kernels.hpp -- the kernels that are instantiated once per padded size np = 128*NCH, and their launcher.
// Each instantiation is its own translation unit (lcqp_nch.hip compiled with -DLCQP_TU_NCH=1,2,3,4,8): the five builds run in
// parallel, and the register allocation of one size cannot perturb another's (cdna_hip_programming.md §5.4 rule 19).
#pragma once
#include "lcqp_dev.hpp"
#include "lcqp_launch.hpp"
#include "../../include/lcqp_synth.h"

using namespace lcqp;

#define LCQP_LDS_N(NCHV)                                    \
    __shared__ double sh_arena[arena_doubles(NCHV)];        \
    __shared__ double sh_red[16];                           \
    __shared__ int sh_ired[16];                             \
    Lds lds{sh_arena, sh_red, sh_ired};
#define LCQP_LDS LCQP_LDS_N(4)


// ---- k_prepare: scales, padding, box rows, ADMM rho vector, phi expressions ----------------------
template <int NCH>
__global__ __launch_bounds__(WG) void k_prepare(DevBatch db)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x, t = threadIdx.x;
    Ctx<NCH> c = make_ctx<NCH>(db, b, lds);
    const int n = db.n, mA = db.mA, nC = db.nC, nComp = db.nComp;
    double dmax = 0.0;
    for (int i = t; i < n; i += WG) dmax = fmax(dmax, fabs(c.Q[(size_t)i * np + i]));
    double scale = block_max(dmax, lds);
    if (!(scale > 1e-300)) scale = 1.0;
    for (int i = n + t; i < np; i += WG) c.Q[(size_t)i * np + i] = 1.0;
    const int nfin = c.info->nfin;
    const int mE = mA + nfin;
    double *l = c.M(M_L), *u = c.M(M_U), *rhov = c.M(M_RHOV);
    for (int k = 0; k < nfin; k++) {
        double* row = c.E + (size_t)(mA + k) * np;
        const int bi = c.boxidx[k];
        for (int i = t; i < np; i += WG) row[i] = (i == bi) ? 1.0 : 0.0;
        if (t == 0) { l[mA + k] = c.V(V_LB)[bi]; u[mA + k] = c.V(V_UB)[bi]; }
    }
    __syncthreads();
    const double rho = db.opt.admmRho * scale;
    for (int r = t; r < mE; r += WG) {
        const double lo = l[r], hi = u[r];
        double rv = rho;
        if (isinf(lo) && isinf(hi)) rv = 0.0;
        else if (lo == hi) rv = rho * db.opt.rhoEqMult;
        rhov[r] = rv;
    }
    // phi expressions, src/LCQProblem.cpp:969-996
    double phiConst = 0.0;
    double* gphi = c.V(V_GPHI);
    if (db.hasLbL || db.hasLbR) {
        const double* lbL = db.lbL + (size_t)b * nComp;
        const double* lbR = db.lbR + (size_t)b * nComp;
        double s = 0.0;
        for (int i = t; i < nComp; i += WG) s += lbL[i] * lbR[i];
        phiConst = block_sum(s, lds);
        double* coef = c.M(M_COEF);
        for (int r = t; r < mA; r += WG) {
            double v = 0.0;
            if (r >= nC && r < nC + nComp) v = lbR[r - nC];          // L' * lbR
            else if (r >= nC + nComp) v = lbL[r - nC - nComp];       // R' * lbL
            coef[r] = v;
        }
        __syncthreads();
        wg_rows<NCH>(c.E, nullptr, mA, nullptr, nullptr, coef, lds, [&](int i, double sum) { gphi[i] = -sum; });
    } else {
        wg_fill(gphi, 0.0, np);
    }
    {   // no dependent-row flags or promotions survive a new setup (qp_polish<ROBUST>)
        int *dep = c.I(I_DEP), *prio = c.I(I_PRIO), *rslot = c.I(I_SLOT);
        for (int r = t; r < db.mEcap; r += WG) { dep[r] = 0; prio[r] = 0; rslot[r] = -1; }
        for (int a = t; a < db.capS; a += WG) c.idx[a] = -1;      // the inverse factor of the working-set matrix starts empty
    }
    if (t == 0) {
        c.info->prioCtr = 0;
        c.info->ndep = 0;
        c.info->mE = mE;
        c.info->scale = scale;
        c.info->sigma = db.opt.admmSigma * scale;
        c.info->rhoAdmm = rho;
        c.info->phiConst = phiConst;
        c.info->haveSolution = 0;
        c.info->nT = 0; c.info->ns = 0; c.info->cNnz = -1;
        c.info->setupFail = 0;
        c.info->isSetup = 1;
    }
}

// ---- k_build_C: C = L'R + R'L (Utilities::MatrixSymmetrizationProduct, src/Utilities.cpp:104-116) ----
// One 64 x 64 tile of the lower triangle per workgroup.  Both products advance in the same loop (four panels of 16 rows of L and R per step:
// half the steps, barriers and exposed loads of two products one after the other), and the mirrored tile goes through LDS so that its rows
// are stored contiguously (rounds 1 - 5: every lane its own 8 bytes at a stride of a row).  Each sum is the same chain as before.
template <int NCH>
__global__ __launch_bounds__(WG, 4) void k_build_C(DevBatch db)      // 128 registers: a workgroup fits where one instance of k_lcqp_run has finished
{
    constexpr int np = 128 * NCH, PL = TILE_PL;
    __shared__ double sP[4 * 16 * PL];      // panels of L_I, R_I, L_J, R_J; afterwards the tile for the mirrored store (64 x 65)
    static_assert(4 * 16 * PL >= 64 * 65, "the tile must fit where the panels were");
    const int ntile = db.nblk * (db.nblk + 1) / 2;
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = bid / ntile, tIdx = bid % ntile;
    int I, J;
    tri_tile(tIdx, I, J);
    const double* Lm = db.E + (size_t)b * db.mEcap * np + (size_t)db.nC * np;
    const double* Rm = Lm + (size_t)db.nComp * np;
    double* C = db.C + (size_t)b * np * np;
    const int nrows = db.nComp;
    const int t = tid_here(), kk = t >> 4, c4 = (t & 15) * 4;
    double a1[4][4], a2[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) a1[i][j] = a2[i][j] = 0.0;
    double2 p[4][2];
    auto fetch = [&](int r) {
        const double2 z{0.0, 0.0};
#pragma unroll
        for (int m = 0; m < 4; m++) p[m][0] = p[m][1] = z;
        if (r < nrows) {
            const double* src[4] = {Lm + (size_t)r * np + 64 * I + c4, Rm + (size_t)r * np + 64 * I + c4,
                                    Lm + (size_t)r * np + 64 * J + c4, Rm + (size_t)r * np + 64 * J + c4};
#pragma unroll
            for (int m = 0; m < 4; m++) { p[m][0] = *reinterpret_cast<const double2*>(src[m]); p[m][1] = *reinterpret_cast<const double2*>(src[m] + 2); }
        }
    };
    fetch(kk);
    for (int k0 = 0; k0 < nrows; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            *reinterpret_cast<double2*>(sP + m * 16 * PL + kk * PL + c4) = p[m][0];
            *reinterpret_cast<double2*>(sP + m * 16 * PL + kk * PL + c4 + 2) = p[m][1];
        }
        __syncthreads();
        if (k0 + 16 < nrows) fetch(k0 + 16 + kk);
        tile_panel(a1, sP, sP + 3 * 16 * PL);                    // L_I' R_J
        tile_panel(a2, sP + 16 * PL, sP + 2 * 16 * PL);          // R_I' L_J
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int li = tile_li(i, j), lj = tile_lj(i, j);
            const double v = a1[i][j] + a2[i][j];
            C[(size_t)(64 * I + li) * np + 64 * J + lj] = v;
            if (I != J) sP[li * 65 + lj] = v;
        }
    if (I == J) return;      // the tile on the diagonal is its own mirror image, entry by entry
    __syncthreads();
    for (int e = t; e < 64 * 64; e += WG) {
        const int r = e >> 6, cidx = e & 63;
        C[(size_t)(64 * J + r) * np + 64 * I + cidx] = sP[cidx * 65 + r];
    }
}

// ---- k_compress_C: C in compressed rows when it is sparse (one-hot L, R give 2 nComp non-zeros) ------------------------------------
// The homotopy applies C to one vector per iterate; from compressed rows that costs 12 bytes per non-zero instead of a sweep over
// np x np doubles.  More than capC non-zeros: cNnz = -1 and C stays a dense sweep.
template <int NCH>
__global__ __launch_bounds__(WG) void k_compress_C(DevBatch db)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x, t = threadIdx.x, l = lane_id(), w = wave_id();
    const double* C = db.C + (size_t)b * np * np;
    int* cp = db.Cp + (size_t)b * (np + 1);
    int* ci = db.Ci + (size_t)b * db.capC;
    double* cv = db.Cv + (size_t)b * db.capC;
    int* cnt = reinterpret_cast<int*>(lds.arena);        // np row counts, then row starts
    const int n = db.n;
    // (up to eight loads of a row, and two rows per wave, are in flight together: a wave that waited for every load by itself made this kernel
    // 0.49 ms of dependent round trips at B = 1024; np is a multiple of 2 NWAVE)
    constexpr int NC = (np / 64 < 8) ? np / 64 : 8;
    for (int r = w; r < np; r += 2 * NWAVE) {
        const int r2 = r + NWAVE;
        int k1 = 0, k2 = 0;
        for (int c0 = 0; c0 < np; c0 += 64 * NC) {
            double v1[NC], v2[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                v1[q] = (r < n) ? C[(size_t)r * np + c0 + 64 * q + l] : 0.0;
                v2[q] = (r2 < n) ? C[(size_t)r2 * np + c0 + 64 * q + l] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NC; q++) { k1 += __popcll(__ballot(v1[q] != 0.0)); k2 += __popcll(__ballot(v2[q] != 0.0)); }
        }
        if (l == 0) { cnt[r] = k1; cnt[r2] = k2; }
    }
    __syncthreads();
    {   // exclusive scan of the np counts: each thread a contiguous chunk
        constexpr int per = (np + WG - 1) / WG;
        const int r0 = t * per;
        int loc = 0;
        for (int q = 0; q < per; q++) if (r0 + q < np) loc += cnt[r0 + q];
        int incl = loc;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const int v = __shfl_up(incl, ofs, 64); if (l >= ofs) incl += v; }
        if (l == 63) lds.ired[8 + w] = incl;
        __syncthreads();
        int run = incl - loc;
        for (int ww = 0; ww < w; ww++) run += lds.ired[8 + ww];
        const int total = lds.ired[8] + lds.ired[9] + lds.ired[10] + lds.ired[11];
        __syncthreads();
        for (int q = 0; q < per; q++) if (r0 + q < np) { const int k = cnt[r0 + q]; cnt[r0 + q] = run; run += k; }
        __syncthreads();
        if (total > db.capC) { if (t == 0) db.info[b].cNnz = -1; return; }
        if (t == 0) { db.info[b].cNnz = total; cp[np] = total; }
    }
    for (int r = w; r < np; r += 2 * NWAVE) {
        const int r2 = r + NWAVE;
        int pos1 = cnt[r], pos2 = cnt[r2];
        if (l == 0) { cp[r] = pos1; cp[r2] = pos2; }
        for (int c0 = 0; c0 < np; c0 += 64 * NC) {
            double v1[NC], v2[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                v1[q] = (r < n) ? C[(size_t)r * np + c0 + 64 * q + l] : 0.0;
                v2[q] = (r2 < n) ? C[(size_t)r2 * np + c0 + 64 * q + l] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const unsigned long long m1 = __ballot(v1[q] != 0.0), m2 = __ballot(v2[q] != 0.0), below = (1ULL << l) - 1ULL;
                if (v1[q] != 0.0) { const int o = pos1 + __popcll(m1 & below); ci[o] = c0 + 64 * q + l; cv[o] = v1[q]; }
                if (v2[q] != 0.0) { const int o = pos2 + __popcll(m2 & below); ci[o] = c0 + 64 * q + l; cv[o] = v2[q]; }
                pos1 += __popcll(m1); pos2 += __popcll(m2);
            }
        }
    }
}

// ---- k_factor: the constant factorisation L1 of Q + sp I (L_K of the ADMM fallback is built on demand, qp_build_K) ----
// One workgroup per instance; its life is chains (16 x 16 sub-block factorisations, dependent tiles): ~0.35 ms alone, ~0.7 ms with four per CU.
// Round 6: profiles/round6/factor_phases*.log -- with 138 registers only three workgroups per CU were resident and the default batch took two
// rounds (a quarter of the workgroups started 0.6 ms late); the copy F1 = Q + sp I in front of the factorisation was 15 % of a workgroup's life.
// MINW = 4 holds the kernel to 128 registers: four workgroups per CU, a batch of more than three workgroups per CU in one residency round
// (-0.12 ms at B = 1024); MINW = 1 (138 registers) is 0.09 ms faster for one workgroup alone -- the launcher picks by the size of the batch.
template <int NCH, int MINW>
__global__ __launch_bounds__(WG, MINW) void k_factor(DevBatch db)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x, t = threadIdx.x;
    Ctx<NCH> c = make_ctx<NCH>(db, b, lds);
    const double scale = c.info->scale;
    __shared__ int sfail;
    int failed = 0;
    double spv = 0.0;
#ifdef LCQP_FACTOR_PROFILE
    unsigned long long fpv[16];
    for (int k = 0; k < 16; k++) fpv[k] = 0;
    fpv[15] = fpv[14] = clock64();
    fpv[13] = wall_clock64();
    unsigned long long* fp = fpv;
#else
    unsigned long long* fp = nullptr;
#endif
    for (int pass = 0; pass < 2; pass++) {
        spv = (pass == 0 ? db.opt.proxSmall : db.opt.proxBig) * scale;
        if (t == 0) sfail = 0;
        __syncthreads();
        FPROF(7);
        // D1 keeps every inverted diagonal block (dense lower) for the TRSM that forms Et
        double minpiv = INFINITY;
        {
            // factor block column by block column so each D block lands in its own slot of D1
            minpiv = wg_chol(c.F1, np, c.nblk, c.n, 0.0, c.D1, nullptr, &sfail, lds, 4096, fp, c.Q, spv);      // F1 = chol(Q + spv I), Q read in place
        }
        __syncthreads();
        failed = sfail;
        __syncthreads();   // every thread has read the flag before thread 0 of the retry pass clears it
        if (!failed && (pass == 1 || minpiv >= db.opt.pivotThreshold * scale)) break;
        if (pass == 1) break;
    }
    if (t == 0) {
        c.info->spv = spv;
        c.info->kReady = 0;          // L_K (ADMM fallback) is built by the first instance that needs it: qp_build_K
        c.info->rnReady = 0;         // row norms of E for the row screening of the residual sweeps: first sweep
        if (failed) c.info->setupFail = 3;
#ifdef LCQP_FACTOR_PROFILE
        fpv[8] = clock64() - fpv[14];      // the whole kernel as this workgroup saw it
        for (int k = 0; k < 9; k++) db.prof[(size_t)b * 16 + k] = fpv[k];
        db.prof[(size_t)b * 16 + 9] = wall_clock64();      // 100 MHz, the same counter on every XCD: when did this workgroup end
        db.prof[(size_t)b * 16 + 10] = fpv[13];            // and start
#endif
    }
}

// ---- k_trsm: Et = E L1^-T, 64 rows of E per workgroup ----------------------------------------------
// Block column J of the result is  Et_J = (E_J - sum_{K<J} Et_K L_JK') D_J'  (D_J: the inverted diagonal block of L1 from k_factor).
//
// trsm_rows_resident (np <= 256): the sums of ALL block columns still to come live in registers (wave w: rows 16w..16w+15, 16 doubles per lane
// and block column), and the block column just finished is the A operand of the products out of a wave-private LDS region -- so E is read
// once, Et written once, and only the blocks of L1 / D1 stream through the workgroup (two 16-deep panels in LDS, one barrier per panel).
// Before (rounds 1 - 5, still the form of the larger sizes below): block column J re-read Et_0..J-1 from memory and wrote Et_J twice:
// 10 GB of traffic per launch of the default workload for 3.2 GB of operands (pmc_fetch_size / pmc_write_size), 1.69 ms at 6 TB/s.
// Every element is the same chain of v_mfma_f64_16x16x4 (k ascending, four per instruction, from zero), so the bits are the same.
template <int NCH>
__device__ __forceinline__ void trsm_rows_resident(const DevBatch& db)
{
    constexpr int np = 128 * NCH, NB = 2 * NCH, PL = TILE_PL;
    __shared__ double sA[NWAVE * 1024];      // per wave: 64 k x 16 rows, entry (k, r) at 16 k + (r ^ swz(k)): conflict-free C-layout writes and A-operand reads
    __shared__ double sB[2][16 * PL];
    const int nrb = (db.mEcap + 63) / 64;
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);      // the row blocks of an instance share L1 and D1: one L2
    const int b = bid / nrb, rb = bid % nrb;
    if (64 * rb >= db.info[b].mE) return;
    const double* E = db.E + (size_t)b * db.mEcap * np + (size_t)(64 * rb) * np;
    double* Et = db.Et + (size_t)b * db.mEcap * np + (size_t)(64 * rb) * np;
    const double* F1 = db.F1 + (size_t)b * np * np;
    const double* D1 = db.D1 + (size_t)b * db.nblk * 4096;
    const int rows = min(64, db.mEcap - 64 * rb);
    const int t = tid_here(), lane = t & 63, w = t >> 6, il = lane & 15, kl = lane >> 4;
    double* A = sA + 1024 * w;
    auto swz = [](int k) { return ((k >> 1) & 7) << 1; };
    const int lr = t >> 2, kq = (t & 3) * 4;      // loader of the B panels: row lr of the block, four consecutive k
    // panel q (16 k) of block `blk` of column J: blk 0 is D_J, blk i > 0 is L_{J+i, J}
    auto src = [&](int J, int blk, int q) -> const double* {
        return blk == 0 ? D1 + (size_t)J * 4096 + lr * 64 + 16 * q + kq
                        : F1 + (size_t)(64 * (J + blk) + lr) * np + 64 * J + 16 * q + kq;
    };
    double2 r0, r1;
    auto fetch = [&](const double* p) { r0 = *reinterpret_cast<const double2*>(p); r1 = *reinterpret_cast<const double2*>(p + 2); };
    auto commit = [&](int buf) {
        double* Bs = sB[buf];
        Bs[(kq + 0) * PL + lr] = r0.x; Bs[(kq + 1) * PL + lr] = r0.y; Bs[(kq + 2) * PL + lr] = r1.x; Bs[(kq + 3) * PL + lr] = r1.y;
    };
    // C layout of this wave's 16 x 64 block: x[a][q] is row (lane >> 4) + 4 q, column 16 a + (lane & 15)
    auto loadE = [&](d4_t (&x)[4], int J) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int li = 16 * w + kl + 4 * q;
                x[a][q] = (li < rows) ? E[(size_t)li * np + 64 * J + 16 * a + il] : 0.0;
            }
    };
    auto toA = [&](const d4_t (&x)[4]) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) { const int k = 16 * a + il, r = kl + 4 * q; A[16 * k + (r ^ swz(k))] = x[a][q]; }
    };
    d4_t acc[NB][4];      // acc[I]: sum_{K<J} Et_K L_IK' of the block columns I > J still to come
#pragma unroll
    for (int I = 0; I < NB; I++)
#pragma unroll
        for (int a = 0; a < 4; a++) acc[I][a] = d4_t{0.0, 0.0, 0.0, 0.0};
    d4_t cur[4];
    fetch(src(0, 0, 0));
    loadE(cur, 0);
    commit(0);
    fetch(src(0, 0, 1));
    toA(cur);
    __syncthreads();
    int pc = 0;
#pragma unroll
    for (int J = 0; J < NB; J++) {
#pragma unroll
        for (int blk = 0; blk < NB - J; blk++) {
            if (blk == 0) {
#pragma unroll
                for (int a = 0; a < 4; a++) cur[a] = d4_t{0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double* Bs = sB[pc & 1];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int k = 16 * q + 4 * k4 + kl;
                    const double av = A[16 * k + (il ^ swz(k))];
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const double bv = Bs[(4 * k4 + kl) * PL + 16 * a + il];
                        if (blk == 0) cur[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, cur[a], 0, 0, 0);
                        else acc[J + blk][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[J + blk][a], 0, 0, 0);
                    }
                }
                // the successors of this panel in the order of the loops
                int J1 = J, b1 = blk, q1 = q + 1;
                if (q1 == 4) { q1 = 0; b1++; if (b1 == NB - J1) { b1 = 0; J1++; } }
                if (J1 < NB) {
                    commit((pc + 1) & 1);
                    int J2 = J1, b2 = b1, q2 = q1 + 1;
                    if (q2 == 4) { q2 = 0; b2++; if (b2 == NB - J2) { b2 = 0; J2++; } }
                    if (J2 < NB) fetch(src(J2, b2, q2));
                }
                __syncthreads();
                pc++;
            }
            if (blk == 0) {
                // cur = Et_J: to memory, and in place of E_J - sum as the A operand of the products with the blocks of L1 below D_J
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int li = 16 * w + kl + 4 * q;
                        if (li < rows) Et[(size_t)li * np + 64 * J + 16 * a + il] = cur[a][q];
                    }
                toA(cur);
                if (J + 1 < NB) loadE(cur, J + 1);      // on its way while the products of this block column run
            }
        }
        if (J + 1 < NB) {
#pragma unroll
            for (int a = 0; a < 4; a++) cur[a] = cur[a] - acc[J + 1][a];
            toA(cur);
        }
    }
}

template <int NCH>
__device__ __forceinline__ void trsm_rows_streamed(const DevBatch& db)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int nrb = (db.mEcap + 63) / 64;
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);      // the row blocks of an instance share L1 and D1: one L2
    const int b = bid / nrb, rb = bid % nrb;
    const InstInfo* info = db.info + b;
    if (64 * rb >= info->mE) return;
    const double* E = db.E + (size_t)b * db.mEcap * np + (size_t)(64 * rb) * np;
    double* Et = db.Et + (size_t)b * db.mEcap * np + (size_t)(64 * rb) * np;
    const double* F1 = db.F1 + (size_t)b * np * np;
    const double* D1 = db.D1 + (size_t)b * db.nblk * 4096;
    const int rows = min(64, db.mEcap - 64 * rb);
    auto rowok = [=](int r) { return (long)(r < rows ? r : -1); };
    auto ident = [](int r) { return (long)r; };
    for (int J = 0; J < db.nblk; J++) {
        double acc[4][4];
        if (J > 0) wg_tile_nt(acc, Et, np, rowok, F1 + (size_t)(64 * J) * np, np, ident, 64 * J, lds);
        else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int li = tile_li(i, j), gj = 64 * J + tile_lj(i, j);
                if (li < rows) Et[(size_t)li * np + gj] = E[(size_t)li * np + gj] - acc[i][j];
            }
        __syncthreads();
        double acc2[4][4];
        wg_tile_nt(acc2, Et + 64 * J, np, rowok, D1 + (size_t)J * 4096, 64, ident, 64, lds);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int li = tile_li(i, j), gj = 64 * J + tile_lj(i, j);
                if (li < rows) Et[(size_t)li * np + gj] = acc2[i][j];
            }
        __syncthreads();
    }
}

// 207 registers, two waves per SIMD (held to 168 -- three waves -- the kernel spills 32 registers and the setup is 0.17 ms slower)
template <int NCH>
__global__ __launch_bounds__(WG) void k_trsm(DevBatch db)
{
#ifndef LCQP_TILE_VALU
    if constexpr (NCH <= 2) trsm_rows_resident<NCH>(db);
    else
#endif
        trsm_rows_streamed<NCH>(db);
}

// the streamed form for every size: 52 registers and 36 KB of LDS, a workgroup fits where ONE instance of k_lcqp_run has finished
// (lcqp_hip_batch_set_overlapped)
template <int NCH>
__global__ __launch_bounds__(WG) void k_trsm_streamed(DevBatch db) { trsm_rows_streamed<NCH>(db); }

// ---- k_build_M: M = Et Et', every entry of every working-set matrix S_W = Et_W Et_W' (lower triangle; readers take M[max][min]) ----
// fp64 MFMA.  Tiles of 128 rows x 64 columns (round 6; rounds 3 - 5: 128 x 128 with a 64 x 64 quadrant per wave -- 128 accumulator registers of
// 226, two waves per SIMD, the matrix pipe busy 62 % of the time): the four waves own 64 x 32 blocks as 4 x 2 blocks of v_mfma_f64_16x16x4_f64,
// 64 accumulator registers, so that three to four workgroups share a CU and another wave has products to issue while one waits at a barrier.  One
// 16-deep panel pair in LDS (k-major, pitches 144 / 80 doubles: conflict-free operand reads); the panels of step k + 16 are fetched into registers
// while the products of step k run.  Column blocks inside the row block (the tiles on the diagonal) take their operand from the row panel.
// Every element is the same chain of instructions as before (k ascending, four per instruction, from zero): the bits are the same.
// Tile t of an instance: row block I (128 rows), column block J (64 columns), J <= 2 I + 1; tiles are numbered row block by row block.
template <int NCH>
__global__ __launch_bounds__(WG, 4) void k_build_M(DevBatch db)
{
    constexpr int np = 128 * NCH;
    constexpr int PA = 144, PB = 80;
    __shared__ double As[16 * PA], Bs[16 * PB];
    const int nb = (db.mMld + 127) / 128, ntile = nb * (nb + 1);
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);      // the tiles of an instance read the same Et (1.3 MB): one L2
    const int b = bid / ntile, tIdx = bid % ntile;
    int I = 0;
    while ((I + 1) * (I + 2) <= tIdx) I++;
    const int J = tIdx - I * (I + 1);
    const int mE = db.info[b].mE, ld = db.mMld;
    if (128 * (J >> 1) >= mE || 64 * J >= ld) return;       // a block of 128 columns beyond the rows in use (then the rows are too)
    const double* Et = db.Et + (size_t)b * db.mEcap * np;
    double* M = db.MM + (size_t)b * ld * ld;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wy = w >> 1, wx = w & 1;
    const int lr = t >> 1, kh = (t & 1) * 8;      // loader of the row panel: row lr of the tile, eight consecutive k
    const int lc = t >> 2, kq = (t & 3) * 4;      // loader of the column panel: row lc of the column block, four consecutive k
    const bool inside = (J >> 1) == I;            // the column block is part of the row block
    const double* ap = (128 * I + lr < mE) ? Et + (size_t)(128 * I + lr) * np + kh : nullptr;
    const double* bp = (!inside && 64 * J + lc < mE) ? Et + (size_t)(64 * J + lc) * np + kq : nullptr;
    d4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    double2 ra[4], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; q++) ra[q] = ap ? *reinterpret_cast<const double2*>(ap + k0 + 2 * q) : double2{0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 2; q++) rb[q] = bp ? *reinterpret_cast<const double2*>(bp + k0 + 2 * q) : double2{0.0, 0.0};
    };
    fetch(0);
    const double* Bsrc = inside ? As + 64 * (J & 1) : Bs;
    const int PBs = inside ? PA : PB;
    for (int k0 = 0; k0 < np; k0 += 16) {
        __syncthreads();      // the panels of the last step are consumed
#pragma unroll
        for (int q = 0; q < 4; q++) { As[(kh + 2 * q) * PA + lr] = ra[q].x; As[(kh + 2 * q + 1) * PA + lr] = ra[q].y; }
        if (!inside) {
#pragma unroll
            for (int q = 0; q < 2; q++) { Bs[(kq + 2 * q) * PB + lc] = rb[q].x; Bs[(kq + 2 * q + 1) * PB + lc] = rb[q].y; }
        }
        __syncthreads();
        if (k0 + 16 < np) fetch(k0 + 16);
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const int kk = 4 * k4 + (lane >> 4), il = lane & 15;
            double av[4], bv[2];
#pragma unroll
            for (int i = 0; i < 4; i++) av[i] = As[kk * PA + 64 * wy + 16 * i + il];
#pragma unroll
            for (int j = 0; j < 2; j++) bv[j] = Bsrc[kk * PBs + 32 * wx + 16 * j + il];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // block (i, j), accumulator register q: row 16 i + (lane >> 4) + 4 q, column 16 j + (lane & 15)   (C/D layout of the f64 MFMA)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int gi = 128 * I + 64 * wy + 16 * i + (lane >> 4) + 4 * q;
            if (gi >= ld) continue;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int gj = 64 * J + 32 * wx + 16 * j + (lane & 15);
                if (gj < ld) M[(size_t)gi * ld + gj] = acc[i][j][q];
            }
        }
}

// ---- the homotopy megakernel: one persistent workgroup per LCQP -----------------------------------
#ifndef LCQP_MINWAVES
#define LCQP_MINWAVES 4      // waves per SIMD the register allocation is held to (4 workgroups per CU)
#endif
// A second build of the two persistent kernels for batches of at most three workgroups per CU, np <= 512 (lcqp_nch.hip with -DLCQP_TU_FEW: LCQP_VARIANT 1,
// LCQP_MINWAVES 2): 256 registers instead of 128 -- 8.0 -> 7.3 ms for one LCQP alone, 11.3 -> 10.3 at B = 16, 13.5 -> 12.3 at B = 128,
// 14.4 -> 13.2 at B = 256, 21.7 -> 20.7 at B = 768 (two workgroups per CU resident, the rest behind them), the same bits; from B = 896 on four
// resident workgroups per CU win (28.6 against 31.9 ms at B = 1024).  profiles/round6/latency/.  (Eight rows in flight per wave instead of
// four would take another 1 - 4 % off the small batches, but the compiler then contracts the dot products of the sweeps into other fused
// multiply-adds: 1e-16 of difference and other iterate counts -- an instance must not depend on the size of the batch it is solved in.)
// V only gives the instantiations of the two builds different names.
#ifndef LCQP_VARIANT
#define LCQP_VARIANT 0
#endif
// LR: the row state of the subsolver lives in LDS (np <= 256 and at most LDS_ROWS_MAX rows of E; lcqp_wg.hpp)
template <int NCH, bool LR, int V>
__global__ __launch_bounds__(WG, LCQP_MINWAVES) void k_lcqp_run(DevBatch db)
{
    LCQP_LDS_N(NCH)
    Ctx<NCH> c = make_ctx<NCH>(db, blockIdx.x, lds);
    lcqp_run<NCH, true, LR>(c);      // with the dependent-row rules: since the factor is updated instead of rebuilt they cost nothing here
                                     // (A/B 73.0 vs 72.0 ms, profiles/round2), so the batched loop and the per-QP path are ONE algorithm
}

// ---- one QP per workgroup with the SubsolverBase semantics ----------------------------------------
template <int NCH, int V>
__global__ __launch_bounds__(WG, LCQP_MINWAVES) void k_qp_solve(DevBatch db, int initial)
{
    LCQP_LDS_N(NCH)
    Ctx<NCH> c = make_ctx<NCH>(db, blockIdx.x, lds);
    const double* y0 = (initial && c.info->hasY0) ? db.y0 + (size_t)c.b * db.nd : nullptr;
    int iters = 0;
    const int ef = qp_solve<NCH, true, true>(c, initial, c.V(V_GK), y0, &iters);   // single-QP path: dependent-row rules and rho adaptation
    if (ef == 0) qp_export<NCH>(c, db.xout + (size_t)c.b * db.n, db.n, db.yout + (size_t)c.b * db.nd);
    if (threadIdx.x == 0) {
        lcqp_stats_t s;
        memset(&s, 0, sizeof(s));
        s.subproblemIter = iters;
        s.qpSolverExitFlag = ef;
        s.returnValue = ef ? LCQP_SUBPROBLEM_SOLVER_ERROR : 0;
        s.admmIter = c.cAdmm; s.trials = c.cTrials; s.factorizations = c.cFact; s.corrections = c.cCorr; s.qpSolves = 1; s.reserved = c.cSweeps;
        db.stats[c.b] = s;
    }
}

// ---- synthetic instances directly in HBM (include/lcqp_synth.h) ---------------------
template <int NCH>
__global__ __launch_bounds__(WG) void k_synth_fill(DevBatch db, uint64_t seed0, uint64_t first)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x, t = threadIdx.x;
    Ctx<NCH> c = make_ctx<NCH>(db, b, lds);
    const int n = db.n, nC = db.nC, nComp = db.nComp, mA = db.mA;
    const uint64_t st = lcqp_synth_state(seed0, first + (uint64_t)b);
    double* Mm = c.F1;   // scratch: M, consumed by k_synth_Q
    for (int e = t; e < np * np; e += WG) {
        const int i = e / np, j = e - i * np;
        Mm[e] = (i < n && j < n) ? lcqp_synth_M(st, n, i, j) : 0.0;
    }
    double* xs = c.V(V_TMP);
    for (int i = t; i < np; i += WG) {
        c.V(V_G)[i] = (i < n) ? lcqp_synth_g(st, n, nC, nComp, i) : 0.0;
        xs[i] = (i < n) ? lcqp_synth_xstar(st, n, nC, nComp, i) : 0.0;
        c.V(V_X0)[i] = 0.0;
        c.V(V_LB)[i] = -INFINITY;
        c.V(V_UB)[i] = INFINITY;
    }
    const double sn = sqrt((double)n);
    for (int r = 0; r < mA; r++) {
        double* row = c.E + (size_t)r * np;
        for (int j = t; j < np; j += WG) {
            double v = 0.0;
            if (j < n) {
                if (r < nC) v = lcqp_synth_Araw(st, n, nC, nComp, r, j) / sn;
                else if (r < nC + nComp) v = (j == r - nC) ? 1.0 : 0.0;
                else v = (j == nComp + (r - nC - nComp)) ? 1.0 : 0.0;
            }
            row[j] = v;
        }
    }
    __syncthreads();
    double *l = c.M(M_L), *u = c.M(M_U);
    for (int r = t; r < mA; r += WG) {
        if (r < nC) {
            // A x* summed left to right with separately rounded products, as the host generator does
            const double* row = c.E + (size_t)r * np;
            double ax = 0.0;
            for (int j = 0; j < n; j++) {
#pragma clang fp contract(off)
                const double pr = row[j] * xs[j];
                ax = ax + pr;
            }
            l[r] = ax - lcqp_synth_slo(st, n, nC, nComp, r);
            u[r] = ax + lcqp_synth_shi(st, n, nC, nComp, r);
        } else { l[r] = 0.0; u[r] = INFINITY; }
    }
    if (t == 0) { c.info->nfin = 0; c.info->hasY0 = 0; c.info->isSetup = 0; }
}

// Q = M'M/n + I in exactly the arithmetic of the host generator (oracle: orc_synth_generate): every element is the sum over
// k ascending of separately rounded products (no FMA contraction), then one division and one addition -- so the instances
// generated in HBM are bit-identical to the ones the CPU oracle generates.  One 64x64 tile of the lower triangle per
// workgroup, a 4x4 block per thread; the generator runs before the timed region of bench.py.
template <int NCH>
__global__ __launch_bounds__(WG) void k_synth_Q(DevBatch db)
{
    constexpr int np = 128 * NCH;
    const int ntile = db.nblk * (db.nblk + 1) / 2;
    const int b = blockIdx.x / ntile, tIdx = blockIdx.x % ntile, t = threadIdx.x;
    int I, J;
    tri_tile(tIdx, I, J);
    const double* Mm = db.F1 + (size_t)b * np * np;
    double* Q = db.Q + (size_t)b * np * np;
    const int gi0 = 64 * I + 4 * (t >> 4), gj0 = 64 * J + 4 * (t & 15);
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    for (int k = 0; k < db.n; k++) {
#pragma clang fp contract(off)      // separately rounded product and sum (plain operators: the pragma does not reach into __dmul_rn)
        const double* mk = Mm + (size_t)k * np;
        double mi[4], mj[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { mi[i] = mk[gi0 + i]; mj[i] = mk[gj0 + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) { const double pr = mi[i] * mj[j]; acc[i][j] = acc[i][j] + pr; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gi = gi0 + i, gj = gj0 + j;
            double v = 0.0;
            if (gi < db.n && gj < db.n) v = acc[i][j] / (double)db.n + (gi == gj ? 1.0 : 0.0);   // a quotient plus 0 or 1: nothing to fuse
            Q[(size_t)gi * np + gj] = v;
            Q[(size_t)gj * np + gi] = v;
        }
}

// ---- building-block kernels for tests / micro-benchmarks ------------------------------------------
template <int NCH>
__global__ __launch_bounds__(WG) void k_util_symv(int n, double alpha, const double* A, const double* bv, const double* cv, double* d)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x;
    double* out = d + (size_t)b * np;
    wg_symv<NCH>(A + (size_t)b * np * np, nullptr, n, bv + (size_t)b * np, nullptr, out, nullptr, nullptr, nullptr, lds);
    for (int i = threadIdx.x; i < np; i += WG) out[i] = alpha * out[i] + cv[(size_t)b * np + i];
}

template <int NCH>
__global__ __launch_bounds__(WG) void k_util_rows(int m, const double* A, const double* x, double* dots, const double* coef, double* outT)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x;
    double* o = outT ? outT + (size_t)b * np : nullptr;
    wg_rows<NCH>(A + (size_t)b * m * np, nullptr, m, x ? x + (size_t)b * np : nullptr, dots ? dots + (size_t)b * m : nullptr,
                 coef ? coef + (size_t)b * m : nullptr, lds, [&](int i, double s) { if (o) o[i] = s; });
}

// the row sweep through a row list with row-indexed scalars (wg_rows<NCH, true>: stage 1 / stage 2 of the subsolver's trials), on its own
template <int NCH>
__global__ __launch_bounds__(WG) void k_util_rows_list(int m, int nlist, const double* A, const int* list, const double* x, double* dots, const double* coef, double* outT)
{
    LCQP_LDS_N(NCH)
    constexpr int np = 128 * NCH;
    const int b = blockIdx.x;
    double* o = outT ? outT + (size_t)b * np : nullptr;
    wg_rows<NCH, true>(A + (size_t)b * m * np, list + (size_t)b * nlist, nlist, x ? x + (size_t)b * np : nullptr, dots ? dots + (size_t)b * m : nullptr,
                       coef ? coef + (size_t)b * m : nullptr, lds, [&](int i, double s) { if (o) o[i] = s; });
}

// ---- launcher of this translation unit's instantiation (declared in lcqp_launch.hpp) -----------------------------------------
template <int NCH>
static void launch_run(int grid, hipStream_t s, const LaunchArgs& a)
{
#ifndef LCQP_NO_LDS_ROWS      // experiment switch: the row state in global memory for every size
    if constexpr (NCH <= 2) {
        // the row state sits behind the routines' scratch (arena[0, LDS_ROWS_OFF)): the sweeps need 6 np doubles there, the triangular solves
        // 5 np, the widest pass over the inverse factor 4 capS, the rotations 3 capS
        static_assert(6 * 128 * NCH <= LDS_ROWS_OFF, "k_lcqp_run<NCH, true>: the sweeps' scratch must end below the row state");
        if (a.db.mEcap <= LDS_ROWS_MAX && 4 * a.db.capS <= LDS_ROWS_OFF) { hipLaunchKernelGGL((k_lcqp_run<NCH, true, LCQP_VARIANT>), dim3(grid), dim3(WG), 0, s, a.db); return; }
    }
#endif
    hipLaunchKernelGGL((k_lcqp_run<NCH, false, LCQP_VARIANT>), dim3(grid), dim3(WG), 0, s, a.db);
}

#ifdef LCQP_TU_FEW
// the translation unit of the second build holds the two persistent kernels only
template <int NCH>
static void launch_impl(int kid, int grid, hipStream_t s, const LaunchArgs& a)
{
    if (kid == ID_k_lcqp_run) launch_run<NCH>(grid, s, a);
    else if (kid == ID_k_qp_solve) hipLaunchKernelGGL((k_qp_solve<NCH, LCQP_VARIANT>), dim3(grid), dim3(WG), 0, s, a.db, a.initial);
}
#else
template <int NCH>
static void launch_impl(int kid, int grid, hipStream_t s, const LaunchArgs& a)
{
    switch (kid) {
        case ID_k_prepare:    hipLaunchKernelGGL((k_prepare<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_build_C:    hipLaunchKernelGGL((k_build_C<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_compress_C: hipLaunchKernelGGL((k_compress_C<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_factor:     hipLaunchKernelGGL((k_factor<NCH, 1>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_factor_full: hipLaunchKernelGGL((k_factor<NCH, NCH <= 2 ? 4 : 1>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_trsm:       hipLaunchKernelGGL((k_trsm<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_trsm_streamed: hipLaunchKernelGGL((k_trsm_streamed<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_build_M:    hipLaunchKernelGGL((k_build_M<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_lcqp_run:   launch_run<NCH>(grid, s, a); break;
        case ID_k_qp_solve:   hipLaunchKernelGGL((k_qp_solve<NCH, LCQP_VARIANT>), dim3(grid), dim3(WG), 0, s, a.db, a.initial); break;
        case ID_k_synth_fill: hipLaunchKernelGGL((k_synth_fill<NCH>), dim3(grid), dim3(WG), 0, s, a.db, a.seed0, a.first); break;
        case ID_k_synth_Q:    hipLaunchKernelGGL((k_synth_Q<NCH>), dim3(grid), dim3(WG), 0, s, a.db); break;
        case ID_k_util_symv:  hipLaunchKernelGGL((k_util_symv<NCH>), dim3(grid), dim3(WG), 0, s, a.n, a.alpha, a.A, a.b, a.c, a.d); break;
        case ID_k_util_rows:  hipLaunchKernelGGL((k_util_rows<NCH>), dim3(grid), dim3(WG), 0, s, a.m, a.A, a.x, a.dots, a.coef, a.outT); break;
        case ID_k_util_rows_list: hipLaunchKernelGGL((k_util_rows_list<NCH>), dim3(grid), dim3(WG), 0, s, a.m, a.n, a.A, a.list, a.x, a.dots, a.coef, a.outT); break;
        default: break;
    }
}
#endif
