// lcqp_sparse.hip -- the SPARSE arm of the hot path on gfx950: B independent LCQPs that share one sparsity pattern, behind
// lcqp_hip_sparse_* (include/lcqp_hip.h).  G lanes of a wavefront per instance (G = 8, 16, 32 or 64: the smallest power of two
// above the half bandwidth of the KKT band), 64 / G instances per wavefront, one wavefront per workgroup (k_sparse_sched).
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
// (all instances of a batch share it): LDL' with a sliding G x G window in registers (G <= 16) or LDS (half bandwidth w <= G - 1 <= 63), triangular
// solves that keep the G pending rows of an instance in its G lanes (axpy form both ways, no reductions in the chain; the finished
// entry is broadcast inside the lane group by DPP).  A group of lanes behaves like a small workgroup of its own: every branch
// condition is uniform inside a group, groups of one wavefront diverge through the execution mask, there are no workgroup barriers.
// Patterns whose KKT band is wider (e.g. the arrow-shaped circle example) are refused here; the host layer runs them on the
// dense kernels behind the same OSQP_SPARSE surface.
#include "lcqp_wg.hpp"
#include "../../include/lcqp_hip.h"
#include "lcqp_sparse_general.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <queue>
#include <string>
#include <vector>

using namespace lcqp;

namespace {

constexpr int SP_WMAX = 63;
constexpr int SP_KBMAX = 16;   // border nodes of the bordered band (rows / variables too dense for a band)
constexpr int WGS = 64;      // one wavefront per workgroup; 64 / G instances in it
#ifndef SP_WAVES_PER_SIMD
#define SP_WAVES_PER_SIMD 2  // register budget of k_sparse_sched: 512 / SP_WAVES_PER_SIMD per lane
#endif
#ifndef SP_SWEEP_RING
#define SP_SWEEP_RING 8      // coefficient chunks (8 steps each) of a band sweep in flight at G = 8: seven chunks = 56 steps ahead of use, 8 800 / 5 600 clocks of
                             // the forward / backward sweep (with 4: 3 800 / 2 400, less than a round trip to memory on a busy machine -- the sweeps of a full
                             // machine took 1.7 x the time of a lone instance's; profiles/round5/sparse_sweep_ring8_ab.log: +5.5 % at B = 1024, +3.6 % at 4096,
                             // -1.6 % at 16 384, +-0 at 65 536; round 4 had measured it at 65 536 only)
#endif
enum { NV_G, NV_GTIL, NV_GPHI, NV_XK, NV_PK, NV_XNEW, NV_GK, NV_QX, NV_CX, NV_QP, NV_CP, NV_TMP, NV_XQ, NV_XA, NV_XT, NV_R1,
       NV_X0, NV_NUM };
enum { MV_L, MV_U, MV_RHOV, MV_YQ, MV_YA, MV_ZA, MV_YT, MV_EX, MV_YK, MV_Y0, MV_LX, MV_LX2, MV_NUM };
enum { MI_ST, MI_STT, MI_STF, MI_NEW, MI_NUM };

struct SpInfo {
    int haveSolution, stfValid, hasY0, bigReg;     // bigReg: this instance needs the safe regularisation of the polish (a Hessian that is only semidefinite)
    double scale, sigma, delta, delta2, phiConst;
    double deltaS, delta2S;                        // the light level tried first (sp_polish)
    double e1max;                                  // largest row 1-norm of E: with |x|_inf the scale of the rounding of a computed E_r x (the active-row test of the polish)
    double hist[64];
    double bytes;        // algorithmic bytes counted by the kernel
    double prof[8];      // -DLCQP_PROFILE: clock ticks per phase (SP_* below)
};
enum { SP_PRODUCTS, SP_ASSEMBLE, SP_FACTOR, SP_FORWARD, SP_BACKWARD, SP_VECTORS, SP_LCQP, SP_RHS, SP_NPHASE };

// ---- the homotopy as a phase machine (round 4) ----------------------------------------------------------------------------------------
// Round 3 ran the whole homotopy of an instance inside one persistent lane group: the 64 / G instances of a wavefront moved in lock step, an
// instance was active in 74 % of its wavefront's trials and refactorised in 26 % of them while its wavefront did in 83 %.  Now an instance is
// a record in memory (SpState + its vectors) that moves through QUEUES, one per phase; a wavefront pops up to 64 / G instances that are in
// the SAME phase, runs that phase for them, and pushes each to the queue of its next phase (k_sparse_sched).  A wavefront therefore only
// ever holds instances doing the same thing; nobody waits for a neighbour's factorisation or for the slowest polish of eight.
enum { PH_START, PH_ROUND, PH_TRIAL, PH_FACTOR, PH_CORRECT, PH_QPEND, PH_NUM };
enum { BY_ASSEMBLE, BY_FACTOR_LDS, BY_FACTOR, BY_SOLVE, BY_BORDER_PREPARE, BY_BORDER_SOLVE, BY_EX, BY_SWEEP, BY_START, BY_E, BY_NUM };
struct SpState {
    // LCQProblem::runSolver (src/LCQProblem.cpp:444-560)
    int initial, histLen, algoStat, totalIter, rc, qpIter;
    double alphak, rho, gmaxNext;
    unsigned long long perturbCounter;
    lcqp_stats_t st;
    // the subsolver call (oracle: sqp_solve)
    int round, n_admm, use_stored, backup_pending, admm_ready, trials0, admm0;
    // the polish (oracle: sqp_polish)
    int trial, reuse, fact_valid, borderTodo, nrefine;      // nrefine: refinement corrections taken for the active rows alone
    double gs, ytol, dpUsed, d2Used, xinf;                  // xinf: |x|_inf behind the last correction
    // work counters
    int cAdmm, cTrials, cFact, cCorr, cSweeps;
    double bytes;
};
// Queues: the batch is cut into pools of `poolSize` consecutive instances (a power of two; the byte offset of an instance inside its pool fits
// 32 bits for every per-instance array: the saddr + 32-bit offset addressing of SpCtx::arr); wavefront w serves pool w % nPools.  Per pool
// and phase a ring of poolSize entries (an instance is in at most one queue) and three counters: tail (next position to write), head (next
// position to read), count (entries published); every ring slot carries a SEQUENCE number beside the instance id (a bounded multi-producer /
// multi-consumer queue after Vyukov; sequence and id share one 64-bit word, written by one store): slot p & mask is free for position p when
// its sequence is p, holds position p's entry when it is p + 1, and is handed on to position p + poolSize by its consumer -- a producer that laps the ring onto a slot whose entry has been claimed but not
// read yet waits instead of overwriting it.  ctl[pool][PH_NUM] = instances of the pool not finished yet.
constexpr int QCTL = 4;      // ints per (pool, phase): tail, head, count, pad

struct EllMat { const int *eidx, *epos, *ptr, *cidx, *cmap; int rows, W, tails; };   // see g_ell

struct SpBatch {
    int B, n, m, nC, nComp, N, Np, w, ld, nnzQ, nnzE, G;  // Np: N rounded up to a multiple of 64 (padding rows: zero coefficients)
    int hasLbL, hasLbR;
    lcqp_options_t opt;
    const int *Qp, *Qi, *Ep, *Ei, *ETp, *ETi, *ETmap, *iperm, *bandQ, *bandE;
    const int *bsrc, *pnode, *qdiag, *Erow;   // band entry -> value it comes from (sp_factor_reg), node of a band position, Q_ii, row of an E entry
    const int *bgate, *bdiag;                 // band entry -> the row of E whose membership in the working set gates it (-1: none); band position -> its diagonal (sp_factor_reg)
    EllMat ellQ, ellE, ellT; // rows of Q, rows of E, columns of E in ELL slabs
    double *Qx, *Ex;         // [B][nnzQ], [B][nnzE] (CSR order)
    double *Kb;              // [B][N*ld] assembled band rows (input of a factorisation): Kb[i*ld + k] = K[i][i-w+k]
    double *KaF, *KaD;       // ADMM KKT factor in the folded layout of band_sweep [B][Np*G], 1/D [B][Np]
    double *KpF, *KpD;       // polish KKT factor
    double* K0;              // [Np][G] per instance: the band rows of [Q, E'; E, .] with every row of E in, diagonal slot Q_ii (variables) -- what sp_factor_reg streams
    int bitWords;            // 32-bit words of a working set's bit set in LDS (sp_ph_factor), 0: it does not fit, the flags are read from memory
    double *nv, *mv, *Nv;    // [B][NV_NUM][n], [B][MV_NUM][m], [B][2][Np]
    // Bordered band (round 3): the last kb positions of the ordering are border nodes -- rows or variables too dense for any band (the
    // coupling constraint and the two shared variables of examples/OptimizeOnCircle.cpp).  K = [Bd U'; U C]: the band engine factorises
    // Bd with the border positions as isolated unit pivots; the border is carried by W = U inv(Bd) (kb band solves per factorisation) and
    // the Schur complement S = C - W U' (kb x kb, dense LDL').  U: per border node the entries it shares with band nodes (Upos: band
    // position, Usrc: entry of Q (k < nnzQ) or of E (nnzQ + k, CSR order), Ugate: the row of E whose membership in the working set gates
    // the entry, -1 none); C: the entries among border nodes, lower triangle (Cb2: the other border node).
    int kb, nU, nCb;
    int lightOK;     // the ordering puts every row behind one of its variables and every Hessian of the batch is safely definite: the polish tries its light regularisation first
    const int *bnode, *Uptr, *Upos, *Usrc, *Ugate, *Cptr, *Cb2, *Csrc, *Cgate;
    double *bW, *bUv, *bS;   // [B][2][kb][Np] W rows, [B][2][nU] gated values of U, [B][2][kb][kb] factor of S   (index 0: polish, 1: ADMM)
    double *lbL, *lbR;       // [B][nComp]
    int* mi;                 // [B][MI_NUM][m]
    SpInfo* info;
    lcqp_stats_t* stats;
    double *xout, *yout;     // [B][n], [B][m]
    // per-iterate tracking (options.storeSteps, src/LCQProblem.cpp:1365-1378), as on the dense path: [B][traceCap][8] = (|statk|inf, phi, rho,
    // alphak, obj, merit, |pk|inf, QP iterations), [B][traceCap][n] = xk, traceLen[B]; traceCap == 0: not allocated
    double *traceS, *traceX;
    int* traceLen;
    int traceCap;
    // phase machine
    SpState* state;          // [B]
    unsigned long long* qring;  // [nPools][PH_NUM][poolSize] ring slots: (sequence number << 32) | instance id, one 64-bit word so that a slot changes hands in one store
    int* qctl;                  // [nPools][PH_NUM + 1][QCTL]
    int poolSize, nPools;
    int wideDiv;                // SIMDs of the device per pool (sp_launch): unfinished instances of the pool / wideDiv = instances of a streaming step
    // General sparse LDL' (round 6; lcqp_sparse_general.hpp): patterns that are neither banded nor bordered -- multifrontal over a nested-
    // dissection tree with dense fronts, one wavefront per instance (G = 64).  general != 0: KaF / KpF hold the panels of the fronts
    // (gLsize doubles per instance instead of Np * G), KaD / KpD 1 / D per position as for the band; kb = 0, lightOK = 0.
    int general, gnF, gMaxFront;
    unsigned gLsize, gStackSize;
    const int *gPiv0, *gNp, *gNb, *gRowPtr, *gRows, *gChildPtr, *gChild, *gRel, *gAsmPtr, *gAsmSrc, *gAsmGate, *gAsmPos;
    const unsigned *gLoff, *gCBoff;
    const int *gMeta, *gChildInfo;   // [gnF][GEN_META] np, nb, piv0, rowPtr, asmPtr, asmEnd, childPtr, childEnd, Loff, CBoff: one load per front; [children][4] nb, CBoff, rowPtr of the child
    double *gStack, *gFront;     // [B][gStackSize] update blocks of the fronts, [B][gMaxFront^2] a front too large for LDS
    size_t kfStride;             // doubles per instance of KaF / KpF
    // algorithmic bytes of one event of each kind (filled by the host: formed in the kernel they are loop invariants the compiler keeps in
    // registers across every phase)
    double by[BY_NUM];
    unsigned long long* qprof;   // [PH_NUM + 1][3] (-DLCQP_SCHED_PROFILE): clock ticks, wavefront steps, instances served per phase; row PH_NUM: ticks / polls without work
};

// ---- addressing: uniform base pointer + 32-bit lane offset -------------------------------------------------------------------------
// Every per-instance array is reached as (base of the wave's first instance: uniform, SGPRs) + (byte offset of the lane's instance
// inside the wave's block + element offset: 32 bits, one VGPR) -- the global_load saddr form.  Per-lane 64-bit pointers would cost two
// VGPRs for each of the ~40 vectors of an instance (the compiler hoists them) and 64-bit VALU address arithmetic at every access.
typedef double dv2 __attribute__((ext_vector_type(2)));
template <class T> struct GRef {
    T* base; unsigned boff;
    __device__ __forceinline__ operator T() const { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)boff); }
    __device__ __forceinline__ T operator=(T v) const { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)boff) = v; return v; }
    __device__ __forceinline__ T operator=(const GRef& o) const { return *this = (T)o; }
    __device__ __forceinline__ T operator+=(T v) const { return *this = (T)(*this) + v; }
    __device__ __forceinline__ T operator-=(T v) const { return *this = (T)(*this) - v; }
};
template <class T> struct GP {
    T* base; unsigned off;                       // uniform base; byte offset of this lane's data
    __device__ __forceinline__ GRef<T> operator[](int i) const { return GRef<T>{base, off + (unsigned)i * (unsigned)sizeof(T)}; }
    __device__ __forceinline__ T ld(int i) const { return (T)(*this)[i]; }
    __device__ __forceinline__ GP<T> operator+(int i) const { return GP<T>{base, off + (unsigned)i * (unsigned)sizeof(T)}; }
    __device__ __forceinline__ dv2 ld2(int i) const { return *reinterpret_cast<const dv2*>(reinterpret_cast<const char*>(base) + (size_t)(off + (unsigned)i * 8u)); }
};
typedef GP<double> GD;
typedef GP<int> GI;
// the value type behind what a load lambda returns (a lambda that returns x[i] returns the reference proxy, not the value)
template <class X> struct val_of { typedef X type; };
template <class T> struct val_of<GRef<T>> { typedef T type; };

extern __shared__ double sp_dyn_lds[];      // G <= 16: the working sets' bit sets (sp_ph_factor); G > 16: the windows of sp_factor_lds

template <int G>
struct SpCtx {
    const SpBatch* db;
    int b, gl;               // instance, lane inside the group
    unsigned gi;             // this instance's index relative to w0
    int w0;                  // first instance of the block of instances the wave addresses (uniform): its own 64 / G instances (k_sparse_setup) or its pool (k_sparse_sched)
    SpInfo* info;
    double* win;             // LDS of this group: G x G window + 16 staged rows
    int cAdmm, cTrials, cFact, cCorr, cSweeps;
    double bytes;
#ifdef LCQP_PROFILE
    unsigned long long tprev, prof[SP_NPHASE];
#endif
    // per-instance arrays: block of the wave's first instance (uniform) + this instance's offset inside it
    template <class T> __device__ __forceinline__ GP<T> arr(T* p, size_t perInst, unsigned extra = 0) const
    { return GP<T>{p + ((size_t)w0 * perInst + extra), gi * (unsigned)perInst * (unsigned)sizeof(T)}; }    // vectors differ in the (SGPR) base only
    __device__ __forceinline__ GD V(int k) const { return arr(db->nv, (size_t)NV_NUM * db->n, (unsigned)k * db->n); }
    __device__ __forceinline__ GD M(int k) const { return arr(db->mv, (size_t)MV_NUM * db->m, (unsigned)k * db->m); }
    __device__ __forceinline__ GI I(int k) const { return arr(db->mi, (size_t)MI_NUM * db->m, (unsigned)k * db->m); }
    __device__ __forceinline__ GD Qx() const { return arr(db->Qx, db->nnzQ); }
    __device__ __forceinline__ GD Ex() const { return arr(db->Ex, db->nnzE); }
    __device__ __forceinline__ GD Nv() const { return arr(db->Nv, (size_t)2 * db->Np); }
    __device__ __forceinline__ GD Kb() const { return arr(db->Kb, (size_t)db->N * db->ld); }
    __device__ __forceinline__ GD KF(bool admm) const { return arr(admm ? db->KaF : db->KpF, db->kfStride); }
    __device__ __forceinline__ GD GStack() const { return arr(db->gStack, db->gStackSize); }
    __device__ __forceinline__ GD GFront() const { return arr(db->gFront, (size_t)db->gMaxFront * db->gMaxFront); }
    __device__ __forceinline__ GD KD(bool admm) const { return arr(admm ? db->KaD : db->KpD, db->Np); }
    __device__ __forceinline__ GD K0() const { return arr(db->K0, (size_t)db->Np * G); }
    __device__ __forceinline__ GD BW(bool admm) const { return arr(db->bW, (size_t)2 * db->kb * db->Np, (unsigned)(admm ? db->kb * db->Np : 0)); }
    __device__ __forceinline__ GD BUv(bool admm) const { return arr(db->bUv, (size_t)2 * db->nU, (unsigned)(admm ? db->nU : 0)); }
    __device__ __forceinline__ GD BS(bool admm) const { return arr(db->bS, (size_t)2 * db->kb * db->kb, (unsigned)(admm ? db->kb * db->kb : 0)); }
};

#ifdef LCQP_PROFILE
#define SPROF(c, P) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (c).prof[P] += t_ - (c).tprev; (c).tprev = t_; } while (0)
#else
#define SPROF(c, P) do { } while (0)
#endif

// Everything a lane computes from its lane number and uniform values is invariant in every loop of the kernel, and the compiler
// hoists it all to the top (hundreds of 64-bit addresses, spilled to scratch at once).  An empty volatile asm cannot be hoisted:
// what is derived from the laundered lane number stays inside the routine that uses it.
__device__ __forceinline__ int here(int lane) { asm volatile("" : "+v"(lane)); return lane; }

// ---- lane-group collectives (every lane of the group takes part; results are uniform inside the group) ------------------------
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// DPP controls: 0xB1 = quad_perm[1,0,3,2], 0x4E = quad_perm[2,3,0,1], 0x141 = row_half_mirror (lane j <- 7 - j of its 8), 0x140 = row_mirror
template <int G> __device__ __forceinline__ double g_sum(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    if (G >= 16) v += dpp_d<0x140>(v);
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
// maximum that keeps a NaN (fmax drops it): a residual with a NaN in it must not pass an acceptance test
__device__ __forceinline__ double nmax(double a, double b) { return (b > a || b != b) ? b : a; }
template <int G> __device__ __forceinline__ double g_max(double v)
{
    v = nmax(v, dpp_d<0xB1>(v));
    v = nmax(v, dpp_d<0x4E>(v));
    v = nmax(v, dpp_d<0x141>(v));
    if (G >= 16) v = nmax(v, dpp_d<0x140>(v));
    if (G >= 32) v = nmax(v, __shfl_xor(v, 16, 64));
    if (G >= 64) v = nmax(v, __shfl_xor(v, 32, 64));
    return v;
}
template <int G> __device__ __forceinline__ int g_sum_i(int v)
{
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    if (G >= 16) v += dpp_i<0x140>(v);
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int G> __device__ __forceinline__ bool g_any(int v)
{
    const unsigned long long mk = __ballot(v != 0);
    if (G == 64) return mk != 0ull;
    const int sh = threadIdx.x & ~(G - 1);
    return ((mk >> sh) & ((1ull << (G & 63)) - 1ull)) != 0ull;
}
// value of lane k of the group (k is a compile-time constant after unrolling)
template <int G> __device__ __forceinline__ double g_bcast(double v, int k)
{
    if (G == 64) return wave_bcast(v, k);
    if (G == 8) {
        double t;
        switch (k & 3) {
            case 0: t = dpp_d<0x00>(v); break;
            case 1: t = dpp_d<0x55>(v); break;
            case 2: t = dpp_d<0xAA>(v); break;
            default: t = dpp_d<0xFF>(v); break;
        }
        const double u = dpp_d<0x141>(t);
        return ((((int)threadIdx.x >> 2) & 1) == (k >> 2)) ? t : u;
    }
    return __shfl(v, k, G);
}
// LDS traffic of one wave is in order; this keeps the compiler from moving LDS accesses across the point
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// global memory written by one lane of the group and read by another: wait for the stores (no barrier: the group is inside one wave)
__device__ __forceinline__ void g_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// ---- loops over the entries of a vector, G lanes per instance ------------------------------------------------------------------
// One or two wavefronts per SIMD cannot hide a load behind other waves, so every loop is software-pipelined: the loads of the next
// tile of U*G elements (load(i) returns them by value) are issued before the stores of the current tile (store(i, v)).  Legal for
// element-wise loops only: store(i, .) must not write what load(j) reads for j != i.
template <int G, int U, class L, class S>
__device__ __forceinline__ void g_map(int n, int gl, L load, S store)
{
    using T = typename val_of<decltype(load(0))>::type;
    T v[U];
    gl = here(gl);
    // the first trip only loads (tile 0): keeping these loads inside the loop keeps their addresses from being hoisted to the top
    // of the kernel as loop invariants of the outer loops (one 64-bit address per vector and tile element, hundreds of registers)
    for (int i0 = gl - U * G; i0 < n; i0 += U * G) {
        T w[U];
        const int i1 = i0 + U * G;
        if (i1 - gl < n) {
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i1 + u * G; w[u] = load(i < n ? i : 0); }
        }
        if (i0 >= 0) {
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i0 + u * G; if (i < n) store(i, v[u]); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = w[u];
    }
}
struct D2 { double a, b; };
struct D3 { double a, b, c; };
struct D4 { double a, b, c, d; };
struct ID { int i; double a; };

// ---- sparse products: lane per row / per column of the pattern, in ELL form ----------------------------------------------------------
// The pattern is shared by the batch, so the host lays it out once as ELL slabs: eidx[q * rows + i] = index of the q-th entry of row
// i into the gathered vector, epos[q * rows + i] = its position in the instance's value array (-1: no such entry), q < W (4 or 8);
// longer rows finish in a scalar tail over the CSR/CSC arrays.  Per tile of U*G rows: the values and the gathered vector entries of
// the tile and the indices of the NEXT tile are in flight together.  xv(j) returns a D2 (two vectors share one pass over the
// matrix); pre(i) loads what the consumer needs beside the sums; out(i, s0, s1, pre) consumes.

template <int G, int U, int W, bool MAP, class Xv, class Pre, class Out>
__device__ __forceinline__ void g_ell(const EllMat& E, int gl, GD vals, Xv xv, Pre pre, Out out)
{
    const int rows = E.rows;
    gl = here(gl);
    // indices of a tile: the column slab and either the position slab (a value map: E^T over E's values) or the row's two pointers --
    // without a map the values of a row lie in row order, position of entry q = ptr[i] + q: two loads per row instead of W (round 5)
    constexpr int PW = MAP ? W : 2;
    constexpr bool mapped = MAP;
    constexpr bool PREF = !MAP;   // with a map the next tile's 2 W indices per row do not fit the register budget of the G = 8 scheduler kernel
    int ci[U][W], pa[U][PW];      // pa: positions (map) / pa[u][0], pa[u][1] = ptr[i], ptr[i + 1] (no map)
    auto load_idx = [&](int t0, int (&c)[U][W], int (&p)[U][PW]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = t0 + u * G; const bool ok = i < rows;
#pragma unroll
            for (int q = 0; q < W; q++) {
                c[u][q] = ok ? E.eidx[q * rows + i] : 0;
                if (mapped) p[u][q % PW] = ok ? E.epos[q * rows + i] : -1;
            }
            if (!mapped) { p[u][0] = ok ? E.ptr[i] : 0; p[u][1] = ok ? E.ptr[i + 1] : 0; }
        }
    };
    if (gl < rows) load_idx(gl, ci, pa);
    for (int i0 = gl; i0 < rows; i0 += U * G) {
        int ps[U][W];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < W; q++)
                ps[u][q] = mapped ? pa[u][q % PW] : ((pa[u][0] + q < pa[u][1]) ? pa[u][0] + q : -1);
        double a[U][W]; D2 xs[U][W];
        typename val_of<decltype(pre(0))>::type pv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int q = 0; q < W; q++) {
                const bool ok = ps[u][q] >= 0;
                const double av = vals[ok ? ps[u][q] : 0];
                const D2 xv2 = xv(ok ? ci[u][q] : 0);
                a[u][q] = ok ? av : 0.0; xs[u][q].a = ok ? xv2.a : 0.0; xs[u][q].b = ok ? xv2.b : 0.0;
            }
            const int i = i0 + u * G;
            pv[u] = pre(i < rows ? i : 0);
        }
        // the indices of the NEXT tile go out behind this tile's values: one round trip per tile instead of two
        int cn[U][W], pn[U][PW];
        const bool more = i0 + U * G < rows;
        if (PREF && more) load_idx(i0 + U * G, cn, pn);
#pragma unroll
        for (int u = 0; u < U; u++) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int q = 0; q < W; q++) { s0 += a[u][q] * xs[u][q].a; s1 += a[u][q] * xs[u][q].b; }
            const int i = i0 + u * G;
            if (i < rows) {
                if (E.tails)
                    for (int k = E.ptr[i] + W; k < E.ptr[i + 1]; k++) {
                        const double av = vals[E.cmap ? E.cmap[k] : k]; const D2 xv2 = xv(E.cidx[k]);
                        s0 += av * xv2.a; s1 += av * xv2.b;
                    }
                out(i, s0, s1, pv[u]);
            }
        }
        if (PREF) {
            if (more) {
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int q = 0; q < W; q++) { ci[u][q] = cn[u][q]; pa[u][q % PW] = pn[u][q % PW]; }
            }
        } else if (more) load_idx(i0 + U * G, ci, pa);
    }
}
// dispatch on the slab width of the pattern (4 or 8)
template <int G, bool MAP, class Xv, class Pre, class Out>
__device__ __forceinline__ void sp_ell(const EllMat& E, int gl, GD vals, Xv xv, Pre pre, Out out)
{
    if (E.W == 4) g_ell<G, 4, 4, MAP>(E, gl, vals, xv, pre, out);
    else g_ell<G, 2, 8, MAP>(E, gl, vals, xv, pre, out);
}
struct NoPre { };

template <int G> __device__ __forceinline__ void sp_Ex(SpCtx<G>& c, GD x, GD out)
{
    SPROF(c, SP_VECTORS);
    sp_ell<G, false>(c.db->ellE, c.gl, c.Ex(), [&](int j) { return D2{x[j], 0.0}; }, [](int) { return NoPre{}; }, [&](int r, double s, double, NoPre) { out[r] = s; });
    g_sync();
    SPROF(c, SP_PRODUCTS);
}
// two vectors through one pass over Q: o0 = Q x0, o1 = Q x1
template <int G> __device__ __forceinline__ void sp_Qx2(SpCtx<G>& c, GD x0, GD x1, GD o0, GD o1)
{
    SPROF(c, SP_VECTORS);
    sp_ell<G, false>(c.db->ellQ, c.gl, c.Qx(), [&](int j) { return D2{x0[j], x1[j]}; }, [](int) { return NoPre{}; },
              [&](int i, double s0, double s1, NoPre) { o0[i] = s0; o1[i] = s1; });
    g_sync();
    SPROF(c, SP_PRODUCTS);
}
// out[i] = base(pre(i)) - (E'y)[i]   (column gather over the CSC of E); returns max |out| over the group
template <int G, class Pre, class Base> __device__ __forceinline__ double sp_ETy(SpCtx<G>& c, GD y, GD out, Pre pre, Base base)
{
    SPROF(c, SP_VECTORS);
    double mx = 0.0;
    sp_ell<G, true>(c.db->ellT, c.gl, c.Ex(), [&](int r) { return D2{y[r], 0.0}; }, pre,
              [&](int i, double s, double, typename val_of<decltype(pre(0))>::type pv) { const double v = base(pv) - s; out[i] = v; mx = nmax(mx, fabs(v)); });
    g_sync();
    SPROF(c, SP_PRODUCTS);
    return g_max<G>(mx);
}
// the residual of the stationarity condition in one pass over Q and E' (both indexed by the variable):
// r1[i] = (-g[i] - (Q x)[i]) - (E'y)[i]; returns max |r1|
// r1 = -g - Q x - E'y; returns |r1|_inf and, in `scale`, max_i(|g_i| + |Q x|_i + |E'y|_i): the residual's own rounding floor is 64 eps of that (round 6, as on
// the dense path: qp_polish in lcqp_dev.hpp)
template <int G> __device__ __forceinline__ double sp_residual(SpCtx<G>& c, GD g, GD x, GD y, GD r1, GD qx, double& scale)
{
    SPROF(c, SP_VECTORS);
    sp_ell<G, false>(c.db->ellQ, c.gl, c.Qx(), [&](int j) { return D2{x[j], 0.0}; }, [](int) { return NoPre{}; }, [&](int i, double s, double, NoPre) { qx[i] = s; });
    g_sync();
    double mx = 0.0, sc = 0.0;
    sp_ell<G, true>(c.db->ellT, c.gl, c.Ex(), [&](int r) { return D2{y[r], 0.0}; }, [&](int i) { return D2{g[i], qx[i]}; },
              [&](int i, double s, double, D2 pv) { const double v = (-pv.a - pv.b) - s; r1[i] = v; mx = nmax(mx, fabs(v)); sc = fmax(sc, fabs(pv.a) + fabs(pv.b) + fabs(s)); });
    g_sync();
    SPROF(c, SP_PRODUCTS);
    scale = g_max<G>(sc);
    return g_max<G>(mx);
}
// C v = L'(R v) + R'(L v) for two vectors: lx = E v (rows of L: nC .. nC+nComp, of R: nC+nComp ..), then a column gather with
// swapped coefficients
template <int G> __device__ __forceinline__ void sp_Cx2(SpCtx<G>& c, GD v0, GD v1, GD o0, GD o1)
{
    SPROF(c, SP_VECTORS);
    GD lx0 = c.M(MV_LX), lx1 = c.M(MV_LX2);
    sp_ell<G, false>(c.db->ellE, c.gl, c.Ex(), [&](int j) { return D2{v0[j], v1[j]}; }, [](int) { return NoPre{}; },
              [&](int r, double s0, double s1, NoPre) { lx0[r] = s0; lx1[r] = s1; });
    g_sync();
    const int nC = c.db->nC, nK = c.db->nComp;
    sp_ell<G, true>(c.db->ellT, c.gl, c.Ex(),
              [&](int r) { const int rr = r >= nC + nK ? r - nK : (r >= nC ? r + nK : -1);      // R'(L v) + L'(R v)
                           return rr >= 0 ? D2{lx0[rr], lx1[rr]} : D2{0.0, 0.0}; },
              [](int) { return NoPre{}; }, [&](int i, double s0, double s1, NoPre) { o0[i] = s0; o1[i] = s1; });
    g_sync();
    SPROF(c, SP_PRODUCTS);
}
// C v from lx = E v that somebody else has computed (the polish leaves E x of its solution), for one vector or two:
// out(i, (C v0)[i], (C v1)[i], pre(i))
template <int G, bool TWO, class Pre, class Out> __device__ __forceinline__ void sp_C_from_Ex(SpCtx<G>& c, GD lx0, GD lx1, Pre pre, Out out)
{
    SPROF(c, SP_VECTORS);
    const int nC = c.db->nC, nK = c.db->nComp;
    sp_ell<G, true>(c.db->ellT, c.gl, c.Ex(),
              [&](int r) { const int rr = r >= nC + nK ? r - nK : (r >= nC ? r + nK : -1);
                           return rr >= 0 ? D2{lx0[rr], TWO ? (double)lx1[rr] : 0.0} : D2{0.0, 0.0}; },
              pre, [&](int i, double s0, double s1, typename val_of<decltype(pre(0))>::type pv) { out(i, s0, s1, pv); });
    g_sync();
    SPROF(c, SP_PRODUCTS);
}
template <int G> __device__ __forceinline__ double sp_maxabs(const SpCtx<G>& c, GD a, int n)
{
    double s = 0.0;
    const int gl = here(c.gl);
#pragma unroll 8
    for (int i = gl; i < n; i += G) s = fmax(s, fabs(a[i]));
    return g_max<G>(s);
}

// ---- KKT assembly into band storage: Kb[i*ld + k] = K[i][i-w+k] in the ordering iperm ----------------------------------------
// variables: Q + dprim I; row r: -(use ? ddual(r) : 1) on the diagonal, its entries of E only when used
template <int G, class Dd, class Use>
__device__ __forceinline__ void sp_assemble(SpCtx<G>& c, double dprim, Dd ddual, Use use)
{
    const SpBatch& db = *c.db;
    const int t = c.gl, ld = db.ld, w = db.ld - 1, n = db.n, m = db.m;
    GD Kb = c.Kb();
    SPROF(c, SP_VECTORS);
    for (int e = t; e < db.N * ld; e += G) Kb[e] = 0.0;
    g_sync();
    for (int k = t; k < db.nnzQ; k += G) { const int o = db.bandQ[k]; if (o >= 0) Kb[o] = c.Qx()[k]; }
    for (int r = t; r < m; r += G) {
        const bool on = use(r);
        if (on) for (int k = db.Ep[r]; k < db.Ep[r + 1]; k++) { const int o = db.bandE[k]; if (o >= 0) Kb[o] = c.Ex()[k]; }      // (-1: an entry of the border)
        Kb[(size_t)db.iperm[n + r] * ld + w] = on ? -ddual(r) : -1.0;
    }
    g_sync();
    for (int i = t; i < n; i += G) Kb[(size_t)db.iperm[i] * ld + w] += dprim;
    g_sync();
    for (int b = t; b < db.kb; b += G) Kb[(size_t)(db.N - db.kb + b) * ld + w] = 1.0;      // border positions: isolated unit pivots of the band
    g_sync();
    c.bytes += db.by[BY_ASSEMBLE];
    SPROF(c, SP_ASSEMBLE);
}

// ---- band LDL' with a sliding G x G window in LDS ---------------------------------------------------------------------------------
// The elimination is a chain of N column steps with O(w^2) work each, run by the G lanes of the instance without barriers; the rows
// that enter the window are fetched 16 columns ahead.  Window slot of K[r][c]: win[(r % G) * G + (c % G)].
// in: Kb assembled band rows; out: the unit lower factor L in the folded layout the sweeps stream (band_sweep), Kd = 1/D:
//     KF[((j / G) * G + r % G) * G + j % G] = L[r][j]          (forward: columns finish in ascending order)
template <int G>
__device__ __forceinline__ void sp_factor_lds(SpCtx<G>& c, GD KF, GD Kd)
{
    constexpr int GM = G - 1;
    const int N = c.db->N, Np = c.db->Np, w = G - 1, ld = G, l = c.gl;
    GD Kb = c.Kb();
    double* win = c.win;                  // G * G
    double* stage = win + G * G;          // 16 rows x G
    for (int r = 0; r <= w && r < N; r++) {
        const int cc = r - w + l;
        if (l <= w && cc >= 0) win[(r & GM) * G + (cc & GM)] = Kb[(size_t)r * ld + l];
    }
    double pre[16];                        // rows j0 + w + 1 .. j0 + w + 16 of the NEXT block of columns, one band entry per lane and row
#pragma unroll
    for (int q = 0; q < 16; q++) { const int rn = w + 1 + q; pre[q] = (l <= w && rn < N) ? Kb[(size_t)rn * ld + l] : 0.0; }
    for (int j0 = 0; j0 < N; j0 += 16) {
#pragma unroll
        for (int q = 0; q < 16; q++) stage[q * G + l] = pre[q];
#pragma unroll
        for (int q = 0; q < 16; q++) { const int rn = j0 + 16 + w + 1 + q; pre[q] = (l <= w && rn < N) ? Kb[(size_t)rn * ld + l] : 0.0; }
        wave_sync();
        const int j1 = min(N, j0 + 16);
        for (int j = j0; j < j1; j++) {
            const int jm = j & GM, r = j + l + 1;
            const double d = win[jm * G + jm];
            const bool mine = (l < w) && (r < N);
            double la = 0.0;
            if (mine) {
                la = win[(r & GM) * G + jm] / d;
                KF[((size_t)(j & ~GM) + (r & GM)) * G + jm] = la;
            }
            if (l == 0) Kd[j] = 1.0 / d;
            const double lad = la * d;
#pragma unroll
            for (int bb = 1; bb < G; bb++) {
                const double lb = g_bcast<G>(la, bb - 1);          // L[j + bb][j]
                if (mine && bb <= l + 1) win[(r & GM) * G + ((j + bb) & GM)] -= lad * lb;
            }
            wave_sync();
            const int rn = j + w + 1;
            if (l <= w && rn < N) win[(rn & GM) * G + ((j + 1 + l) & GM)] = stage[(j - j0) * G + l];
            wave_sync();
        }
    }
    c.bytes += c.db->by[BY_FACTOR_LDS];
    c.cFact++;
    SPROF(c, SP_FACTOR);
}

// ---- band LDL' with the rows in registers and the pivot row through LDS (G <= 16) ---------------------------------------------------
// Round 5.  Lane l of the group holds ONE row r (r % G == l) of the part of the band that is still to be eliminated, in UPPER form
// relative to its own diagonal: wr[k] = K[r][r + k], k = 0 .. G-1 (by symmetry the entries of column r below the diagonal).  At step j
// the lane of row j puts its row into the group's LDS buffer; every other lane -- row i = j + a, a = 1 .. G-1 -- reads the part of the
// pivot row that reaches its own columns, p[k] = K[j][i + k] = buf[a + k] (one LDS read per entry at a lane-dependent address; behind the
// G entries of the buffer lie G zeros, so what is outside the band needs no predicate), forms its multiplier L[i][j] = p[0] / d_j as
// p[0] * (1 / d_j) and subtracts L[i][j] * p[k] from its row.  The lane whose row is finished takes over row j + G (gathered three
// blocks ahead from the instance's assembled rows).  Per step: 4 LDS writes, G + 1 LDS reads, one division, G fused multiply-adds.
// (Round 4 kept row r in LOWER form with the entry of column c in register slot c % G, so that every access had a static index, and
// moved the pivot and the G-1 multipliers of a step through the lane group by DPP: six moves and selects per double, two divisions,
// about a hundred instructions and 780 clocks per step for G = 8 -- the factorisation was the longest chain of an instance,
// profiles/round5/sparse_sched_profile_small_batches.log.)  The oracle's band_factor does the same arithmetic in the same order.
// Rows >= N are identity rows.  What depends on the working set is applied while a row is loaded: bgate (shared by the batch) names the
// row of E whose membership gates an entry (-1: none), bdiag what the diagonal is (>= -1 a variable: Q_ii, in K0, + dprim; -2 - rr the
// constraint row rr; INT_MIN a border position).  K0 / bgate are in the same upper form: entry k of row r is K[r + k][r].
__shared__ double sp_piv_lds[2 * WGS];      // per lane group: the pivot row (G doubles) and G zeros behind it
template <int G, class Dd, class Use>
__device__ __forceinline__ void sp_factor_reg(SpCtx<G>& c, GD KF, GD Kd, double dprim, Dd ddual, Use use)
{
    constexpr int GM = G - 1;
    const int N = c.db->N, Np = c.db->Np, l = here(c.gl);
    GD K0 = c.K0();
    const int NG = (N + GM) & ~GM;
    const int* __restrict__ bgate = c.db->bgate;
    const int* __restrict__ bdiag = c.db->bdiag;
    double* buf = sp_piv_lds + (size_t)(here((int)threadIdx.x) / G) * (2 * G);
    // A row is fetched in two halves: row_issue starts the loads (the instance's assembled row, the gates and the diagonal code: 9 x 16
    // bytes) and row_finish, one block of G steps later, applies what depends on the working set.  (Until round 5 both sat at the top of a
    // block: the gating consumed the loads it had just issued, one memory round trip per block of G steps -- most of a factorisation's
    // time.)
    struct RawRow { double val[G]; int gate[G]; int bd; };
    auto row_issue = [&](RawRow& rw, int r) {
        const int rr = (r < N) ? r : 0;      // (rows >= N are identity rows: the loads are harmless, row_finish ignores them)
#pragma unroll
        for (int k = 0; k < G; k += 4) { const int4 g4 = *reinterpret_cast<const int4*>(bgate + (size_t)rr * G + k); rw.gate[k] = g4.x; rw.gate[k + 1] = g4.y; rw.gate[k + 2] = g4.z; rw.gate[k + 3] = g4.w; }
        rw.bd = bdiag[rr];
#pragma unroll
        for (int k = 0; k < G; k += 2) { const dv2 v = K0.ld2(rr * G + k); rw.val[k] = v.x; rw.val[k + 1] = v.y; }
    };
    auto row_finish = [&](double* dst, const RawRow& rw, int r) {
        if (r < N) {
#pragma unroll
            for (int k = 1; k < G; k++) dst[k] = (rw.gate[k] < 0 || use(rw.gate[k])) ? rw.val[k] : 0.0;
            const int bd = rw.bd;
            if (bd == INT_MIN) dst[0] = 1.0;                           // a border position: an isolated unit pivot of the band
            else if (bd >= -1) dst[0] = rw.val[0] + dprim;
            else { const int rr = -2 - bd; dst[0] = use(rr) ? -ddual(rr) : -1.0; }
        } else {
#pragma unroll
            for (int k = 0; k < G; k++) dst[k] = (k == 0) ? 1.0 : 0.0;
        }
    };
    // G = 8: the lane's row of the next block is held gated (nx) and the rows of the two blocks behind it are in flight (ra, rb): a row is
    // requested FOUR blocks before it is the pivot row's neighbour and gated two blocks after the request -- under load a round trip to
    // memory is longer than the ~2.3 us a block of eight steps takes, and with one block between request and use the chain waited for it at every
    // block (the factorisation steps of a full machine took 1.7 x the time of a lone instance).  The block loop is unrolled by two so that
    // each buffer is named statically (a copy from one to the other would wait for the load).  G = 16 has registers for one gated row
    // ahead and one in flight only (DEEP: with a second one in flight scratch).
    constexpr bool DEEP = (G == 8);
    double wr[G], nx[G], kf[DEEP ? G : 1];
    RawRow ra, rb;
    row_issue(ra, l); row_finish(wr, ra, l);
    row_issue(ra, G + l); row_finish(nx, ra, G + l);
    row_issue(ra, 2 * G + l);
    if (DEEP) row_issue(rb, 3 * G + l);
    buf[G + l] = 0.0;                                // the zeros behind the pivot row
    double rinv = 1.0;
    auto block = [&](int j0, RawRow& raw) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int ag = (l - u) & GM;                               // this lane holds row j + ag, j = j0 + u (ag == 0: the pivot row)
            if (ag == 0) {
#pragma unroll
                for (int k = 0; k < G; k += 2) { dv2 v; v.x = wr[k]; v.y = wr[k + 1]; *reinterpret_cast<dv2*>(buf + k) = v; }
            }
            asm volatile("" ::: "memory");      // (LDS traffic of one wavefront is in order: the reads below see the pivot lane's stores)
            const double ri = 1.0 / buf[0];
            double p[G];
#pragma unroll
            for (int k = 0; k < G; k++) p[k] = buf[ag + k];              // K[j][j + ag + k]: zero beyond the band (the padding)
            asm volatile("" ::: "memory");
            const double la = (ag != 0) ? p[0] * ri : 0.0;                // L[j + ag][j]
            if (DEEP) kf[u] = la;
            else KF[(j0 + l) * G + u] = la;                                 // (G = 16: no registers for a block of the factor; one 8-byte store per step)
            if (ag == 0) {                                               // row j is finished: row j + G enters this lane
                rinv = ri;
#pragma unroll
                for (int k = 0; k < G; k++) wr[k] = nx[k];
            } else {
#pragma unroll
                for (int k = 0; k < G; k++) wr[k] -= la * p[k];
            }
        }
        // forward layout: row (j0 + l) of the block holds L[.][j0 + u] in column u (zero on and above this lane's own step)
        if (DEEP) {
#pragma unroll
            for (int k = 0; k < G; k += 2) { dv2 v; v.x = kf[k]; v.y = kf[k + 1]; *reinterpret_cast<dv2*>(reinterpret_cast<char*>(KF.base) + (size_t)(KF.off + (unsigned)((j0 + l) * G + k) * 8u)) = v; }
        }
        if (j0 + l < Np) Kd[j0 + l] = rinv;
        row_finish(nx, raw, j0 + 2 * G + l);          // requested two blocks ago (G = 16: one)
        row_issue(raw, j0 + (DEEP ? 4 : 3) * G + l);
    };
    if (DEEP) {
        for (int j0 = 0; j0 < NG; j0 += 2 * G) {
            block(j0, ra);
            if (j0 + G < NG) block(j0 + G, rb);
        }
    } else {
        for (int j0 = 0; j0 < NG; j0 += G) block(j0, ra);
    }
    c.bytes += c.db->by[BY_FACTOR];      // matrix entries read, factor and 1/D written
    c.cFact++;
    SPROF(c, SP_FACTOR);
}
// the band part of the KKT matrix [Q + dprim I, E_use'; E_use, -diag(ddual)] factorised: assembled on the fly (G <= 16) or through the band array
// ---- general sparse LDL' (round 6): multifrontal over the dissection tree, one wavefront per instance ------------------------------------
// Symbolic side: lcqp_sparse_general.hpp (fronts in postorder: pivots = a leaf region or a separator, update rows = the boundary of the
// region the front closes; assembly lists; positions of a child's update rows in its parent's front; storage offsets).  CPU restatement of
// the loops below, checked against a dense solve: tests/cpp/general_ldl_test.cpp.  The reference's OSQP arm factorises the same matrix
// with QDLDL whatever the pattern (src/SubsolverOSQP.cpp:136-152).
// A front F (ff x ff, column-major, lower triangle used) lives in LDS when ff <= 64, else in the instance's front buffer; its pivots go
// in blocks of GEN_JB: the block's columns (the panel, rows below included) are staged in LDS, eliminated there, stored scaled into the
// factor's panel storage, and applied to the rest of the front as ONE rank-GEN_JB update -- GEN_JB fused multiply-adds per entry read and
// written.  No pivoting: K is quasi-definite for delta, delta2 > 0, and every symmetric permutation of a quasi-definite matrix factorises.
constexpr int GEN_JB = 8;
constexpr int GEN_LDS_FRONT = 64;        // fronts up to this size are factorised inside LDS
constexpr int GEN_MAX_FRONT = 576;       // panel of the largest front: 576 x 8 doubles beside nothing else in the 40 KB of a wavefront

constexpr int GEN_META = 12;
constexpr int GEN_BITS_OFF = GEN_MAX_FRONT * GEN_JB + 16;        // doubles: the working set as a bit set behind the panel and 1 / D of a block
constexpr int GEN_BITS_WORDS = (GEN_LDS_FRONT * GEN_LDS_FRONT + 16 * 64 - GEN_BITS_OFF) * 2;      // 32-bit words that fit the rest of the window

// `in(r)`: is row r of E in the working set -- a bit in LDS (sp_general_factor builds the set once per factorisation: the gate of an entry and
// the diagonal of a row node are then no round trip to memory)
template <bool LDSF, class FA, class Dd, class In>
__device__ __forceinline__ void sp_general_front(SpCtx<64>& c, int f, FA F, double* P, GD Lst, GD Kd, GD stack, double dprim, Dd ddual, In in)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl), n = db.n, nnzQ = db.nnzQ;
    // everything the front needs to know about itself in ONE load (the dependent chain of a front is what a factorisation costs: about
    // 33 us per front with a load per field, sixteen round trips; profiles/round6/general_ldl_timing.log)
    const int* mt = db.gMeta + (size_t)f * GEN_META;
    const int np = mt[0], nb = mt[1], piv0 = mt[2], asm0 = mt[4], asm1 = mt[5], ch0 = mt[6], ch1 = mt[7];
    const unsigned Loff = (unsigned)mt[8], CBoff = (unsigned)mt[9];
    const int ff = np + nb;
    auto sync = [&]() { if (LDSF) wave_sync(); else g_sync(); };
#ifdef GEN_PROFILE      // (diagnostic: the parts of a front on the profile slots the rest of the engine hardly uses: zero -> products, assembly -> status test, children -> vectors, elimination -> factorisation, stores -> rhs)
#define GPROF(c, P) SPROF(c, P)
#else
#define GPROF(c, P) do { } while (0)
#endif
    GPROF(c, SP_LCQP);
    for (int e = t; e < ff * ff; e += 64) F[e] = 0.0;
    sync();
    GPROF(c, SP_PRODUCTS);
    {   // the entries of K whose column is a pivot of this front (one entry of Q or E each: distinct positions); rows of E gated by value
        GD Qv = c.Qx(), Ev = c.Ex();
        for (int e = asm0 + t; e < asm1; e += 64) {
            const int src = db.gAsmSrc[e], gate = db.gAsmGate[e], pos = db.gAsmPos[e];
            const double v = (src >= nnzQ ? (double)Ev[src - nnzQ] : (double)Qv[src]);
            F[pos] = (gate >= 0 && !in(gate)) ? 0.0 : v;
        }
    }
    sync();
    for (int j = t; j < np; j += 64) {      // diagonals: Q_ii + dprim; -ddual for an active row, -1 for a decoupled one
        const int node = db.pnode[piv0 + j];
        if (node < n) F[j + ff * j] += dprim;
        else F[j + ff * j] = in(node - n) ? -ddual(node - n) : -1.0;
    }
    sync();
    GPROF(c, SP_ASSEMBLE);
    for (int ci = ch0; ci < ch1; ci++) {      // extend-add: the children's update blocks, one child after the other
        const int* cm = db.gChildInfo + (size_t)ci * 4;
        const int nbc = cm[0];
        GD CB = stack + cm[1];
        const int* rel = db.gRel + cm[2];
        for (int e = t; e < nbc * nbc; e += 64) {
            const int b = e / nbc, a = e - b * nbc;
            if (a >= b) F[rel[a] + ff * rel[b]] += (double)CB[a + nbc * b];
        }
        sync();
    }
    GPROF(c, SP_VECTORS);
    double* dv = c.win + GEN_MAX_FRONT * GEN_JB;      // 1 / D of the block's pivots (behind the largest panel either variant uses)
    GD Lp = Lst + (int)Loff;
    for (int j0 = 0; j0 < np; j0 += GEN_JB) {
        const int jb = min(GEN_JB, np - j0), h = ff - j0;
        if (h <= 256) {
            // The panel of the block in REGISTERS: lane t holds rows t, t + 64, t + 128, t + 192 of the panel, eight entries each.  The eight
            // pivots are eliminated with v_readlane broadcasts (the pivot rows are rows 0 .. 7: lanes 0 .. 7 of the first register set) --
            // a division and at most seven broadcast + fused multiply-add steps per pivot, ~100 clocks, where the version through LDS below paid a
            // dependent LDS round trip per step (~2000 clocks per pivot with one wavefront per SIMD: 23 % + 16 % of a factorisation's time).
            // Same operations on the same values: li = p[cc] / d, p[c2] -= li * P[c2][cc]; rows above the diagonal compute entries nobody reads.
            constexpr int QP = 4;
            double p[QP][GEN_JB];
#pragma unroll
            for (int q = 0; q < QP; q++)
#pragma unroll
                for (int cc = 0; cc < GEN_JB; cc++) { const int i = t + 64 * q; p[q][cc] = (i < h && cc < jb) ? (double)F[(j0 + i) + ff * (j0 + cc)] : 0.0; }
            double dvr[GEN_JB];
#pragma unroll
            for (int cc = 0; cc < GEN_JB; cc++) {
                dvr[cc] = 0.0;
                if (cc < jb) {
                    const double dinv = 1.0 / wave_bcast(p[0][cc], cc);
                    dvr[cc] = dinv;
                    if (t == 0) { dv[cc] = dinv; Kd[piv0 + j0 + cc] = dinv; }
#pragma unroll
                    for (int c2 = cc + 1; c2 < GEN_JB; c2++) {
                        if (c2 < jb) {
                            const double pc = wave_bcast(p[0][cc], c2);      // P[c2][cc], unscaled
#pragma unroll
                            for (int q = 0; q < QP; q++) p[q][c2] -= (p[q][cc] * dinv) * pc;
                        }
                    }
                }
            }
            // the scaled columns are the factor's panel; the unscaled ones go to LDS for the tiles below
#pragma unroll
            for (int q = 0; q < QP; q++) {
                const int i = t + 64 * q;
                if (i < h) {
#pragma unroll
                    for (int cc = 0; cc < GEN_JB; cc++) {
                        P[i * GEN_JB + cc] = p[q][cc];
                        if (cc < jb && i > cc) Lp[(j0 + i) + ff * (j0 + cc)] = p[q][cc] * dvr[cc];
                    }
                }
            }
            wave_sync();
        } else {
            // the panel of the block: rows j0 .. ff-1, columns j0 .. j0+jb-1 -> P[(i - j0) * JB + c]
            for (int e = t; e < h * GEN_JB; e += 64) { const int cc = e / h, i = e - cc * h; P[i * GEN_JB + cc] = (cc < jb) ? (double)F[(j0 + i) + ff * (j0 + cc)] : 0.0; }
            wave_sync();
            for (int cc = 0; cc < jb; cc++) {      // eliminate inside the panel (columns unscaled: column c holds l_ic d_c)
                const double dinv = 1.0 / P[cc * GEN_JB + cc];
                if (t == 0) { dv[cc] = dinv; Kd[piv0 + j0 + cc] = dinv; }
                for (int i = cc + 1 + t; i < h; i += 64) {
                    const double li = P[i * GEN_JB + cc] * dinv;
                    for (int c2 = cc + 1; c2 < jb; c2++) if (c2 <= i) P[i * GEN_JB + c2] -= li * P[c2 * GEN_JB + cc];
                }
                wave_sync();
            }
            // the scaled columns are the factor's panel (column-major, ld = ff)
            for (int e = t; e < h * jb; e += 64) { const int cc = e / h, i = e - cc * h; if (i > cc) Lp[(j0 + i) + ff * (j0 + cc)] = P[i * GEN_JB + cc] * dv[cc]; }
        }
        // rank-jb update of what lies behind the block, on the fp64 matrix cores: 16 x 16 tiles over the lower triangle of F[k0.., k0..],
        // each D = A B with A[i][c] = (l_ic d_c) / d_c ... = P[i][c] dv[c] and B[c][k] = P[k][c] (v_mfma_f64_16x16x4_f64, two per tile: c = 0..3, 4..7).
        // Operand lane map: A[i = lane & 15][c = lane >> 4], B[c = lane >> 4][k = lane & 15]; result register r of a lane: row (lane >> 4) + 4 r,
        // column lane & 15 (cdna_hip_programming.md).  A row or column outside the front feeds only results that are not written.
        // (The VALU version -- a lane per row, eight fused multiply-adds behind four 16-byte LDS reads per entry, half the lanes idle in a
        // triangle -- was 41 % of a factorisation: profiles/round6/README.md.)
        const int k0 = j0 + jb;
        {
            const int lr = t >> 4, lc = t & 15;
            const double dva = (lr < jb) ? dv[lr] : 0.0, dvb = (lr + 4 < jb) ? dv[lr + 4] : 0.0;
            const int nt = (ff - k0 + 15) >> 4;
            for (int ti = 0; ti < nt; ti++) {
                const int ia = k0 + 16 * ti + lc;
                const double a0 = (ia < ff && lr < jb) ? P[(ia - j0) * GEN_JB + lr] * dva : 0.0;
                const double a1 = (ia < ff && lr + 4 < jb) ? P[(ia - j0) * GEN_JB + lr + 4] * dvb : 0.0;
                for (int tk = 0; tk <= ti; tk++) {
                    const int kb = k0 + 16 * tk + lc;
                    const double b0 = (kb < ff && lr < jb) ? P[(kb - j0) * GEN_JB + lr] : 0.0;
                    const double b1 = (kb < ff && lr + 4 < jb) ? P[(kb - j0) * GEN_JB + lr + 4] : 0.0;
                    d4_t acc = {0.0, 0.0, 0.0, 0.0};
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                    double fv[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int i = k0 + 16 * ti + lr + 4 * r; fv[r] = (i < ff && kb < ff && i >= kb) ? (double)F[i + ff * kb] : 0.0; }
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int i = k0 + 16 * ti + lr + 4 * r; if (i < ff && kb < ff && i >= kb) F[i + ff * kb] = fv[r] - acc[r]; }
                }
            }
        }
        sync();
    }
    GPROF(c, LDSF ? SP_FACTOR : SP_LCQP);      // (elimination of a front in LDS / of a front in memory)
    {   // the update block goes onto the stack (its place was fixed by the host: where its children's blocks lay)
        GD CB = stack + (int)CBoff;
        for (int e = t; e < nb * nb; e += 64) { const int b = e / nb, a = e - b * nb; if (a >= b) CB[a + nb * b] = (double)F[(np + a) + ff * (np + b)]; }
    }
    g_sync();
    GPROF(c, SP_RHS);
}

template <class Dd, class Use>
__device__ __forceinline__ void sp_general_factor(SpCtx<64>& c, GD Lst, GD Kd, double dprim, Dd ddual, Use use)
{
    const SpBatch& db = *c.db;
    double* Fl = c.win;                                       // 64 x 64 front in LDS
    double* P = c.win + GEN_LDS_FRONT * GEN_LDS_FRONT;        // panel of a front in LDS (64 x 8) ...
    GD stack = c.GStack(), Fg = c.GFront();
    SPROF(c, SP_VECTORS);
    // the working set as a bit set in LDS (behind everything a front uses of the window)
    unsigned* bits = reinterpret_cast<unsigned*>(c.win + GEN_BITS_OFF);
    const int m = db.m, words = (m + 31) >> 5;
    const bool haveBits = words <= GEN_BITS_WORDS;
    if (haveBits) {
        for (int w = c.gl; w < words; w += 64) {
            unsigned word = 0u;
#pragma unroll 8
            for (int k = 0; k < 32; k++) { const int r = w * 32 + k; if (r < m && use(r)) word |= 1u << k; }
            bits[w] = word;
        }
        wave_sync();
    }
    for (int f = 0; f < db.gnF; f++) {
        const int* mt = db.gMeta + (size_t)f * GEN_META;
        const int ff = mt[0] + mt[1];
        if (haveBits) {
            auto in = [=](int r) { return ((bits[r >> 5] >> (r & 31)) & 1u) != 0u; };
            if (ff <= GEN_LDS_FRONT) sp_general_front<true>(c, f, Fl, P, Lst, Kd, stack, dprim, ddual, in);
            else sp_general_front<false>(c, f, Fg, c.win, Lst, Kd, stack, dprim, ddual, in);      // ... or of a front in memory (up to GEN_MAX_FRONT x 8: the whole window)
        } else {
            if (ff <= GEN_LDS_FRONT) sp_general_front<true>(c, f, Fl, P, Lst, Kd, stack, dprim, ddual, use);
            else sp_general_front<false>(c, f, Fg, c.win, Lst, Kd, stack, dprim, ddual, use);
        }
    }
    c.bytes += db.by[BY_FACTOR];
    SPROF(c, SP_FACTOR);
}

// K z = b in place (b in the ordering of the fronts): forward over the fronts in postorder, 1 / D, backward in reverse.  Per front the
// right-hand side's entries (pivots and update rows) are gathered into LDS, the panel is staged in LDS in chunks of columns (all lanes load,
// many loads in flight) and the columns are applied one after the other: an axpy per column forward, a dot product per column backward.
template <bool FWD>
__device__ __forceinline__ void sp_general_sweep(SpCtx<64>& c, GD Lst, GD b)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl);
    double* bl = c.win;                         // ff entries
    double* Pc = c.win + GEN_MAX_FRONT;         // a chunk of columns: (ff) x cw, column-major
    constexpr int CHUNK = GEN_LDS_FRONT * GEN_LDS_FRONT + 16 * 64 - GEN_MAX_FRONT;      // doubles left in the window
    for (int q = 0; q < db.gnF; q++) {
        const int f = FWD ? q : db.gnF - 1 - q;
        const int* mt = db.gMeta + (size_t)f * GEN_META;
        const int np = mt[0], nb = mt[1], ff = np + nb, piv0 = mt[2];
        const int* rows = db.gRows + mt[3];
        GD Lp = Lst + mt[8];
        if (ff <= 64) {
            // A front of at most 64 rows (every leaf, every merged separator: most of the pivots): lane t IS row t.  The right-hand side lives in
            // one register per lane, a pivot's value travels by v_readlane, the panel's column (forward) or row (backward) comes from LDS with an
            // address that does not depend on the chain -- a pivot step is a broadcast and a fused multiply-add, ~30 clocks, where the version
            // through LDS (below, kept for larger fronts) paid a read - modify - write round trip of the right-hand side per pivot, ~1000 clocks
            // with one wavefront per SIMD.  Backward in axpy form too (a finished x_i leaves every earlier row), so no reduction sits in the chain.
            const int ldp = ff | 1;                                 // odd leading dimension: the row access of the backward sweep is free of bank conflicts
            double* Pf = c.win;
            wave_sync();
            // forward: lane t reads ITS entry of column j straight from the factor (coalesced; the addresses do not depend on the chain, so the
            // loads of all columns are in flight together) -- no staging in LDS; backward needs rows of the panel: staged with an odd leading dimension
#ifdef GEN_STAGE_BACKWARD
            if (!FWD) for (int e = t; e < ff * np; e += 64) { const int cc = e / ff, i = e - cc * ff; Pf[i + ldp * cc] = (i > cc) ? (double)Lp[i + ff * cc] : 0.0; }
#endif
            double x = (t < ff) ? (double)b[t < np ? piv0 + t : rows[t - np]] : 0.0;
            wave_sync();
            if (FWD) {
                for (int j0 = 0; j0 < np; j0 += 8) {
                    double lv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int j = j0 + u; lv[u] = (j < np && t > j && t < ff) ? (double)Lp[t + ff * j] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int j = j0 + u; if (j < np) { const double yj = wave_bcast(x, j); x -= lv[u] * yj; } }
                }
                if (t < ff) b[t < np ? piv0 + t : rows[t - np]] = x;
            } else {
#ifdef GEN_STAGE_BACKWARD
                for (int i = ff - 1; i >= 1; i--) {
                    const double xi = wave_bcast(x, i);
                    const double lit = (t < i && t < np) ? Pf[i + ldp * t] : 0.0;      // L[i][t]: a row of the panel
                    x -= lit * xi;
                }
#else
                // backward: lane t (a pivot) walks down ITS column of the panel, L[i][t] for i = ff-1 .. t+1 -- contiguous per lane (a cache line serves
                // eight steps), a stride of ff between the lanes; again no address depends on the chain, eight loads in flight (a row-major copy of
                // the panels for coalesced rows was measured: slower, 3.43 -> 3.59 s, the extra stores cost more than the strides)
                for (int i0 = ff - 1; i0 >= 1; i0 -= 8) {
                    double lv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int i = i0 - u; lv[u] = (i >= 1 && t < i && t < np) ? (double)Lp[i + ff * t] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int i = i0 - u; if (i >= 1) { const double xi = wave_bcast(x, i); x -= lv[u] * xi; } }
                }
#endif
                if (t < np) b[piv0 + t] = x;
            }
            g_sync();
            continue;
        }
        if (ff <= 256) {
            // the same for a front of up to 256 rows: lane t holds rows t, t + 64, t + 128, t + 192 in four registers
            constexpr int QR = 4;
            double x[QR];
#pragma unroll
            for (int q = 0; q < QR; q++) { const int i = t + 64 * q; x[q] = (i < ff) ? (double)b[i < np ? piv0 + i : rows[i - np]] : 0.0; }
            auto bc = [&](int i) {      // value of row i: a broadcast from the register of lane i & 63 that holds chunk i >> 6 (uniform selection)
                const int qi = i >> 6, li = i & 63;
                double v = wave_bcast(x[0], li);
                if (qi == 1) v = wave_bcast(x[1], li);
                if (qi == 2) v = wave_bcast(x[2], li);
                if (qi == 3) v = wave_bcast(x[3], li);
                return v;
            };
            if (FWD) {
                for (int j0 = 0; j0 < np; j0 += 4) {
                    double lv[4][QR];
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int q = 0; q < QR; q++) { const int j = j0 + u, i = t + 64 * q; lv[u][q] = (j < np && i > j && i < ff) ? (double)Lp[i + ff * j] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = j0 + u;
                        if (j < np) {
                            const double yj = bc(j);
#pragma unroll
                            for (int q = 0; q < QR; q++) x[q] -= lv[u][q] * yj;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < QR; q++) { const int i = t + 64 * q; if (i < ff) b[i < np ? piv0 + i : rows[i - np]] = x[q]; }
            } else {
                for (int i0 = ff - 1; i0 >= 1; i0 -= 4) {
                    double lv[4][QR];
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int q = 0; q < QR; q++) { const int i = i0 - u, j = t + 64 * q; lv[u][q] = (i >= 1 && j < i && j < np) ? (double)Lp[i + ff * j] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 - u;
                        if (i >= 1) {
                            const double xi = bc(i);
#pragma unroll
                            for (int q = 0; q < QR; q++) x[q] -= lv[u][q] * xi;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < QR; q++) { const int j = t + 64 * q; if (j < np) b[piv0 + j] = x[q]; }
            }
            g_sync();
            continue;
        }
        for (int i = t; i < ff; i += 64) bl[i] = (double)b[i < np ? piv0 + i : rows[i - np]];
        const int cw = max(1, min(np, CHUNK / ff));
        for (int c0 = FWD ? 0 : ((np - 1) / cw) * cw; FWD ? c0 < np : c0 >= 0; c0 += FWD ? cw : -cw) {
            const int c1 = min(np, c0 + cw), h = ff - c0;      // rows c0 .. ff-1 of the columns c0 .. c1-1
            wave_sync();
            for (int e = t; e < h * (c1 - c0); e += 64) { const int cc = e / h, i = e - cc * h; Pc[i + h * cc] = (i > cc) ? (double)Lp[(c0 + i) + ff * (c0 + cc)] : 0.0; }
            wave_sync();
            if (FWD) {
                for (int cc = 0; cc < c1 - c0; cc++) {
                    const double yj = bl[c0 + cc];
                    for (int i = cc + 1 + t; i < h; i += 64) bl[c0 + i] -= Pc[i + h * cc] * yj;
                    wave_sync();
                }
            } else {
                for (int cc = c1 - c0 - 1; cc >= 0; cc--) {
                    double sacc = 0.0;
                    for (int i = cc + 1 + t; i < h; i += 64) sacc += Pc[i + h * cc] * bl[c0 + i];
                    sacc = g_sum<64>(sacc);
                    if (t == 0) bl[c0 + cc] -= sacc;
                    wave_sync();
                }
            }
        }
        wave_sync();
        if (FWD) { for (int i = t; i < ff; i += 64) b[i < np ? piv0 + i : rows[i - np]] = bl[i]; }
        else { for (int i = t; i < np; i += 64) b[piv0 + i] = bl[i]; }
        g_sync();
    }
}

__device__ __forceinline__ void sp_general_solve(SpCtx<64>& c, bool admm, GD b)
{
    const SpBatch& db = *c.db;
    SPROF(c, SP_VECTORS);
    sp_general_sweep<true>(c, c.KF(admm), b);
    {
        GD Kd = c.KD(admm);
        g_map<64, 8>(db.N, c.gl, [&](int p) { return D2{b[p], Kd[p]}; }, [&](int p, D2 v) { b[p] = v.a * v.b; });
        g_sync();
    }
    SPROF(c, SP_FORWARD);
    sp_general_sweep<false>(c, c.KF(admm), b);
    SPROF(c, SP_BACKWARD);
    c.bytes += db.by[BY_SOLVE];
}

template <int G, class Dd, class Use>
__device__ __forceinline__ void sp_factor_band(SpCtx<G>& c, GD KF, GD Kd, double dprim, Dd ddual, Use use)
{
    if constexpr (G == 64) { if (c.db->general) { sp_general_factor(c, KF, Kd, dprim, ddual, use); return; } }
    if constexpr (G <= 16) sp_factor_reg<G>(c, KF, Kd, dprim, ddual, use);
    else { sp_assemble<G>(c, dprim, ddual, use); sp_factor_lds<G>(c, KF, Kd); }
}

// ---- band sweeps: L y = b, z = y / D (forward) and L' x = z (backward), in place ----------------------------------------------------
// Position p (0 .. Np-1 in processing order: row p forward, row Np-1-p backward) is pending in lane p % G of the group while steps
// p-G+1 .. p run; every step broadcasts the finished entry inside the group and every lane subtracts its multiple -- axpy form, no
// reduction in the chain.  The folded layouts put the coefficient lane u needs at step s of a block of G steps at K[(blk + u) * G + s]:
// one 8 G-byte row per lane and block, the rows of a group contiguous (G x G doubles per block), streamed RING chunks ahead of use.
// Right-hand sides and 1/D for the next 64 positions are loaded while the current 64 run.
template <int G, bool FWD>
__device__ __forceinline__ void band_sweep(GD K, GD Kd, GD b, int Np, int gl)
{
    constexpr int CH = G < 16 ? G : 16, NCHUNK = 64 / CH, BPS = 64 / G, RING = (G == 8) ? SP_SWEEP_RING : 2;
    static_assert(RING >= 2 && NCHUNK % RING == 0, "the ring slots are assigned statically per 64 positions: RING has to divide 64 / CH (6 gave wrong coefficients and a run without end)");
    gl = here(gl);
    auto at = [&](int p) -> int { return FWD ? p : Np - 1 - p; };
    double cf[RING][CH];
    auto load_chunk = [&](double* dst, int sb, int ck) {
        const int s0 = ck * CH, bi = s0 / G, so = s0 % G;
        if (sb < Np) {
            if (FWD) {
                const int e0 = (sb + bi * G + gl) * G + so;
#pragma unroll
                for (int q = 0; q < CH / 2; q++) { const dv2 v = K.ld2(e0 + 2 * q); dst[2 * q] = v.x; dst[2 * q + 1] = v.y; }
            } else {
                // the same array read the other way: at step k of a block (row r = Np-1-(pb+k) finishes) the lane whose pending
                // row is i needs L[r][i] = K[((i / G) * G + r % G) * G + i % G]; i % G = G-1-gl, r % G = G-1-k, and i lies in the
                // row block of r for gl > k, in the one below for gl < k.  One double per lane and step, a group reads 8 G bytes
                // of one or two rows.
                const int pb = sb + bi * G;
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int k = so + q;
                    const int iblk = Np - G - pb - (gl < k ? G : 0);
                    dst[q] = (iblk >= 0) ? K.ld((iblk + (G - 1 - k)) * G + (G - 1 - gl)) : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < CH; q++) dst[q] = 0.0;
        }
    };
    // rh[q], dl[q]: right-hand side and 1/D of this lane's position in block q of the current 64 positions; a slot is refilled for
    // the next 64 as soon as it has been consumed
    double rh[BPS], dl[BPS];
#pragma unroll
    for (int q = 0; q < BPS; q++) { rh[q] = b[at(q * G + gl)]; dl[q] = FWD ? Kd.ld(q * G + gl) : 1.0; }
#pragma unroll
    for (int ck = 0; ck < RING - 1; ck++) load_chunk(cf[ck], 0, ck);
    double cur = rh[0];
    if (64 < Np) rh[0] = b[at(64 + gl)];
    for (int sb = 0; sb < Np; sb += 64) {
        const bool more = sb + 64 < Np, more2 = sb + 128 < Np;
        double res = 0.0;
#pragma unroll
        for (int ck = 0; ck < NCHUNK; ck++) {
            { const int nck = ck + RING - 1; load_chunk(cf[nck % RING], sb + 64 * (nck / NCHUNK), nck % NCHUNK); }
#pragma unroll
            for (int q = 0; q < CH; q++) {
                const int s = ck * CH + q, k = s % G, bi = s / G;
                const double yj = g_bcast<G>(cur, k);
                cur -= cf[ck % RING][q] * yj;
                if (gl == k) { res = yj; cur = rh[(bi + 1) % BPS]; }      // block bi+1 of these 64, or block 0 of the next 64 (already refilled)
                if (k == G - 1) {                                          // block bi is complete: store it, refill its slots
                    b[at(sb + bi * G + gl)] = FWD ? res * dl[bi] : res;
                    if (bi + 1 < BPS) { if (more) { rh[bi + 1] = b[at(sb + 64 + (bi + 1) * G + gl)]; } }
                    else if (more2) rh[0] = b[at(sb + 128 + gl)];
                    if (FWD && more) dl[bi] = Kd.ld(sb + 64 + bi * G + gl);
                }
            }
        }
    }
}

template <int G>
__device__ __forceinline__ void sp_solve_band(SpCtx<G>& c, bool admm, GD b)
{
    if constexpr (G == 64) { if (c.db->general) { sp_general_solve(c, admm, b); return; } }
    const int Np = c.db->Np;
    SPROF(c, SP_VECTORS);
    band_sweep<G, true>(c.KF(admm), c.KD(admm), b, Np, c.gl);
    g_sync();
    SPROF(c, SP_FORWARD);
    band_sweep<G, false>(c.KF(admm), c.KD(admm), b, Np, c.gl);
    g_sync();
    SPROF(c, SP_BACKWARD);
    c.bytes += c.db->by[BY_SOLVE];
}

// ---- the border (oracle: kkt_factor / kkt_solve) ---------------------------------------------------------------------------------
// After the band factorisation: the gated values of U (sp_border_prepare), W = U inv(Bd) -- one band solve per border node, run by the
// caller through ITS call site of the band solve (sp_border_column scatters column j; the kernel carries one copy of the sweeps per
// context) --, then S = C - W U' and its LDL' (sp_border_schur).
template <int G, class Use>
__device__ __forceinline__ void sp_border_prepare(SpCtx<G>& c, bool admm, Use use)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl), kb = db.kb, Np = db.Np, nnzQ = db.nnzQ, nU = db.nU;
    GD W = c.BW(admm), Uv = c.BUv(admm), Qv = c.Qx(), Ev = c.Ex();
    for (int e = t; e < nU; e += G) {
        const int src = db.Usrc[e], gate = db.Ugate[e];
        Uv[e] = (gate >= 0 && !use(gate)) ? 0.0 : (src >= nnzQ ? (double)Ev[src - nnzQ] : (double)Qv[src]);
    }
    for (int p = t; p < kb * Np; p += G) W[p] = 0.0;
    g_sync();
}
template <int G>
__device__ __forceinline__ GD sp_border_column(SpCtx<G>& c, bool admm, int b)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl);
    GD wb = c.BW(admm) + b * db.Np, Uv = c.BUv(admm);
    for (int e = db.Uptr[b] + t; e < db.Uptr[b + 1]; e += G) wb[db.Upos[e]] = Uv[e];
    g_sync();
    return wb;
}
template <int G, class Dd, class Use>
__device__ __forceinline__ void sp_border_schur(SpCtx<G>& c, bool admm, double dprim, Dd ddual, Use use)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl), kb = db.kb, Np = db.Np, nnzQ = db.nnzQ, nvar = db.n;
    GD W = c.BW(admm), Uv = c.BUv(admm), S = c.BS(admm), Qv = c.Qx(), Ev = c.Ex();
    // C: the border block itself (lane 0; a handful of entries), then S = C - W U'
    if (t == 0) {
        for (int e = 0; e < kb * kb; e++) S[e] = 0.0;
        for (int a = 0; a < kb; a++) {
            const int node = db.bnode[a];
            double dg;
            if (node < nvar) { const int qd = db.qdiag[node]; dg = (qd >= 0 ? (double)Qv[qd] : 0.0) + dprim; }
            else { const int rr = node - nvar; dg = use(rr) ? -ddual(rr) : -1.0; }
            S[a * kb + a] = dg;
            for (int e = db.Cptr[a]; e < db.Cptr[a + 1]; e++) {
                const int b2 = db.Cb2[e], src = db.Csrc[e], gate = db.Cgate[e];
                const double v = (gate >= 0 && !use(gate)) ? 0.0 : (src >= nnzQ ? (double)Ev[src - nnzQ] : (double)Qv[src]);
                S[a * kb + b2] += v; S[b2 * kb + a] += v;
            }
        }
    }
    g_sync();
    for (int a = 0; a < kb; a++)
        for (int b2 = 0; b2 < kb; b2++) {
            GD wb = W + b2 * Np;
            double sacc = 0.0;
            for (int e = db.Uptr[a] + t; e < db.Uptr[a + 1]; e += G) sacc += Uv[e] * wb[db.Upos[e]];
            sacc = g_sum<G>(sacc);
            if (t == 0) S[a * kb + b2] -= sacc;
        }
    g_sync();
    if (t == 0) {      // S = L D L' in place: L below the diagonal, D on it (quasi-definite: no pivoting)
        for (int j = 0; j < kb; j++) {
            double d = S[j * kb + j];
            for (int k = 0; k < j; k++) { const double ljk = S[j * kb + k]; d -= ljk * ljk * S[k * kb + k]; }
            S[j * kb + j] = d;
            for (int i = j + 1; i < kb; i++) {
                double v = S[i * kb + j];
                for (int k = 0; k < j; k++) v -= S[i * kb + k] * S[j * kb + k] * S[k * kb + k];
                S[i * kb + j] = v / d;
            }
        }
    }
    g_sync();
    c.bytes += db.by[BY_BORDER_PREPARE];
}
// After the band solve of b (the border positions pass through it untouched): the border unknowns from S, then the band part corrected
template <int G>
__device__ __forceinline__ void sp_border_solve(SpCtx<G>& c, bool admm, GD b)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl), kb = db.kb, Np = db.Np, Nb = db.N - db.kb;
    GD W = c.BW(admm), Uv = c.BUv(admm), S = c.BS(admm);
    for (int a = 0; a < kb; a++) {
        double sacc = 0.0;
        for (int e = db.Uptr[a] + t; e < db.Uptr[a + 1]; e += G) sacc += Uv[e] * b[db.Upos[e]];
        sacc = g_sum<G>(sacc);
        if (t == 0) b[Nb + a] -= sacc;
    }
    g_sync();
    if (t == 0) {
        for (int i = 0; i < kb; i++) { double v = b[Nb + i]; for (int k = 0; k < i; k++) v -= S[i * kb + k] * b[Nb + k]; b[Nb + i] = v; }
        for (int i = 0; i < kb; i++) b[Nb + i] = b[Nb + i] / S[i * kb + i];
        for (int i = kb - 1; i >= 0; i--) { double v = b[Nb + i]; for (int k = i + 1; k < kb; k++) v -= S[k * kb + i] * b[Nb + k]; b[Nb + i] = v; }
    }
    g_sync();
    for (int p = t; p < Nb; p += G) {
        double acc = 0.0;
        for (int a = 0; a < kb; a++) acc += W[a * Np + p] * b[Nb + a];
        b[p] -= acc;
    }
    g_sync();
    c.bytes += db.by[BY_BORDER_SOLVE];
}
// K x = b in place: the band solve, then the border
template <int G>
__device__ __forceinline__ void sp_solve(SpCtx<G>& c, bool admm, GD b)
{
    sp_solve_band<G>(c, admm, b);
    if (c.db->kb > 0) sp_border_solve<G>(c, admm, b);
}

// ---- ADMM iterations (OSQP, KKT form; oracle: sqp_admm) ----------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void sp_admm(SpCtx<G>& c, GD g, int n_it)
{
    const SpBatch& db = *c.db;
    const int t = c.gl, n = db.n, m = db.m;
    const double alpha = db.opt.admmAlpha, sigma = c.info->sigma;
    GD xa = c.V(NV_XA), ya = c.M(MV_YA), za = c.M(MV_ZA), b = c.Nv();
    GD l = c.M(MV_L), u = c.M(MV_U), rhov = c.M(MV_RHOV);
    for (int it = 0; it < n_it; it++) {
        for (int i = t; i < n; i += G) b[db.iperm[i]] = sigma * xa[i] - g[i];
        for (int r = t; r < m; r += G) b[db.iperm[n + r]] = za[r] - ya[r] / rhov[r];
        g_sync();
        sp_solve<G>(c, true, b);
        for (int r = t; r < m; r += G) {
            const double rv = rhov[r];
            const double zt = za[r] + (b[db.iperm[n + r]] - ya[r]) / rv;
            const double zr = alpha * zt + (1.0 - alpha) * za[r];
            if (isinf(l[r]) && isinf(u[r])) { za[r] = zr; ya[r] = 0.0; continue; }
            const double zn = fmin(fmax(zr + ya[r] / rv, l[r]), u[r]);
            ya[r] += rv * (zr - zn);
            za[r] = zn;
        }
        for (int i = t; i < n; i += G) xa[i] = alpha * b[db.iperm[i]] + (1.0 - alpha) * xa[i];
        g_sync();
        c.cAdmm++;
    }
}

// ---- the phases of an instance's homotopy (k_sparse_sched runs them) -----------------------------------------------------------------------
// Every routine below is called by the G lanes of one instance with that instance's context and state record (all values uniform inside the
// group) and returns the phase the instance enters next (PH_NUM: finished).  Together they are runSolver (src/LCQProblem.cpp:444-560), the
// subsolver call (oracle: sqp_solve) and the polish (oracle: sqp_polish) of round 3, cut at the points where instances of one wavefront used
// to part: before a trial, before a factorisation, before a correction, at the end of a QP.
struct StRow { int s; double e, lo, hi, y; };
struct I2 { int a, b; };
struct I2D { int a, b; double y; };
struct ID4 { int p, s; double lo, hi, e; };
struct ID2 { int s; double v, y; };
struct ID3 { int s; double lo, hi, z, y; };

// the polish starts (oracle: sqp_polish, entry): tolerances scale with 1 + |g|_inf
template <int G>
__device__ __forceinline__ int sp_polish_begin(SpCtx<G>& c, SpState& S, GD g, int reuse)
{
    const lcqp_options_t& o = c.db->opt;
    S.gs = 1.0 + ((reuse && S.gmaxNext >= 0.0) ? S.gmaxNext : sp_maxabs<G>(c, g, c.db->n));      // (the LCQP level knows max|g| of the vector it has just formed)
    S.ytol = o.feasTol * S.gs;
    S.fact_valid = 0; S.borderTodo = 0;
    S.dpUsed = c.info->delta; S.d2Used = c.info->delta2;      // regularisation of the factorisation in use
    S.trial = 0; S.reuse = reuse; S.nrefine = 0; S.xinf = 0.0;
    return PH_TRIAL;
}

// the instance is finished: statistics and solution go out
template <int G>
__device__ __forceinline__ int sp_finish(SpCtx<G>& c, SpState& S)
{
    const SpBatch& db = *c.db;
    const int t = here(c.gl), n = db.n, m = db.m;
    GD xk = c.V(NV_XK), yk = c.M(MV_YK);
    S.st.status = S.algoStat; S.st.returnValue = S.rc;
    S.st.admmIter = c.cAdmm; S.st.trials = c.cTrials; S.st.factorizations = c.cFact; S.st.corrections = c.cCorr; S.st.reserved = c.cSweeps;
    for (int i = t; i < n; i += G) db.xout[(size_t)c.b * n + i] = xk[i];
    for (int r = t; r < m; r += G) db.yout[(size_t)c.b * m + r] = yk[r];
    if (t == 0) { db.stats[c.b] = S.st; c.info->bytes += c.bytes; }
    c.bytes = 0.0;
    return PH_NUM;
}

// a polish that did not settle (oracle: the tail of the round loop of sqp_solve): twice as many ADMM iterations, or the QP has failed
template <int G>
__device__ __forceinline__ int sp_polish_failed(SpCtx<G>& c, SpState& S)
{
    const lcqp_options_t& o = c.db->opt;
    S.n_admm = 2 * S.n_admm;
    if (S.n_admm < 10) S.n_admm = 10;
    if (S.n_admm > 400) S.n_admm = 400;
    S.round++;
    if (S.round < o.maxRounds) return PH_ROUND;
    S.qpIter = (c.cTrials - S.trials0) + (c.cAdmm - S.admm0);
    S.st.subproblemIter += S.qpIter; S.st.qpSolverExitFlag = 1; S.st.qpSolves++;
    S.rc = LCQP_SUBPROBLEM_SOLVER_ERROR;
    return sp_finish<G>(c, S);
}

// PH_TRIAL: the head of one trial of the polish -- E x, the status test, the true residual when the working set did not change, acceptance;
// else the leaving rows, and whether the factorisation still matches the working set
template <int G>
__device__ __forceinline__ int sp_ph_trial(SpCtx<G>& c, SpState& S, GD g)
{
    const SpBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = c.gl, n = db.n, m = db.m, trial = S.trial;
    GD x = c.V(NV_XT), r1 = c.V(NV_R1), qx = c.V(NV_TMP), yt = c.M(MV_YT), ex = c.M(MV_EX);
    GD l = c.M(MV_L), u = c.M(MV_U);
    GI st = c.I(MI_STT), stf = c.I(MI_STF), newst = c.I(MI_NEW);
    const double gs = S.gs, ytol = S.ytol;
    c.cTrials++;
    // Two stages, as on the dense path (oracle: sqp_polish).  Stage 1 is what every trial needs: E x, for the status test.  Stage 2 -- the
    // true residual, one pass over Q and one over E' -- runs only when stage 1 changed nothing: after a correction the residual is zero on
    // the old working set up to rounding and regularisation, so when the set changes the next right-hand side is known without it (the
    // multipliers of the leaving rows, below); the trial that accepts always has the true residual.
    double res_stat = 0.0;
    int have_r1 = 0;
    double res_eq = 0.0, bmax = 0.0;
    int chg = 0, act = 0, loose = 0;
    // active rows are held to the rounding floor of a computed E_r x, 16 eps (|b_r| + |E_r|_1 |x|_inf), before a point is accepted (round 5;
    // oracle: sqp_polish; dense twin: qp_polish in lcqp_dev.hpp): runSolver ends on phi < 1e3 eps, a sum of products of such residuals
    const double exScale = c.info->e1max * S.xinf;
    auto status = [&](int r, StRow v) {
        int ns = v.s;
        if (v.s == ST_INACT) {
            const double ftol = o.feasTol * (1.0 + fabs(v.e));
            if (v.e < v.lo - ftol) ns = ST_LOWER;
            else if (v.e > v.hi + ftol) ns = ST_UPPER;
        } else {
            const double bb = (v.s == ST_UPPER) ? v.hi : v.lo;
            const bool above = !(fabs(bb - v.e) <= 16.0 * 2.221e-16 * (fabs(bb) + exScale));      // (a NaN counts) a row at the rounding floor of its computed E_r x is not a residual (round 6)
            if (above) res_eq = nmax(res_eq, fabs(bb - v.e));
            bmax = fmax(bmax, fabs(bb));
            loose |= above;
            if (v.s == ST_LOWER && v.y > ytol) ns = ST_INACT;
            if (v.s == ST_UPPER && v.y < -ytol) ns = ST_INACT;
        }
        newst[r] = ns;
        chg += (ns != v.s);
        act += (ns != ST_INACT);
    };
    if (trial == 0 && S.reuse) {
        // hot start with an unchanged (x, y): r1 = r1_last + (g_last - g) and E x are in place -- the residual and E x of the accepted
        // trial stay where they are, and the LCQP level adds (g_last - g) to r1 in the pass that forms the new g (sp_ph_qpend)
        have_r1 = 1;
        g_map<G, 4>(m, t, [&](int r) { return StRow{st[r], ex[r], l[r], u[r], yt[r]}; }, status);
    } else {
        // E x and the status test in ONE pass (round 5): the row's state rides along as the product's `pre`, the sum goes straight into the
        // test and into ex (the right-hand side of the correction needs it) -- no second pass over st, ex, l, u, y
        SPROF(c, SP_VECTORS);
        sp_ell<G, false>(db.ellE, c.gl, c.Ex(), [&](int j) { return D2{x[j], 0.0}; },
                  [&](int r) { return StRow{st[r], 0.0, l[r], u[r], yt[r]}; },
                  [&](int r, double s0, double, StRow v) { ex[r] = s0; v.e = s0; status(r, v); });
        g_sync();
        SPROF(c, SP_PRODUCTS);
        c.bytes += db.by[BY_EX];
    }
    const int changed = g_sum_i<G>(chg), nact = g_sum_i<G>(act);
    res_eq = g_max<G>(res_eq);
    bmax = g_max<G>(bmax);
    SPROF(c, SP_ASSEMBLE);      // (profile builds: the status test on its own)
    double rscale = 0.0;
    if (!have_r1 && (trial == 0 || !changed)) {
        res_stat = sp_residual<G>(c, g, x, yt, r1, qx, rscale);
        c.cSweeps++;
        c.bytes += db.by[BY_SWEEP];
        have_r1 = 1;
    }
    if (trial > 0 && !changed && res_stat <= fmax(o.resTol * gs, 64.0 * 2.221e-16 * rscale) && res_eq <= o.resTol * (1.0 + bmax)) {
        if (!(g_any<G>(loose) && S.nrefine < 2 && trial + 1 < o.maxTrials)) return PH_QPEND;      // a verified KKT point
        S.nrefine++;      // ... whose active rows can be held more exactly: one more correction
    }
    if (changed && trial > 0) {
        if (trial >= 2 && nact > n && changed > max(n / 2, 32)) return sp_polish_failed<G>(c, S);       // overshooting cold start: hand over to ADMM
        // leaving rows: their multipliers leave the residual (r1 += E_r' y_r), then the new working set takes over
        GD ytmp = c.M(MV_LX);
        g_sync();
        g_map<G, 8>(m, t, [&](int r) { return I2D{newst[r], st[r], yt[r]}; },
                    [&](int r, I2D v) { const bool leaves = (v.a == ST_INACT && v.b != ST_INACT); ytmp[r] = leaves ? -v.y : 0.0; st[r] = v.a; if (leaves && v.y != 0.0) yt[r] = 0.0; });
        g_sync();
        // r1 - E'(-y_leaving) = r1 + E'y_leaving; without a true residual r1 is the predicted one: nothing was left on the old working set
        if (have_r1) sp_ETy<G>(c, ytmp, r1, [&](int i) { return r1[i]; }, [](double v) { return v; });
        else sp_ETy<G>(c, ytmp, r1, [](int) { return NoPre{}; }, [](NoPre) { return 0.0; });
        S.fact_valid = 0;
    }
    if (!S.fact_valid) {
        int diff = (c.info->stfValid == 0);
#pragma unroll 8
        for (int r = t; r < m; r += G) diff |= ((stf[r] != ST_INACT) != (st[r] != ST_INACT));
        if (g_any<G>(diff)) return PH_FACTOR;
        S.fact_valid = 1;
    }
    return PH_CORRECT;
}

// PH_FACTOR: the band LDL' of [Q + delta I, Ea'; Ea, -delta2 I] for the working set in MI_STT
template <int G>
__device__ __forceinline__ int sp_ph_factor(SpCtx<G>& c, SpState& S)
{
    const SpBatch& db = *c.db;
    const int t = c.gl, n = db.n, m = db.m;
    GI st = c.I(MI_STT), stf = c.I(MI_STF);
    // Two levels of regularisation, as on the dense path (proxSmall / proxBig).  A correction with the safe level leaves
    // delta dx and delta2 dy (1e-8, 1e-9 relative) in the true residuals: every QP paid one refinement trial -- a sweep, a band
    // solve, the vector passes -- for the regularisation alone.  The light level (1e-12, 1e-14) is accepted at once.  The band
    // LDL' is not pivoted, so the light level needs an ordering in which every row follows one of its variables (lightOK, chosen
    // by the host when every Hessian of the batch is safely definite) and is only kept when every pivot has the sign its node
    // prescribes and a safe size; a variable's pivot failing makes the safe level permanent for the instance.
    int level = (db.lightOK && !c.info->bigReg) ? 0 : 1;
    // the working set as a bit set in LDS: the factorisation asks for the membership of the row behind every entry of E it meets (up to
    // 2 w per band row) -- from memory these were gathers of 4-byte flags, 8 instances apart in one wavefront
    unsigned* bits = nullptr;
    if (G <= 16 && db.bitWords > 0) {
        bits = reinterpret_cast<unsigned*>(sp_dyn_lds) + (size_t)((int)threadIdx.x / G) * db.bitWords;
        for (int w = t; w < db.bitWords; w += G) {
            unsigned word = 0u;
#pragma unroll 8
            for (int k = 0; k < 32; k++) { const int r = w * 32 + k; if (r < m && st[r] != ST_INACT) word |= 1u << k; }
            bits[w] = word;
        }
        wave_sync();
    }
    for (;;) {
        S.dpUsed = level ? c.info->delta : c.info->deltaS;
        S.d2Used = level ? c.info->delta2 : c.info->delta2S;
        const double d2 = S.d2Used;
        if (bits) sp_factor_band<G>(c, c.KF(false), c.KD(false), S.dpUsed, [=](int) { return d2; }, [=](int r) { return ((bits[r >> 5] >> (r & 31)) & 1u) != 0u; });
        else sp_factor_band<G>(c, c.KF(false), c.KD(false), S.dpUsed, [=](int) { return d2; }, [=](int r) { return st[r] != ST_INACT; });
        if (level == 1) break;
        int badVar = 0, badRow = 0;
        GD Kd = c.KD(false);
        const double sc = c.info->scale, vmax = 1.0 / (1e-8 * sc), rmax = sc / 1e-8;
        const int Nb = db.N - db.kb;
#pragma unroll 4
        for (int p = t; p < Nb; p += G) {
            const double kd = Kd[p];            // 1 / D
            const int node = db.pnode[p];
            if (node < n) badVar |= !(kd > 0.0 && kd < vmax);
            else if (st[node - n] != ST_INACT) badRow |= !(kd < 0.0 && kd > -rmax);
        }
        const bool bv = g_any<G>(badVar), br = g_any<G>(badRow);
        if (!bv && !br) break;
        if (bv && t == 0) c.info->bigReg = 1;
        level = 1;
    }
    if (db.kb > 0) { sp_border_prepare<G>(c, false, [=](int r) { return st[r] != ST_INACT; }); S.borderTodo = db.kb; }
    g_map<G, 8>(m, t, [&](int r) { return st[r]; }, [&](int r, int v) { stf[r] = v; });
    if (t == 0) c.info->stfValid = 1;
    g_sync();
    S.fact_valid = 1;
    return PH_CORRECT;
}

// PH_CORRECT: [Q + delta I, Ea'; Ea, -delta2 I][dx; dy] = [r1; ba - Ea x], x += dx, y += dy; then the next trial (or the polish has run out of trials)
template <int G>
__device__ __forceinline__ int sp_ph_correct(SpCtx<G>& c, SpState& S)
{
    const SpBatch& db = *c.db;
    const int t = c.gl, n = db.n, m = db.m;
    GD x = c.V(NV_XT), r1 = c.V(NV_R1), yt = c.M(MV_YT), ex = c.M(MV_EX), b = c.Nv();
    GD l = c.M(MV_L), u = c.M(MV_U);
    GI st = c.I(MI_STT);
    const int* iperm = db.iperm;
    SPROF(c, SP_VECTORS);
    // (deep tiles: the band's lane groups are 8 lanes wide, a pass over n is n / (8 U) round trips)
    g_map<G, 16>(n, t, [&](int i) { return ID{iperm[i], r1[i]}; }, [&](int, ID v) { b[v.i] = v.a; });
    g_map<G, 6>(m, t, [&](int r) { return ID4{iperm[n + r], st[r], l[r], u[r], ex[r]}; },
                [&](int, ID4 v) { b[v.p] = (v.s != ST_INACT) ? ((v.s == ST_UPPER) ? v.hi : v.lo) - v.e : 0.0; });
    g_sync();
    SPROF(c, SP_RHS);
    // one call site of the band solve: first the columns of W = U inv(Bd) a fresh factorisation owes (none for a plain band), then b
    const int borderTodo = S.borderTodo;
    for (int jb = 0; jb <= borderTodo; jb++) {
        GD vec = b;
        if (jb < borderTodo) vec = sp_border_column<G>(c, false, jb);
        else if (borderTodo > 0) { const double d2 = S.d2Used; sp_border_schur<G>(c, false, S.dpUsed, [=](int) { return d2; }, [=](int r) { return st[r] != ST_INACT; }); }
        sp_solve_band<G>(c, false, vec);
    }
    S.borderTodo = 0;
    if (db.kb > 0) sp_border_solve<G>(c, false, b);
    double xm = 0.0;
    g_map<G, 12>(n, t, [&](int i) { return D2{b[iperm[i]], x[i]}; }, [&](int i, D2 v) { const double xn = v.b + v.a; x[i] = xn; xm = fmax(xm, fabs(xn)); });
    S.xinf = g_max<G>(xm);
    g_map<G, 10>(m, t, [&](int r) { return ID2{st[r], b[iperm[n + r]], yt[r]}; }, [&](int r, ID2 v) { if (v.s != ST_INACT) yt[r] = v.y + v.v; });
    g_sync();
    c.cCorr++;
    S.trial++;
    if (S.trial >= c.db->opt.maxTrials) return sp_polish_failed<G>(c, S);
    return PH_TRIAL;
}

// the subsolver call starts (oracle: sqp_solve, entry): SubsolverBase::solve on the OSQP arm.  A hot start from the stored solution goes
// straight to the polish; everything else takes the round preamble (PH_ROUND).
template <int G>
__device__ __forceinline__ int sp_qp_begin(SpCtx<G>& c, SpState& S, GD g)
{
    const SpBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = c.gl, n = db.n, m = db.m, initial = S.initial;
    GD xq = c.V(NV_XQ), xa = c.V(NV_XA), xt = c.V(NV_XT);
    GD yq = c.M(MV_YQ), ya = c.M(MV_YA), yt = c.M(MV_YT);
    GD l = c.M(MV_L), u = c.M(MV_U);
    GI st = c.I(MI_ST), stt = c.I(MI_STT);
    S.trials0 = c.cTrials; S.admm0 = c.cAdmm; S.qpIter = 0;
    S.n_admm = initial ? o.admmFirst : o.admmHot;
    S.use_stored = (!initial && c.info->haveSolution && S.n_admm == 0);
    // The ADMM iterate (xa, ya) starts as a copy of the stored solution.  A hot start hands the stored solution to the polish directly and
    // the copy is made only if the polish fails and an ADMM round follows (one QP in a thousand on the synthetic workload).
    S.backup_pending = 0;
    if (initial) {
        GD x0 = c.V(NV_X0), y0 = c.M(MV_Y0);
        const int hasY0 = c.info->hasY0;
        g_map<G, 8>(n, t, [&](int i) { return x0[i]; }, [&](int i, double v) { xq[i] = v; xa[i] = v; });
        g_map<G, 8>(m, t, [&](int r) { return y0[r]; }, [&](int r, double v) { const double yv = hasY0 ? -v : 0.0; yq[r] = yv; ya[r] = yv; });
    } else if (S.use_stored) {
        S.backup_pending = 1;
    } else {
        g_map<G, 8>(n, t, [&](int i) { return xq[i]; }, [&](int i, double v) { xa[i] = v; });
        g_map<G, 8>(m, t, [&](int r) { return yq[r]; }, [&](int r, double v) { ya[r] = v; });
    }
    g_sync();
    S.admm_ready = 0; S.round = 0;
    if (!S.use_stored) return PH_ROUND;
    g_map<G, 4>(m, t, [&](int r) { return ID3{st[r], l[r], u[r], 0.0, yq[r]}; },
                [&](int r, ID3 v) { const int s = (v.lo == v.hi) ? ST_EQ : v.s; stt[r] = s; yt[r] = (s != ST_INACT) ? v.y : 0.0; });
    // (xt is xq already: the stored solution is the accepted trial vector of the QP before, copied from xt in sp_ph_qpend, and nothing has
    // written xt since -- use_stored is only set behind that copy)
    g_sync();
    return sp_polish_begin<G>(c, S, g, 1);
}

// PH_ROUND: the preamble of a round that does not start from the stored solution -- the first QP of a homotopy, and every round after a
// polish that failed: ADMM iterations from (xa, ya), the working set they propose, the polish from there
template <int G>
__device__ __forceinline__ int sp_ph_round(SpCtx<G>& c, SpState& S, GD g)
{
    const SpBatch& db = *c.db;
    const int t = c.gl, n = db.n, m = db.m;
    GD xq = c.V(NV_XQ), xa = c.V(NV_XA), xt = c.V(NV_XT);
    GD yq = c.M(MV_YQ), ya = c.M(MV_YA), za = c.M(MV_ZA), yt = c.M(MV_YT);
    GD l = c.M(MV_L), u = c.M(MV_U);
    GI stt = c.I(MI_STT);
    if (S.backup_pending && S.round > 0) {
        g_map<G, 8>(n, t, [&](int i) { return xq[i]; }, [&](int i, double v) { xa[i] = v; });
        g_map<G, 8>(m, t, [&](int r) { return yq[r]; }, [&](int r, double v) { ya[r] = v; });
        g_sync();
        S.backup_pending = 0;
    }
    if (!S.admm_ready) {
        sp_Ex<G>(c, xa, za);
        g_map<G, 8>(m, t, [&](int r) { return D3{za[r], l[r], u[r]}; },
                    [&](int r, D3 v) { za[r] = fmin(fmax(v.a, v.b), v.c); if (isinf(v.b) && isinf(v.c)) ya[r] = 0.0; });
        g_sync();
        S.admm_ready = 1;
    }
    if (S.n_admm > 0) sp_admm<G>(c, g, S.n_admm);
    g_map<G, 4>(m, t, [&](int r) { return ID3{0, l[r], u[r], za[r], ya[r]}; },
                [&](int r, ID3 v) {
                    int s = ST_INACT;
                    if (isfinite(v.lo) && (v.z - v.lo < -v.y)) s = ST_LOWER;
                    if (isfinite(v.hi) && (v.hi - v.z < v.y)) s = ST_UPPER;
                    if (v.lo == v.hi) s = ST_EQ;
                    stt[r] = s;
                    yt[r] = (s != ST_INACT) ? v.y : 0.0;
                });
    g_map<G, 8>(n, t, [&](int i) { return xa[i]; }, [&](int i, double v) { xt[i] = v; });
    g_sync();
    return sp_polish_begin<G>(c, S, g, 0);
}
// ---- LCQProblem::runSolver, OSQP_SPARSE arm (oracle: orc_sparse_lcqp_solve) ----------------------------------------------------
// PH_START: everything in front of the first QP
template <int G>
__device__ __forceinline__ int sp_ph_start(SpCtx<G>& c, SpState& S)
{
    const SpBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = c.gl, n = db.n;
    GD g = c.V(NV_G), gtil = c.V(NV_GTIL), xk = c.V(NV_XK), gk = c.V(NV_GK);
    GD Qx = c.V(NV_QX), Cx = c.V(NV_CX), Qp = c.V(NV_QP), Cp = c.V(NV_CP);
    memset(&S.st, 0, sizeof(S.st));
    S.rc = 0; S.qpIter = 0; S.histLen = 0; S.algoStat = 0; S.totalIter = 0;
    S.alphak = 1.0; S.rho = o.initialPenaltyParameter;
    S.perturbCounter = 0;
    if (db.traceCap > 0 && t == 0) db.traceLen[c.b] = 0;     // a run that records nothing leaves an empty trace
    { GD x0 = c.V(NV_X0);
      g_map<G, 8>(n, t, [&](int i) { return D2{x0[i], g[i]}; }, [&](int i, D2 v) { xk[i] = v.a; gtil[i] = v.b; }); }
    g_sync();
    // Q x0 and C x0 once; from here on both follow the steps (sp_ph_qpend)
    sp_Qx2<G>(c, xk, xk, Qx, Qp); sp_Cx2<G>(c, xk, xk, Cx, Cp);
    c.bytes += db.by[BY_START];
    if (o.solveZeroPenaltyFirst) { for (int i = t; i < n; i += G) gk[i] = g[i]; g_sync(); }
    else { const double rho = S.rho; for (int i = t; i < n; i += G) gk[i] = rho * Cx[i] + gtil[i]; g_sync(); }
    S.initial = 1;
    S.gmaxNext = -1.0;      // max |gk| of the next QP when this level has formed it (-1: the subsolver looks)
    return sp_qp_begin<G>(c, S, gk);
}

// PH_QPEND: the subsolver has a verified solution (oracle: the exit of sqp_solve), then one iterate of runSolver's loop up to the next QP
template <int G>
__device__ __forceinline__ int sp_ph_qpend(SpCtx<G>& c, SpState& S)
{
    const SpBatch& db = *c.db;
    const lcqp_options_t& o = db.opt;
    const int t = c.gl, n = db.n, m = db.m, nC = db.nC, nK = db.nComp;
    GD g = c.V(NV_G), gtil = c.V(NV_GTIL), gphi = c.V(NV_GPHI), xk = c.V(NV_XK), pk = c.V(NV_PK), xnew = c.V(NV_XNEW), gk = c.V(NV_GK);
    GD Qx = c.V(NV_QX), Cx = c.V(NV_CX), Qp = c.V(NV_QP), Cp = c.V(NV_CP);
    GD yk = c.M(MV_YK), lx = c.M(MV_LX);
    const bool hasPhi = db.hasLbL || db.hasLbR;
    const double phiConst = c.info->phiConst;
    double* hist = c.info->hist;
    {   // the solution becomes the stored one (sqp_solve's exit)
        GD xq = c.V(NV_XQ), xt = c.V(NV_XT), yq = c.M(MV_YQ), yt = c.M(MV_YT);
        GI st = c.I(MI_ST), stt = c.I(MI_STT);
        S.qpIter = (c.cTrials - S.trials0) + (c.cAdmm - S.admm0);
        // (x: in the pass below that forms pk)
        g_map<G, 8>(m, t, [&](int r) { return ID{stt[r], yt[r]}; }, [&](int r, ID v) { yq[r] = v.a; st[r] = v.i; });
        if (t == 0) c.info->haveSolution = 1;
        g_sync();
    }
    SPROF(c, SP_VECTORS);
    S.st.subproblemIter += S.qpIter; S.st.qpSolverExitFlag = 0; S.st.qpSolves++;
    double rho = S.rho, alphak = S.alphak;
    auto updatePenalty = [&]() {
        if (o.nDynamicPenalty > 0) S.histLen = 0;
        rho *= o.penaltyUpdateFactor;
        S.st.rhoOpt = rho;
        if (hasPhi) { for (int i = t; i < n; i += G) gtil[i] = g[i] + rho * gphi[i]; g_sync(); }
    };
    // What the subsolver's accepted trial leaves behind makes every product of this level but one unnecessary (the dense kernel does
    // the same, lcqp_dev.hpp: lcqp_run): Q xq is in NV_TMP (sp_residual), E xq in MV_EX, and its residual r1 = -gk - Q xq - E'yq (NV_R1, gk still the vector of that QP)
    // gives E'yq.  So pk = xq - xk, Q pk = Q xq - Q xk with Q xk kept up to date below, C xq = L'(R xq) + R'(L xq) is one column
    // gather over E with the entries of E xq, C pk = C xq - C xk, and the stationarity needs no pass over E' of its own.
    // (round 2: one pass over Q, one over E, two over E' per iterate.)
    GD qxs = c.V(NV_TMP), exs = c.M(MV_EX), r1s = c.V(NV_R1), gs0 = gk;
    {
        GD xq = c.V(NV_XQ), xt = c.V(NV_XT), yq = c.M(MV_YQ);
        g_map<G, 8>(n, t, [&](int i) { return D4{xt[i], xk[i], qxs[i], Qx[i]}; }, [&](int i, D4 v) { xq[i] = v.a; xnew[i] = v.a; pk[i] = v.a - v.b; Qp[i] = v.c - v.d; });
        g_map<G, 8>(m, t, [&](int r) { return yq[r]; }, [&](int r, double v) { yk[r] = -v; });     // src/SubsolverOSQP.cpp:196-199
        g_sync();
    }
    const int initial = S.initial;
    bool perturbed = false;
    if (initial) S.st.rhoOpt = rho;
    else if (o.perturbStep) {
        perturbed = true;
        const uint64_t pc = S.perturbCounter;
        for (int i = t; i < n; i += G) {
            uint64_t z = o.perturbSeed + (pc + (uint64_t)i + 1ULL) * opaque_u64(0x9E3779B97F4A7C15ULL);
            z = (z ^ (z >> 30)) * opaque_u64(0xBF58476D1CE4E5B9ULL); z = (z ^ (z >> 27)) * opaque_u64(0x94D049BB133111EBULL); z = z ^ (z >> 31);
            xk[i] += ((int)(z % 3ULL) - 1) * 2.221e-16;
        }
        S.perturbCounter += (uint64_t)n;
        g_sync();
    }
    double sq = 0.0, sl = 0.0;      // pk'(Q + rho C) pk and pk'((Q + rho C) xk + g~)
    if (perturbed) {
        // the perturbation has to reach the penalty gradient rho C xk -- it is there to break the symmetry of problems like warm_up
        // (perturbStep :1353-1362) -- so C xk is taken from the perturbed xk: one more pass over E, the column gather carries two
        // vectors.  (Q xk is not: 2.2e-16 per component is below its rounding; the dense kernel does the same.)
        sp_Ex<G>(c, xk, lx);
        sp_C_from_Ex<G, true>(c, exs, lx, [](int) { return NoPre{}; }, [&](int i, double cxq, double cxk, NoPre) { Cx[i] = cxk; Cp[i] = cxq - cxk; });
        c.bytes += 3.0 * db.by[BY_E];
    } else {
        // C pk and, in the same pass, the two sums of the step length (round 5: they were a pass of their own over pk, Qp, Cp, Qx, Cx, gtil)
        struct D5 { double cx, pk, qp, qx, gt; };
        sp_C_from_Ex<G, false>(c, exs, exs, [&](int i) { return D5{Cx[i], pk[i], Qp[i], Qx[i], gtil[i]}; },
                               [&](int i, double cxq, double, D5 v) {
                                   const double cp = cxq - v.cx;
                                   Cp[i] = cp;
                                   sq += v.pk * (v.qp + rho * cp); sl += v.pk * ((v.qx + rho * v.cx) + v.gt);
                               });
        c.bytes += db.by[BY_E];
    }
    if (!initial) {
        if (perturbed) {
#pragma unroll 4
            for (int i = t; i < n; i += G) { sq += pk[i] * (Qp[i] + rho * Cp[i]); sl += pk[i] * ((Qx[i] + rho * Cx[i]) + gtil[i]); }
        }
        const double qk = g_sum<G>(sq), lk = g_sum<G>(sl);
        alphak = 1.0;
        if (qk > 0 && lk < 0) alphak = fmin(-lk / qk, 1.0);
    }
    S.initial = 0;
    // the step, the products that follow it, and updateStationarity without a box term: statk = Qk xk + g_tilde - E'yk with
    // E'yk = -E'yq = gs0 + Q xq + r1s
    double statMax = 0.0, phiSum = 0.0;
    { struct D10 { double a, b, c, d, e, f, g0, q, r, gt, gp; };
      g_map<G, 3>(n, t, [&](int i) { return D10{xk[i], pk[i], Qx[i], Qp[i], Cx[i], Cp[i], gs0[i], qxs[i], r1s[i], gtil[i], hasPhi ? (double)gphi[i] : 0.0}; },
                  [&](int i, D10 v) {
                      const double qn = v.c + alphak * v.d, cn = v.e + alphak * v.f, xn = v.a + alphak * v.b;
                      xk[i] = xn; Qx[i] = qn; Cx[i] = cn;
                      statMax = nmax(statMax, fabs(((qn + rho * cn) + v.gt) - ((v.g0 + v.q) + v.r)));
                      phiSum += (hasPhi ? v.gp * xn : 0.0) + 0.5 * xn * cn;      // getPhi of the new iterate, in its order of summation
                  }); }
    g_sync();
    const double statInf = g_max<G>(statMax);
    const double phiStep = phiConst + g_sum<G>(phiSum);      // (xk and C xk do not change again in this iterate: every getPhi below is this value)
    int totalIter = S.totalIter;
    if (db.traceCap > 0 && totalIter < db.traceCap) {   // storeSteps :488-490, printIteration :1528-1576 (the host rebuilds both from this)
        const double phiNow = phiStep;
        double so = 0.0, sm = 0.0, pm = 0.0;
        for (int i = t; i < n; i += G) { const double xv = xk[i]; so += g[i] * xv + 0.5 * xv * Qx[i]; sm += 0.5 * rho * xv * Cx[i]; pm = fmax(pm, fabs(pk[i])); }
        const double objNow = g_sum<G>(so), meritNow = objNow + g_sum<G>(sm), stepNow = g_max<G>(pm);
        double* ts = db.traceS + ((size_t)c.b * db.traceCap + totalIter) * 8;
        double* tx = db.traceX + ((size_t)c.b * db.traceCap + totalIter) * n;
        if (t == 0) {
            ts[0] = statInf; ts[1] = phiNow; ts[2] = rho; ts[3] = alphak; ts[4] = objNow; ts[5] = meritNow; ts[6] = stepNow; ts[7] = (double)S.qpIter;
            db.traceLen[c.b] = totalIter + 1;
        }
        for (int i = t; i < n; i += G) tx[i] = xk[i];
    }
    totalIter++; S.totalIter = totalIter; S.st.iterTotal++;
    bool leyffer = false;
    const int nd = o.nDynamicPenalty;
    if (nd > 0) {
        const double cur = phiStep;
        if (S.histLen < nd) { if (t == 0) hist[S.histLen] = cur; S.histLen++; g_sync(); }
        else {
            if (!(cur < o.complementarityTolerance)) {
                leyffer = true;
                for (int i = 0; i < nd; i++) if (cur < o.etaDynamicPenalty * hist[i]) { leyffer = false; break; }
            }
            g_sync();
            if (t == 0) { for (int i = 0; i + 1 < nd; i++) hist[i] = hist[i + 1]; hist[nd - 1] = cur; }
            g_sync();
        }
    }
    if (leyffer) { updatePenalty(); S.st.iterOuter++; }
    bool done = false;
    if (statInf < o.stationarityTolerance) {
        if (phiStep < o.complementarityTolerance) {
            sp_Ex<G>(c, xk, lx);
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
            S.algoStat = wflag ? 1 : (sflag ? 4 : (mflag ? 3 : 2));
            g_sync();
            for (int i = t; i < nK; i += G) { const double Lx = lx[nC + i], Rx = lx[nC + nK + i]; yk[nC + i] -= rho * Rx; yk[nC + nK + i] -= rho * Lx; }
            g_sync();
            S.rc = 0;
            done = true;
        } else {
            updatePenalty(); S.st.iterOuter++;
        }
    }
    if (!done && totalIter > o.maxIterations) { S.rc = LCQP_MAX_ITERATIONS_REACHED; done = true; }
    if (!done && rho > o.maxPenaltyParameter) { S.rc = LCQP_MAX_PENALTY_REACHED; done = true; }
    S.rho = rho; S.alphak = alphak;
    if (done) return sp_finish<G>(c, S);
    // the next QP's linear term; its hot start needs r1 = r1_last + (g_last - g): the residual of the accepted trial is still in NV_R1
    { GD r1 = c.V(NV_R1);
      double gm = 0.0;
      g_map<G, 8>(n, t, [&](int i) { return D4{Cx[i], gtil[i], gk[i], r1[i]}; },
                  [&](int i, D4 v) { const double gn = rho * v.a + v.b; gk[i] = gn; r1[i] = v.d + (v.c - gn); gm = fmax(gm, fabs(gn)); });
      S.gmaxNext = g_max<G>(gm); }
    g_sync();
    SPROF(c, SP_LCQP);
    return sp_qp_begin<G>(c, S, gk);
}


template <int G>
__device__ __forceinline__ SpCtx<G> sp_ctx(const SpBatch& db, int b, int w0, int lane)
{
    SpCtx<G> c;
    c.db = &db; c.b = b; c.gl = lane & (G - 1); c.gi = (unsigned)(b - w0); c.w0 = w0;
    c.info = db.info + b;
    c.win = sp_dyn_lds + (size_t)(lane / G) * (G * G + 16 * G);
    c.cAdmm = c.cTrials = c.cFact = c.cCorr = c.cSweeps = 0;
    c.bytes = 0.0;
#ifdef LCQP_PROFILE
    for (int k = 0; k < SP_NPHASE; k++) c.prof[k] = 0;
    c.tprev = __builtin_amdgcn_s_memtime();
#endif
    return c;
}

// ---- setup: scales, rho vector, phi expressions, the ONE factorisation of the ADMM KKT matrix ----------------------------------
template <int G>
__global__ __launch_bounds__(WGS) void k_sparse_setup(SpBatch db)
{
    const int b = blockIdx.x * (64 / G) + threadIdx.x / G;
    if (b >= db.B) return;
    SpCtx<G> c = sp_ctx<G>(db, b, blockIdx.x * (64 / G), (int)threadIdx.x);
    const int t = c.gl, n = db.n, m = db.m, nC = db.nC, nK = db.nComp;
#ifdef LCQP_PROFILE
    if (t == 0) for (int k = 0; k < SP_NPHASE; k++) c.info->prof[k] = 0.0;
#endif
    double dmax = 0.0;
    for (int i = t; i < n; i += G)
        for (int k = db.Qp[i]; k < db.Qp[i + 1]; k++) if (db.Qi[k] == i) dmax = fmax(dmax, fabs(c.Qx()[k]));
    double scale = g_max<G>(dmax);
    if (!(scale > 1e-300)) scale = 1.0;
    const double rho = db.opt.admmRho * scale;
    GD l = c.M(MV_L), u = c.M(MV_U), rhov = c.M(MV_RHOV);
    for (int r = t; r < m; r += G) {
        double rv = rho;
        if (isinf(l[r]) && isinf(u[r])) rv = 1e-6 * rho;
        else if (l[r] == u[r]) rv = rho * db.opt.rhoEqMult;
        rhov[r] = rv;
    }
    // phi expressions (src/LCQProblem.cpp:969-996)
    double phiConst = 0.0;
    GD gphi = c.V(NV_GPHI);
    if (db.hasLbL || db.hasLbR) {
        GD lbL = c.arr(db.lbL, nK), lbR = c.arr(db.lbR, nK);
        double s = 0.0;
        for (int i = t; i < nK; i += G) s += lbL[i] * lbR[i];
        phiConst = g_sum<G>(s);
        GD coef = c.M(MV_LX);
        for (int r = t; r < m; r += G) coef[r] = (r >= nC + nK) ? lbL[r - nC - nK] : ((r >= nC) ? lbR[r - nC] : 0.0);     // R'lbL + L'lbR
        g_sync();
        sp_ETy<G>(c, coef, gphi, [](int) { return 0.0; }, [](double v) { return v; });
    } else {
        for (int i = t; i < n; i += G) gphi[i] = 0.0;
    }
    double e1 = 0.0;
    {
        GD Ev = c.Ex();
        for (int r = t; r < m; r += G) { double s1 = 0.0; for (int k = db.Ep[r]; k < db.Ep[r + 1]; k++) s1 += fabs(Ev[k]); e1 = fmax(e1, s1); }
        e1 = g_max<G>(e1);
    }
    if (t == 0) {
        c.info->e1max = e1;
        c.info->scale = scale; c.info->sigma = db.opt.admmSigma * scale; c.info->delta = db.opt.proxBig * scale; c.info->delta2 = 1e-9 / scale;
        c.info->phiConst = phiConst; c.info->haveSolution = 0; c.info->stfValid = 0; c.info->bytes = 0.0;
        c.info->deltaS = db.opt.proxSmall * scale; c.info->delta2S = 1e-14 / scale; c.info->bigReg = 0;
    }
    g_sync();
    if constexpr (G <= 16) {
        // the assembled band rows every factorisation of this instance streams (sp_factor_reg): the one gather from the values of Q and E
        GD K0 = c.K0(), Qv = c.Qx(), Ev = c.Ex();
        const int nnzQ = db.nnzQ;
#pragma unroll 2
        for (int r = t; r < db.N; r += G) {      // upper form: entry k of row r is K[r + k][r] (sp_factor_reg), the diagonal in entry 0
#pragma unroll
            for (int k = 1; k < G; k++) {
                const int code = db.bsrc[r * G + k];
                K0[r * G + k] = (code >= nnzQ) ? (double)Ev[code - nnzQ] : ((code >= 0) ? (double)Qv[code] : 0.0);
            }
            const int bd = db.bdiag[r];
            K0[r * G] = (bd >= 0) ? (double)Qv[bd] : 0.0;
        }
        g_sync();
    }
    sp_factor_band<G>(c, c.KF(true), c.KD(true), db.opt.admmSigma * scale, [=](int r) { return 1.0 / rhov[r]; }, [](int) { return true; });
    if (db.kb > 0) {
        sp_border_prepare<G>(c, true, [](int) { return true; });
        for (int jb = 0; jb < db.kb; jb++) { GD vec = sp_border_column<G>(c, true, jb); sp_solve_band<G>(c, true, vec); }
        sp_border_schur<G>(c, true, db.opt.admmSigma * scale, [=](int r) { return 1.0 / rhov[r]; }, [](int) { return true; });
    }
    if (t == 0) c.info->bytes = c.bytes;
}

// ---- the scheduler: persistent wavefronts that serve the phase queues of their pool -------------------------------------------------------
// Queue discipline (per pool and phase): push = take a position (atomicAdd on tail), wait until its slot's sequence says "free for this
// position" (at once, unless the ring has been lapped), store (position + 1, instance id) into it, atomicAdd on count; pop = claim entries of
// count (compare-and-swap), take as many positions (atomicAdd on head), wait for each slot's sequence to say "holds this position" (its pusher
// writes it a few instructions after taking the position), acquire fence, hand the slot on (store (position + poolSize, -)).  An instance is in at most one queue, so a ring of poolSize entries never overflows.
// Everything an instance's phase wrote to memory is released by the fence in front of its push and acquired by the fence behind the pop of
// whichever wavefront runs its next phase.
// agent scope: a pool is served by wavefronts of several XCDs, whose L2s are not coherent with each other -- the release writes the L2 back, the
// acquire invalidates L1 and L2 (workgroup-scope fences in their place: stale vectors, more iterates, 15 - 25 % SLOWER; profiles/round4)
#ifdef SP_XCD_POOLS
// EXPERIMENT (round 5, -DSP_XCD_POOLS; needs a pool count that is a multiple of 8, LCQP_SPARSE_POOL): a pool is served only by wavefronts of
// ONE XCD (chosen by the XCC id the wavefront reads from its hardware register: wavefronts do not migrate), so everything an instance's
// phases read and write goes through one L2 and the hand-over needs no L2 write-back and no L2 invalidation: the release waits for the
// stores to reach L2 (the vector L1 is write-through), the acquire invalidates the vector L1 only.  Measured (profiles/round5/
// sparse_xcd_pools_light_fences_dropped.log): same results, 10 400 - 10 800 against 12 700 LCQPs/s at B = 65 536, 73 against 11 000 at
// B = 16 384 (one pool per XCD: the wavefronts of an XCD all wait on one pool's queues).  The L2 invalidations are not where the traffic is.
#define SP_ACQUIRE() do { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); asm volatile("buffer_inv sc0" ::: "memory"); } while (0)
#define SP_RELEASE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define SP_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define SP_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#endif
__device__ __forceinline__ int q_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long q_load64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void q_store64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long q_slot(int seq, int id) { return ((unsigned long long)(unsigned)seq << 32) | (unsigned)id; }

// Lanes per instance depend on the phase (round 4, second step): the band phases (PH_FACTOR, PH_CORRECT, PH_ROUND with its ADMM band solves) need
// the G lanes the folded band layout is made for and take 64 / G instances per wavefront; the STREAMING phases (PH_START, PH_TRIAL, PH_QPEND:
// sparse products and element-wise loops, 60 % of the time) take ONE instance with all 64 lanes -- a wavefront instruction then reads 512
// contiguous bytes of one instance instead of eight 64-byte pieces of eight instances, the access pattern this part streams best
// (tools/micro/stream_pattern.py: 2.4 TB/s for eight far-apart pieces per wavefront, 5.8 TB/s for one stream).
#ifndef SP_WIDE_LANES
#define SP_WIDE_LANES 64      // (experiment switch: 0 = every phase with G lanes per instance, the first phase machine of round 4)
#endif
#ifndef SP_CHAIN
#define SP_CHAIN 1            // (experiment switch: 0 = every phase change goes through the queues)
#endif
#ifndef SP_WIDE_BATCH
#define SP_WIDE_BATCH 16      // passes of a streaming step: instances popped at once = SP_WIDE_BATCH * 64 / SP_WIDE_LANES
#endif
#ifndef SP_CHAIN_BAND
#define SP_CHAIN_BAND 1
#endif
#ifndef SP_BAND_BATCH
#define SP_BAND_BATCH 1       // passes of a band step
#endif
static_assert(SP_WIDE_BATCH * (SP_WIDE_LANES ? 64 / SP_WIDE_LANES : 8) <= 64 && SP_BAND_BATCH <= 8, "a step holds its instances one per lane");
template <int G, int GP>
__device__ __forceinline__ int sp_run_phase(const SpBatch& db, int ph, int b, int w0, int lane)
{
    SpCtx<GP> c = sp_ctx<GP>(db, b, w0, lane);
    // INVARIANT (lock step): the record lives in global memory and EVERY lane of the instance's group reads and writes it -- the same
    // values, in the same instruction (S.trial++, S.round++, S.nrefine++ ...: a load and a store the group issues together).  That is
    // well defined only because (1) every branch that leads to an access of S is uniform inside the group (conditions are group
    // reductions or values read from S itself), (2) the group's lanes are lanes of ONE wavefront (G <= 64), so a load and the store
    // behind it are not separated by another lane's store, and (3) no other group touches this record while the instance is in a phase
    // (an instance is in at most one queue; the hand-over is ordered by the queue's fences).  A phase routine that accesses S under a
    // condition that differs between the lanes of a group breaks this.  (A private copy in registers, written back by lane 0, would
    // cost 50 VGPRs of the 249 the kernel has: the band factorisation's register window takes the rest.)
    SpState& S = db.state[b];
    c.cAdmm = S.cAdmm; c.cTrials = S.cTrials; c.cFact = S.cFact; c.cCorr = S.cCorr; c.cSweeps = S.cSweeps; c.bytes = S.bytes;
    GD gk = c.V(NV_GK);
    int next;
    if constexpr (GP == G) {
        // every phase can run with the band's lane group (the streaming ones do when SP_WIDE_LANES == 0 or G == 64)
        switch (ph) {
            case PH_START:   c.cAdmm = c.cTrials = c.cFact = c.cCorr = c.cSweeps = 0; c.bytes = 0.0; next = sp_ph_start<GP>(c, S); break;
            case PH_ROUND:   next = sp_ph_round<GP>(c, S, gk); break;
            case PH_TRIAL:   next = sp_ph_trial<GP>(c, S, gk); break;
            case PH_FACTOR:  next = sp_ph_factor<GP>(c, S); break;
            case PH_CORRECT: next = sp_ph_correct<GP>(c, S); break;
            default:         next = sp_ph_qpend<GP>(c, S); break;
        }
    } else {
        switch (ph) {
            case PH_START:   c.cAdmm = c.cTrials = c.cFact = c.cCorr = c.cSweeps = 0; c.bytes = 0.0; next = sp_ph_start<GP>(c, S); break;
            case PH_TRIAL:   next = sp_ph_trial<GP>(c, S, gk); break;
            default:         next = sp_ph_qpend<GP>(c, S); break;
        }
    }
    S.cAdmm = c.cAdmm; S.cTrials = c.cTrials; S.cFact = c.cFact; S.cCorr = c.cCorr; S.cSweeps = c.cSweeps; S.bytes = c.bytes;
#ifdef LCQP_PROFILE
    SPROF(c, SP_VECTORS);
    if (c.gl == 0) for (int k = 0; k < SP_NPHASE; k++) c.info->prof[k] += (double)c.prof[k];
#endif
    return next;
}

template <int G>
__global__ __launch_bounds__(WGS, (G <= 8 ? SP_WAVES_PER_SIMD : 1)) void k_sparse_sched(SpBatch db)
{
    constexpr int IPW = 64 / G;
    constexpr int GW = (SP_WIDE_LANES > G) ? SP_WIDE_LANES : G;      // lanes per instance of the streaming phases
    constexpr int IPWW = 64 / GW;
    __shared__ int s_id[WGS], s_next[WGS], s_ctl[2];      // the step's instances, where each goes next; [0] how many they are, [1] polls without work in a row
    // the pool's queues, derived afresh in front of the pop and in front of the push: nothing of the scheduler is alive across a phase
    struct Q { int w0, mask; unsigned long long* ring; int* ctl; int* remaining; };
    auto queues = [&]() {
#ifdef SP_XCD_POOLS
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        int pool = (int)(xcc & 7u) + 8 * (int)((blockIdx.x >> 3) % (unsigned)(db.nPools >> 3));
#else
        int pool = blockIdx.x % db.nPools;
#endif
        asm volatile("" : "+s"(pool));
        Q q;
        q.w0 = pool * db.poolSize; q.mask = db.poolSize - 1;
        q.ring = db.qring + (size_t)pool * PH_NUM * db.poolSize;
        q.ctl = db.qctl + (size_t)pool * (PH_NUM + 1) * QCTL;
        q.remaining = q.ctl + PH_NUM * QCTL;
        return q;
    };
    if (threadIdx.x == 0) s_ctl[1] = 0;
    for (;;) {
        const int lane = here((int)threadIdx.x);      // per step: nothing derived from the lane number is carried around the loop (it would be hoisted and spilled)
        const Q q0 = queues();
        const int w0 = q0.w0, mask = q0.mask;
        unsigned long long* ring = q0.ring;
        int* ctl = q0.ctl;
        int* remaining = q0.remaining;
        // ---- pop: one phase for the whole wavefront (lane 0 decides).  A band phase with a full wavefront's worth of instances first (the long
        // steps run at full width), else the fullest streaming queue, else whatever a band queue holds.
        int ph = -1, take = 0, base = 0;
        if (lane == 0) {
            int cnt[PH_NUM];
            for (int k = 0; k < PH_NUM; k++) cnt[k] = q_load(&ctl[k * QCTL + 2]);
            auto isWide = [](int k) { return GW != G && (k == PH_START || k == PH_TRIAL || k == PH_QPEND); };
            int best = 0;
            for (int k = 0; k < PH_NUM; k++) if (!isWide(k) && cnt[k] >= IPW && cnt[k] > best) { best = cnt[k]; ph = k; }
            if (ph < 0) for (int k = 0; k < PH_NUM; k++) if (isWide(k) && cnt[k] > best) { best = cnt[k]; ph = k; }
            if (ph < 0) for (int k = 0; k < PH_NUM; k++) if (cnt[k] > best) { best = cnt[k]; ph = k; }
            if (ph >= 0) {
                // instances of a streaming step: they run one after the other on this wavefront, so a long step makes the last of them wait
                // for the others while wavefronts elsewhere poll -- P = unfinished instances per SIMD of the machine up to 4, P / 2 beyond,
                // at most SP_WIDE_BATCH (the fences of a step are paid once for all of them: full steps for batches that fill the
                // machine).  profiles/round5/sparse_wide_batch_rule.log: against 16 per step +20 % at B = 1024, +15 % at 2048, +11 % at
                // 4096, +7 % at 8192, the same from 16 384 on
                int wb = SP_WIDE_BATCH;
                if (isWide(ph)) { const int P = q_load(remaining) / db.wideDiv; wb = max(1, min(SP_WIDE_BATCH, P <= 4 ? P : max(4, P >> 1))); }
                take = min(best, isWide(ph) ? IPWW * wb : IPW * SP_BAND_BATCH);
                if (atomicCAS(&ctl[ph * QCTL + 2], best, best - take) == best) base = atomicAdd(&ctl[ph * QCTL + 1], take);
                else { ph = -1; take = 0; }      // somebody else moved the counter: look again
            }
        }
        ph = __builtin_amdgcn_readfirstlane(ph); take = __builtin_amdgcn_readfirstlane(take); base = __builtin_amdgcn_readfirstlane(base);
#ifdef LCQP_SCHED_PROFILE
        const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
        if (take == 0) {
            if (q_load(remaining) <= 0) break;
            const int idle = s_ctl[1] + 1;
            if (lane == 0) s_ctl[1] = idle;
            // back off: a wavefront that finds nothing polls again later and later (each poll reads the pool's counters through the L2 all
            // wavefronts of the pool share; with s_sleep(32) per poll a quarter of idle wavefronts halved the rate of the working ones:
            // profiles/round5/sparse_waves.log) -- 2^min(idle, SP_BACKOFF_MAX) / 8 sleeps of 127 x 64 clocks, at most ~ 60 us
#ifndef SP_BACKOFF_MAX
#define SP_BACKOFF_MAX 7
#endif
            if (idle > 2) { const int reps = (1 << min(idle, SP_BACKOFF_MAX)) >> 3; for (int k = 0; k < max(reps, 1); k++) __builtin_amdgcn_s_sleep(127); }
#ifdef LCQP_SCHED_PROFILE
            if (lane == 0) { atomicAdd(&db.qprof[PH_NUM * 3 + 0], __builtin_amdgcn_s_memtime() - tq0); atomicAdd(&db.qprof[PH_NUM * 3 + 1], 1ull); }
#endif
            continue;
        }
        if (lane == 0) { s_ctl[1] = 0; s_ctl[0] = take; }
        int myid = -1;
        const size_t qoff = (size_t)ph * db.poolSize;
        if (lane < take) {
            const int pos = base + lane;
            unsigned long long e;
            while ((int)((e = q_load64(&ring[qoff + (pos & mask)])) >> 32) != pos + 1) __builtin_amdgcn_s_sleep(1);
            myid = (int)(unsigned)e;
        }
        SP_ACQUIRE();                                                                    // (also orders the slot's read in front of handing it on)
        if (lane < take) {
            const int pos = base + lane;
            q_store64(&ring[qoff + (pos & mask)], q_slot(pos + db.poolSize, -1));
        }
        const bool wide = GW != G && (ph == PH_START || ph == PH_TRIAL || ph == PH_QPEND);
        // ---- run the phase for the instances popped (lane groups without one idle through it).  The instances of the step and where each goes
        // next are kept in LDS, one per lane: no register of the scheduler is alive across a phase.
        s_id[lane] = myid;
        __builtin_amdgcn_wave_barrier();
        if (wide) {
            // streaming phases: the instances popped (up to SP_WIDE_BATCH) one after the other, GW lanes each, so that the fences of the step
            // -- the release writes back the whole XCD's L2 -- are paid once for all of them.  A streaming phase that leads to another one
            // (an accepted trial -> the end of its QP -> the first trial of the next) stays here: same lanes, same instance, its own stores in
            // program order -- no queue, no fences.
            for (int j = 0; j < take; j += IPWW) {
                const int ln = here(lane), bj = s_id[j + ln / GW];
                int nj = -1;
                if (bj >= 0)
                    for (int p = ph;; p = nj) {
                        nj = sp_run_phase<G, GW>(db, p, bj, w0, ln);
                        if (SP_CHAIN == 0 || !(nj == PH_TRIAL || nj == PH_QPEND)) break;
                    }
                if ((ln & (GW - 1)) == 0) s_next[j + ln / GW] = nj;
            }
        } else {
            // band phases: 64 / G instances side by side, SP_BAND_BATCH such passes per step
            for (int j = 0; j < take; j += IPW) {
                const int l0 = here(lane), bj = s_id[j + l0 / G];
                int nj = -1;
                if (bj >= 0)
                    for (int p = ph;; p = nj) {      // (a factorisation is always followed by its correction: same lanes, same instances)
                        nj = sp_run_phase<G, G>(db, p, here(bj), w0, here(l0));      // (re-laundered per pass: nothing of the instance's addressing is carried around the loop)
                        if (SP_CHAIN_BAND == 0 || !(p == PH_FACTOR && nj == PH_CORRECT)) break;
                    }
                if ((l0 & (G - 1)) == 0) s_next[j + l0 / G] = nj;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- push: what this wavefront wrote is released, then every instance goes to the queue of its next phase
        SP_RELEASE();
        if (here((int)threadIdx.x) < s_ctl[0]) {
            const Q q1 = queues();
            const int b = s_id[threadIdx.x], next = s_next[threadIdx.x];
            if (next == PH_NUM) atomicSub(q1.remaining, 1);
            else {
                const int pos = atomicAdd(&q1.ctl[next * QCTL + 0], 1), sl = pos & q1.mask;
                const size_t noff = (size_t)next * db.poolSize;
                while ((int)(q_load64(&q1.ring[noff + sl]) >> 32) != pos) __builtin_amdgcn_s_sleep(1);      // (waits only when the ring has been lapped onto a slot that is still being read)
                q_store64(&q1.ring[noff + sl], q_slot(pos + 1, b));
                atomicAdd(&q1.ctl[next * QCTL + 2], 1);
            }
        }
#ifdef LCQP_SCHED_PROFILE
        if (lane == 0) { atomicAdd(&db.qprof[ph * 3 + 0], __builtin_amdgcn_s_memtime() - tq0); atomicAdd(&db.qprof[ph * 3 + 1], 1ull); atomicAdd(&db.qprof[ph * 3 + 2], (unsigned long long)take); }
#endif
    }
}

// fills the queues of a run: every instance of every pool into PH_START
__global__ void k_sparse_sched_init(SpBatch db)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = db.nPools * (PH_NUM + 1) * QCTL;
    if (b < nq) {
        const int pool = b / ((PH_NUM + 1) * QCTL), k = (b / QCTL) % (PH_NUM + 1), f = b % QCTL;
        const int inPool = max(0, min(db.poolSize, db.B - pool * db.poolSize));
        int v = 0;
        if (k == PH_NUM) v = (f == 0) ? inPool : 0;                        // remaining
        else if (k == PH_START) v = (f == 0 || f == 2) ? inPool : 0;       // tail = count = instances of the pool, head = 0
        db.qctl[b] = v;
    }
    if (b < db.nPools * db.poolSize) {
        const int pool = b / db.poolSize, i = b % db.poolSize;
        for (int k = 0; k < PH_NUM; k++) {
            const bool filled = (k == PH_START && pool * db.poolSize + i < db.B);      // position i of the start queue holds instance i of the pool
            db.qring[((size_t)pool * PH_NUM + k) * db.poolSize + i] = filled ? q_slot(i + 1, pool * db.poolSize + i) : q_slot(i, -1);
        }
    }
}

template <int G>
static void sp_launch(const SpBatch& db, hipStream_t stream, hipEvent_t mid)
{
    const int ipw = 64 / G, grid = (db.B + ipw - 1) / ipw;
    // LDS: the working sets' bit sets of sp_ph_factor (G <= 16), the window of sp_factor_lds (per group G x G and 16 staged rows) otherwise
    const size_t ldsBytes = G <= 16 ? sizeof(unsigned) * (size_t)ipw * db.bitWords : sizeof(double) * (size_t)ipw * (G * G + 16 * G);
    hipLaunchKernelGGL(k_sparse_setup<G>, dim3(grid), dim3(WGS), ldsBytes, stream, db);
    (void)hipEventRecord(mid, stream);
    const int ninit = std::max(db.nPools * db.poolSize, db.nPools * (PH_NUM + 1) * QCTL);
    hipLaunchKernelGGL(k_sparse_sched_init, dim3((ninit + 255) / 256), dim3(256), 0, stream, db);
    // persistent wavefronts: as many as the batch has work for, at most what the device holds at once (they leave when their pool is done);
    // a multiple of the number of pools so that every pool is served
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // More wavefronts than one per 64 / G instances for batches that do not fill the machine (round 5, profiles/round5/sparse_waves_small_batches.log):
    // a streaming phase runs ONE instance on a wavefront's 64 lanes, so a wavefront that holds eight instances serialises their trial heads and
    // QP ends; with a wavefront per instance (up to 256) or per four instances B = 64 gains 52 %, 256: 44 %, 512: 39 %, 1024: 23 %, 2048: 12 %,
    // 8192: 4 % (4096: +-0); idle wavefronts back off exponentially, so the surplus costs nothing.
    int waves = std::max(grid, std::max(std::min(db.B, 256), db.B / 4));
    const int resident = cus * 4 * (G <= 8 ? SP_WAVES_PER_SIMD : 1);      // (k_sparse_sched's launch bounds)
    waves = std::min(waves, resident);
    if (const char* e = std::getenv("LCQP_SPARSE_WAVES")) { const int v = std::atoi(e); if (v >= 1) waves = std::min(v, resident); }      // experiment switch
    waves = ((waves + db.nPools - 1) / db.nPools) * db.nPools;
    SpBatch dbs = db;
    dbs.wideDiv = std::max(1, cus * 4 / std::max(1, db.nPools));
    if (const char* e = std::getenv("LCQP_SPARSE_WIDE_DIV")) { const int v = std::atoi(e); if (v >= 1) dbs.wideDiv = v; }      // experiment switch
    hipLaunchKernelGGL(k_sparse_sched<G>, dim3(waves), dim3(WGS), ldsBytes, stream, dbs);
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
    // two orderings of the band (lcqp_hip_sparse_create): [0] reverse Cuthill-McKee, [1] the same with every row behind its first variable
    struct Ord { std::vector<int> perm; int *iperm, *bandQ, *bandE, *bsrc, *bgate, *bdiag, *pnode, *Upos; bool rowsFollow; } ord[2];
    bool hasB;
    int useB;                      // ordering of the last sp_choose_ordering
    std::vector<int> qdiagHost;    // entry of Q_ii in the value array
    std::vector<double> diagRatio; // per instance: min_i Q_ii / max_i Q_ii of the loaded Hessian (1: not loaded yet)
    bool loaded, ran;
};

// Ordering [1] and the light regularisation of the polish are for batches whose Hessians are safely definite, judged by their diagonals
// (min Q_ii >= 1e-6 max Q_ii in every loaded instance); the pivot check of sp_polish covers what the diagonals do not show.
static void sp_choose_ordering(lcqp_hip_sparse* h)
{
    bool definite = h->loaded;      // nothing loaded yet: the plain ordering
    for (double r : h->diagRatio) definite = definite && (r >= 1e-6);
    const int k = (definite && h->hasB) ? 1 : 0;
    const lcqp_hip_sparse::Ord& o = h->ord[k];
    SpBatch& d = h->db;
    d.iperm = o.iperm; d.bandQ = o.bandQ; d.bandE = o.bandE; d.bsrc = o.bsrc; d.bgate = o.bgate; d.bdiag = o.bdiag; d.pnode = o.pnode; d.Upos = o.Upos;
    d.lightOK = (definite && o.rowsFollow) ? 1 : 0;
    h->useB = k;
}

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
    const int n = nV, m = nC + 2 * nComp, N = n + m;
    // the pattern is taken at its word below, so it is checked first: column pointers start at 0 and never decrease; row indices inside a
    // column strictly increase (sorted, no duplicates: a duplicate would share one band slot and lose a value); Q structurally symmetric
    // (both triangles given, as the reference hands Q to OSQP's P -- only entries with a mirror image reach the band)
    for (int pass = 0; pass < 2; pass++) {
        const int* P = pass ? Ap : Qp; const int* I = pass ? Ai : Qi; const int rows = pass ? m : n;
        if (P[0] != 0) { g_sp_err = "column pointers must start at 0"; return nullptr; }
        for (int c = 0; c < n; c++) {
            if (P[c + 1] < P[c]) { g_sp_err = "column pointers must not decrease"; return nullptr; }
            for (int k = P[c]; k < P[c + 1]; k++) {
                if (I[k] < 0 || I[k] >= rows) { g_sp_err = pass ? "row index out of bounds" : "Q index out of bounds"; return nullptr; }
                if (k > P[c] && I[k] <= I[k - 1]) { g_sp_err = "row indices of a column must be sorted and free of duplicates"; return nullptr; }
            }
        }
    }
    for (int c = 0; c < n; c++)
        for (int k = Qp[c]; k < Qp[c + 1]; k++) {
            const int r = Qi[k];
            if (r == c) continue;
            if (!std::binary_search(Qi + Qp[r], Qi + Qp[r + 1], c)) { g_sp_err = "Q must be structurally symmetric (both triangles given)"; return nullptr; }
        }
    const int nnzQ = Qp[n], nnzA = Ap[n];
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
    // Ordering: reverse Cuthill-McKee; while the half bandwidth exceeds what a lane group covers, the node of highest degree moves to the
    // border (at most SP_KBMAX nodes), the positions behind the band.  Arrow-shaped KKT matrices (a coupling row, a shared variable:
    // examples/OptimizeOnCircle.cpp:44) become a narrow band plus a few border nodes.
    std::vector<int> permA, border;
    bool general = false;
    if (const char* e = std::getenv("LCQP_SPARSE_GENERAL")) general = std::atoi(e) == 1;      // test hook: the general LDL' on a pattern the band engine would take
    lcqp_general::Symbolic sym;
    std::vector<char> isBorder(N, 0);
    std::vector<std::vector<int>> sub(N);
    auto bandwidth = [&](const std::vector<int>& pm) {
        std::vector<int> ip(N);
        for (int p = 0; p < N; p++) ip[pm[p]] = p;
        int wv = 0;
        for (int v = 0; v < N; v++) if (!isBorder[v]) for (int u : sub[v]) wv = std::max(wv, std::abs(ip[v] - ip[u]));
        return wv;
    };
    int wA = 0;
    for (; !general;) {
        for (int v = 0; v < N; v++) { sub[v].clear(); if (!isBorder[v]) for (int u : adj[v]) if (!isBorder[u]) sub[v].push_back(u); }
        std::vector<int> full;
        rcm_order(N, sub, full);
        permA.clear();
        for (int v : full) if (!isBorder[v]) permA.push_back(v);
        for (int v : border) permA.push_back(v);
        wA = bandwidth(permA);
        if (wA <= SP_WMAX) break;
        if ((int)border.size() >= SP_KBMAX) {
            // neither a banded nor a bordered problem: the general sparse LDL' (round 6; until then such a pattern was refused here and ran densified)
            general = true;
            break;
        }
        int best = -1; size_t deg = 0;
        for (int v = 0; v < N; v++) if (!isBorder[v] && sub[v].size() > deg) { deg = sub[v].size(); best = v; }
        if (best < 0) { g_sp_err = "ordering failed"; return nullptr; }
        isBorder[best] = 1; border.push_back(best);
    }
    if (general) {
        sym = lcqp_general::analyze(n, m, adj, Qp, Qi, Ep.data(), Ei.data(), 32);
        if (sym.maxFront > GEN_MAX_FRONT || sym.Lsize >= (1LL << 28) || sym.stackSize >= (1LL << 28)) {
            g_sp_err = "general sparse LDL' of this pattern: largest front " + std::to_string(sym.maxFront) + " (limit " + std::to_string(GEN_MAX_FRONT) + "), " +
                       std::to_string((long long)sym.Lsize) + " factor entries: too dense for the sparse engine (use the dense kernels)";
            return nullptr;
        }
        border.clear(); std::fill(isBorder.begin(), isBorder.end(), 0);
        permA = sym.perm; wA = 0;
        for (int v = 0; v < N; v++) sub[v] = adj[v];
    }
    const int kb = (int)border.size(), Nband = N - kb;
    // A second ordering of the same band nodes for batches whose Hessians are safely definite (chosen at run time, sp_choose_ordering): a
    // constraint row eliminated before every variable it touches gets the bare dual regularisation as its pivot (the LDL' is not pivoted),
    // which rules out the light regularisation of the polish (sp_polish).  Here such rows move to just behind their first variable, so that
    // every row pivot is -(delta2 + e D^-1 e').  With a Hessian that is nearly flat in that variable the same move is harmful (two active
    // rows that hinge on it cancel), hence the choice by the data.
    // Reverse Cuthill-McKee happens to put most multiplier nodes in front of their variables, and a band is as wide backwards: the second
    // ordering is the first one reversed, and what rows are still in front of all their variables move behind the first of them.
    std::vector<int> permB(permA);
    std::reverse(permB.begin(), permB.begin() + Nband);
    {
        std::vector<int> pos(N, -1), base(permB);
        for (int p = 0; p < Nband; p++) pos[base[p]] = p;
        std::vector<std::pair<double, int>> key(Nband);
        for (int p = 0; p < Nband; p++) {
            const int v = base[p];
            double k = p;
            if (v >= n) {
                int first = 1 << 30;
                for (int u : sub[v]) if (u < n && pos[u] >= 0) first = std::min(first, pos[u]);
                if (first != (1 << 30) && first > p) k = first + 0.5;
            }
            key[p] = {k, v};
        }
        std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
        for (int p = 0; p < Nband; p++) permB[p] = key[p].second;
    }
    const int wB = bandwidth(permB);
    auto lanes_for = [](int wv) { return wv < 8 ? 8 : (wv < 16 ? 16 : (wv < 32 ? 32 : 64)); };
    const bool hasB = !general && wB <= SP_WMAX && lanes_for(wB) == lanes_for(wA);      // not at the price of a wider lane group
    int w = hasB ? std::max(wA, wB) : wA;
    if (w < 1) w = 1;
    // lanes per instance: the smallest of 8, 16, 32, 64 above the half bandwidth (LCQP_SPARSE_LANES raises it: test hook); the general LDL' takes a wavefront
    int G = general ? 64 : lanes_for(w);
    if (const char* e = std::getenv("LCQP_SPARSE_LANES")) { const int v = std::atoi(e); if ((v == 16 || v == 32 || v == 64) && v > G) G = v; }
    const int ld = G, wS = G - 1;          // band rows are stored G wide: entry k of row i is K[i][i - (G-1) + k] (zero outside the true band)
    // what depends on the ordering: inverse permutation, band slot of every entry of Q and E (-1: not in the band -- an upper-triangle entry
    // of Q, or an entry of the border), where every off-diagonal band entry comes from (assembly inside the factorisation: -1 nothing,
    // k < nnzQ the entry k of Q, nnzQ + k the entry k of E in CSR order), the node behind a band position, the band positions of U
    struct OrdMaps { std::vector<int> perm, iperm, bandQ, bandE, bsrc, bgate, bdiag, Upos; };
    std::vector<int> qdiag(n, -1), Erow(nnzA);
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) if (Qi[k] == i) qdiag[i] = k;
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) Erow[k] = r;
    // the border: per border node its entries with band nodes (U) and with border nodes of lower index (C); the enumeration order does not
    // depend on the ordering of the band
    std::vector<int> Uptr(kb + 1, 0), Uother, Usrc, Ugate, Cptr(kb + 1, 0), Cb2, Csrc, Cgate, bidx(N, -1);
    for (int b = 0; b < kb; b++) bidx[border[b]] = b;
    for (int b = 0; b < kb; b++) {
        const int v = border[b];
        auto put = [&](int other, int src, int gate) {
            if (bidx[other] < 0) { Uother.push_back(other); Usrc.push_back(src); Ugate.push_back(gate); }
            else if (bidx[other] < b) { Cb2.push_back(bidx[other]); Csrc.push_back(src); Cgate.push_back(gate); }
        };
        if (v < n) {
            for (int k = Qp[v]; k < Qp[v + 1]; k++) if (Qi[k] != v) put(Qi[k], k, -1);                       // Q is symmetric: row v = column v
            for (int kc = ETp[v]; kc < ETp[v + 1]; kc++) put(n + ETi[kc], nnzQ + ETmap[kc], ETi[kc]);       // column v of E
        } else {
            const int r = v - n;
            for (int k = Ep[r]; k < Ep[r + 1]; k++) put(Ei[k], nnzQ + k, r);
        }
        Uptr[b + 1] = (int)Uother.size(); Cptr[b + 1] = (int)Cb2.size();
    }
    const int nU = (int)Uother.size(), nCb = (int)Cb2.size();
    auto build_maps = [&](const std::vector<int>& pm) {
        OrdMaps M;
        M.perm = pm; M.iperm.assign(N, 0);
        for (int p = 0; p < N; p++) M.iperm[pm[p]] = p;
        if (general) {      // no band: the maps of the band engines stay empty (sp_assemble / sp_factor_reg are never entered)
            M.bandQ.assign(nnzQ, -1); M.bandE.assign(nnzA, -1); M.bsrc.assign(1, -1); M.bgate.assign(1, -1); M.bdiag.assign(N, -1); M.Upos.assign(1, 0);
            return M;
        }
        M.bandQ.assign(nnzQ, -1); M.bandE.assign(nnzA, -1); M.bsrc.assign((size_t)N * ld, -1);
        for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) { const int pi = M.iperm[i], pj = M.iperm[Qi[k]]; if (pj <= pi && pi < Nband) M.bandQ[k] = pi * ld + wS - (pi - pj); }
        for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) { const int pr = M.iperm[n + r], pc = M.iperm[Ei[k]]; const int hi = std::max(pr, pc), lo = std::min(pr, pc); if (hi < Nband) M.bandE[k] = hi * ld + wS - (hi - lo); }
        for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) if (Qi[k] != i && M.bandQ[k] >= 0) M.bsrc[M.bandQ[k]] = k;
        for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) if (M.bandE[k] >= 0) M.bsrc[M.bandE[k]] = nnzQ + k;
        // the same information one level of indirection shorter (sp_factor_reg: load_row): the gating row of every band entry, the diagonal of every position
        M.bgate.assign((size_t)N * ld, -1);
        for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) if (M.bandE[k] >= 0) M.bgate[M.bandE[k]] = r;
        if (G <= 16) {
            // sp_factor_reg keeps a row in UPPER form relative to its diagonal: entry k of row r is K[r + k][r] = the lower-form entry
            // (r + k, ld - 1 - k); the diagonal slot (k = 0) is described by bdiag
            std::vector<int> su((size_t)N * ld, -1), gu((size_t)N * ld, -1);
            for (int r = 0; r < N; r++)
                for (int k = 1; k < ld && r + k < N; k++) { su[(size_t)r * ld + k] = M.bsrc[(size_t)(r + k) * ld + (ld - 1 - k)]; gu[(size_t)r * ld + k] = M.bgate[(size_t)(r + k) * ld + (ld - 1 - k)]; }
            M.bsrc.swap(su); M.bgate.swap(gu);
        }
        M.bdiag.assign(N, -1);
        for (int p_ = 0; p_ < N; p_++) {
            const int node = pm[p_];
            if (p_ >= Nband) M.bdiag[p_] = INT_MIN;
            else if (node < n) M.bdiag[p_] = qdiag[node];
            else M.bdiag[p_] = -2 - (node - n);
        }
        M.Upos.resize(nU);
        for (int e = 0; e < nU; e++) M.Upos[e] = M.iperm[Uother[e]];
        return M;
    };
    OrdMaps mapsA = build_maps(permA), mapsB = hasB ? build_maps(permB) : OrdMaps();
    if (hipSetDevice(device) != hipSuccess) { g_sp_err = "hipSetDevice failed"; return nullptr; }
    lcqp_hip_sparse* h = new (std::nothrow) lcqp_hip_sparse();
    if (!h) return nullptr;
    struct Guard { lcqp_hip_sparse* h; ~Guard() { if (h) lcqp_hip_sparse_destroy(h); } } guard{h};      // an exception below must not leak the handle
    h->device = device; h->nnzA = nnzA; h->loaded = false; h->ran = false; h->csr2csc = csr2csc; h->hasB = hasB; h->useB = 0; h->qdiagHost = qdiag; h->diagRatio.assign(batch, 1.0);
    h->stream = nullptr; h->ev0 = h->ev1 = h->ev2 = nullptr;
    SpBatch& d = h->db;
    memset(&d, 0, sizeof(d));
    d.B = batch; d.n = n; d.m = m; d.nC = nC; d.nComp = nComp; d.N = N; d.Np = ((N + 63) / 64) * 64; d.w = w; d.ld = ld; d.nnzQ = nnzQ; d.nnzE = nnzA; d.G = G; d.kb = kb; d.nU = nU; d.nCb = nCb;
    d.general = general ? 1 : 0;
    d.kfStride = general ? (size_t)sym.Lsize : (size_t)d.Np * G;
    if (general) { d.gnF = sym.nF; d.gMaxFront = sym.maxFront; d.gLsize = (unsigned)sym.Lsize; d.gStackSize = (unsigned)std::max<long long>(sym.stackSize, 1); d.w = 0; }
    d.bitWords = (G <= 16 && (size_t)(64 / G) * ((m + 31) / 32) * sizeof(unsigned) <= 16384) ? (m + 31) / 32 : 0;      // at most 16 KB of LDS per wavefront
    if (const char* e = std::getenv("LCQP_SPARSE_NOBITS")) { if (std::atoi(e) == 1) d.bitWords = 0; }                    // test hook: the path of problems with more rows than that
    {   // algorithmic bytes per event (what each event has to read and write once: 8-byte values, 4-byte indices)
        const double dN = N, dq = nnzQ, de = nnzA, Nb = N - kb;
        d.by[BY_ASSEMBLE] = 8.0 * (dN * ld + dq + de) + 4.0 * (dq + de);
        d.by[BY_FACTOR_LDS] = 8.0 * (3.0 * dN * (w + 1));
        d.by[BY_FACTOR] = 12.0 * (dq + de) + 8.0 * dN * (w + 2);
        d.by[BY_SOLVE] = 8.0 * (2.0 * dN * w + 4.0 * dN);
        d.by[BY_BORDER_PREPARE] = 8.0 * (2.0 * (double)nU + (double)kb * d.Np);
        d.by[BY_BORDER_SOLVE] = 8.0 * ((double)kb * Nb + 2.0 * Nb + nU);
        d.by[BY_EX] = 12.0 * de + 8.0 * (n + m);
        d.by[BY_SWEEP] = 12.0 * (dq + de) + 8.0 * (3.0 * n + m);
        d.by[BY_START] = 12.0 * dq + 2.0 * 12.0 * de;
        d.by[BY_E] = 12.0 * de;
        if (general) {      // a factorisation reads every value of Q and E once and writes the panels and 1 / D; a solve reads the panels twice
            d.by[BY_FACTOR] = 12.0 * (dq + de) + 8.0 * ((double)sym.Lsize + dN);
            d.by[BY_SOLVE] = 8.0 * (2.0 * (double)sym.Lsize + 4.0 * dN);
        }
    }
    const size_t Np = d.Np;
    lcqp_hip_options_default(&d.opt);
    bool ok = hipStreamCreate(&h->stream) == hipSuccess && hipEventCreate(&h->ev0) == hipSuccess && hipEventCreate(&h->ev1) == hipSuccess &&
              hipEventCreate(&h->ev2) == hipSuccess;
    const size_t B = batch;
    ok = ok && (d.Qp = sp_alloc<int>(h, n + 1, Qp)) && (d.Qi = sp_alloc<int>(h, nnzQ, Qi)) && (d.Ep = sp_alloc<int>(h, m + 1, Ep.data())) &&
         (d.Ei = sp_alloc<int>(h, nnzA, Ei.data())) && (d.ETp = sp_alloc<int>(h, n + 1, ETp.data())) && (d.ETi = sp_alloc<int>(h, nnzA, ETi.data())) &&
         (d.ETmap = sp_alloc<int>(h, nnzA, ETmap.data())) &&
         (d.qdiag = sp_alloc<int>(h, n, qdiag.data())) && (d.Erow = sp_alloc<int>(h, nnzA, Erow.data()));
    for (int k = 0; k < (hasB ? 2 : 1); k++) {
        OrdMaps& M = k ? mapsB : mapsA;
        lcqp_hip_sparse::Ord& o = h->ord[k];
        // the light regularisation needs every row of the band behind one of its variables (sp_polish)
        o.rowsFollow = true;
        for (int r = 0; r < m; r++) {
            if (M.iperm[n + r] >= Nband) continue;
            bool follows = false;
            for (int e = Ep[r]; e < Ep[r + 1]; e++) follows = follows || M.iperm[Ei[e]] < M.iperm[n + r];
            o.rowsFollow = o.rowsFollow && follows;
        }
        ok = ok && (o.iperm = sp_alloc<int>(h, N, M.iperm.data())) && (o.bandQ = sp_alloc<int>(h, nnzQ, M.bandQ.data())) &&
             (o.bandE = sp_alloc<int>(h, nnzA, M.bandE.data())) && (o.bsrc = sp_alloc<int>(h, M.bsrc.size(), M.bsrc.data())) &&
             (o.bgate = sp_alloc<int>(h, M.bgate.size(), M.bgate.data())) && (o.bdiag = sp_alloc<int>(h, M.bdiag.size(), M.bdiag.data())) &&
             (o.pnode = sp_alloc<int>(h, N, M.perm.data())) && (o.Upos = sp_alloc<int>(h, nU, M.Upos.data()));
        o.perm = std::move(M.perm);
    }
    if (ok) sp_choose_ordering(h);
    if (kb > 0)
        ok = ok && (d.bnode = sp_alloc<int>(h, kb, border.data())) && (d.Uptr = sp_alloc<int>(h, kb + 1, Uptr.data())) &&
             (d.Usrc = sp_alloc<int>(h, nU, Usrc.data())) && (d.Ugate = sp_alloc<int>(h, nU, Ugate.data())) && (d.Cptr = sp_alloc<int>(h, kb + 1, Cptr.data())) &&
             (d.Cb2 = sp_alloc<int>(h, nCb, Cb2.data())) && (d.Csrc = sp_alloc<int>(h, nCb, Csrc.data())) && (d.Cgate = sp_alloc<int>(h, nCb, Cgate.data())) &&
             (d.bW = sp_alloc<double>(h, (size_t)batch * 2 * kb * d.Np)) && (d.bUv = sp_alloc<double>(h, (size_t)batch * 2 * nU)) &&
             (d.bS = sp_alloc<double>(h, (size_t)batch * 2 * kb * kb));
    // ELL slabs of the three gathers (g_ell): rows of Q, rows of E, columns of E
    auto make_ell = [&](EllMat& e, int rows, const std::vector<int>& ptr, const std::vector<int>& idx, const int* map, const int* dptr, const int* didx, const int* dmap) {
        int mx = 0;
        for (int i = 0; i < rows; i++) mx = std::max(mx, ptr[i + 1] - ptr[i]);
        const int W = mx <= 4 ? 4 : 8;
        std::vector<int> ei((size_t)W * rows, 0), ep((size_t)W * rows, -1);
        for (int i = 0; i < rows; i++)
            for (int q = 0; q < W && ptr[i] + q < ptr[i + 1]; q++) { const int k = ptr[i] + q; ei[(size_t)q * rows + i] = idx[k]; ep[(size_t)q * rows + i] = map ? map[k] : k; }
        e.rows = rows; e.W = W; e.tails = mx > W ? 1 : 0; e.ptr = dptr; e.cidx = didx; e.cmap = dmap;
        e.epos = nullptr;      // without a map the position of entry q of row i is ptr[i] + q (g_ell): no position slab
        return (e.eidx = sp_alloc<int>(h, ei.size(), ei.data())) && (!map || (e.epos = sp_alloc<int>(h, ep.size(), ep.data())));
    };
    ok = ok && make_ell(d.ellQ, n, std::vector<int>(Qp, Qp + n + 1), std::vector<int>(Qi, Qi + nnzQ), nullptr, d.Qp, d.Qi, nullptr) &&
         make_ell(d.ellE, m, Ep, Ei, nullptr, d.Ep, d.Ei, nullptr) && make_ell(d.ellT, n, ETp, ETi, ETmap.data(), d.ETp, d.ETi, d.ETmap);
    ok = ok && (d.Qx = sp_alloc<double>(h, B * nnzQ)) && (d.Ex = sp_alloc<double>(h, B * nnzA)) &&
         (d.Kb = sp_alloc<double>(h, (G > 16 && !general) ? B * N * ld : 0)) &&      // the band array is only written by the LDS-window factorisation
         (d.KaF = sp_alloc<double>(h, B * d.kfStride)) && (d.KaD = sp_alloc<double>(h, B * Np)) &&
         (d.KpF = sp_alloc<double>(h, B * d.kfStride)) && (d.KpD = sp_alloc<double>(h, B * Np)) &&
         (d.K0 = sp_alloc<double>(h, G <= 16 ? B * Np * G : 0)) &&
         (d.nv = sp_alloc<double>(h, B * NV_NUM * n)) && (d.mv = sp_alloc<double>(h, B * MV_NUM * m)) && (d.Nv = sp_alloc<double>(h, B * 2 * Np)) &&
         (d.lbL = sp_alloc<double>(h, B * nComp)) && (d.lbR = sp_alloc<double>(h, B * nComp)) && (d.mi = sp_alloc<int>(h, B * MI_NUM * m)) &&
         (d.info = sp_alloc<SpInfo>(h, B)) && (d.stats = sp_alloc<lcqp_stats_t>(h, B)) && (d.xout = sp_alloc<double>(h, B * n)) &&
         (d.yout = sp_alloc<double>(h, B * m));
    if (general) {
        std::vector<unsigned> lo(sym.Loff.begin(), sym.Loff.end()), co(sym.CBoff.begin(), sym.CBoff.end());
        std::vector<int> meta((size_t)sym.nF * GEN_META, 0), cinfo(std::max<size_t>(sym.child.size(), 1) * 4, 0);
        for (int f = 0; f < sym.nF; f++) {
            int* mt = meta.data() + (size_t)f * GEN_META;
            mt[0] = sym.np[f]; mt[1] = sym.nb[f]; mt[2] = sym.piv0[f]; mt[3] = sym.rowPtr[f]; mt[4] = sym.asmPtr[f]; mt[5] = sym.asmPtr[f + 1];
            mt[6] = sym.childPtr[f]; mt[7] = sym.childPtr[f + 1]; mt[8] = (int)sym.Loff[f]; mt[9] = (int)sym.CBoff[f];
        }
        for (size_t ci = 0; ci < sym.child.size(); ci++) { const int ch = sym.child[ci]; cinfo[4 * ci] = sym.nb[ch]; cinfo[4 * ci + 1] = (int)sym.CBoff[ch]; cinfo[4 * ci + 2] = sym.rowPtr[ch]; }
        ok = ok && (d.gPiv0 = sp_alloc<int>(h, sym.piv0.size(), sym.piv0.data())) && (d.gNp = sp_alloc<int>(h, sym.np.size(), sym.np.data())) &&
             (d.gNb = sp_alloc<int>(h, sym.nb.size(), sym.nb.data())) && (d.gRowPtr = sp_alloc<int>(h, sym.rowPtr.size(), sym.rowPtr.data())) &&
             (d.gRows = sp_alloc<int>(h, std::max<size_t>(sym.rows.size(), 1), sym.rows.empty() ? nullptr : sym.rows.data())) &&
             (d.gChildPtr = sp_alloc<int>(h, sym.childPtr.size(), sym.childPtr.data())) &&
             (d.gChild = sp_alloc<int>(h, std::max<size_t>(sym.child.size(), 1), sym.child.empty() ? nullptr : sym.child.data())) &&
             (d.gRel = sp_alloc<int>(h, std::max<size_t>(sym.rel.size(), 1), sym.rel.empty() ? nullptr : sym.rel.data())) &&
             (d.gAsmPtr = sp_alloc<int>(h, sym.asmPtr.size(), sym.asmPtr.data())) && (d.gAsmSrc = sp_alloc<int>(h, sym.asmSrc.size(), sym.asmSrc.data())) &&
             (d.gAsmGate = sp_alloc<int>(h, sym.asmGate.size(), sym.asmGate.data())) && (d.gAsmPos = sp_alloc<int>(h, sym.asmPos.size(), sym.asmPos.data())) &&
             (d.gLoff = sp_alloc<unsigned>(h, lo.size(), lo.data())) && (d.gCBoff = sp_alloc<unsigned>(h, co.size(), co.data())) &&
             (d.gMeta = sp_alloc<int>(h, meta.size(), meta.data())) && (d.gChildInfo = sp_alloc<int>(h, cinfo.size(), cinfo.data())) &&
             (d.gStack = sp_alloc<double>(h, B * d.gStackSize)) && (d.gFront = sp_alloc<double>(h, B * (size_t)d.gMaxFront * d.gMaxFront));
    }
    {
        // pools of the phase machine (k_sparse_sched): the largest power of two of instances whose per-instance arrays all stay below 4 GiB
        // (the 32-bit lane offsets of SpCtx::arr), at most the batch rounded up to a power of two
        size_t perInst = sizeof(double) * std::max<size_t>({(size_t)nnzQ, (size_t)nnzA, 2 * Np, (size_t)((G > 16 && !general) ? (size_t)N * ld : 0), d.kfStride,
                                                            general ? (size_t)d.gStackSize : 0, general ? (size_t)d.gMaxFront * d.gMaxFront : 0,
                                                            2 * (size_t)kb * Np, 2 * (size_t)nU, (size_t)NV_NUM * n, (size_t)MV_NUM * m, (size_t)nComp});
        perInst = std::max(perInst, sizeof(int) * (size_t)MI_NUM * m);
        int pool = 1;
        while ((size_t)(2 * pool) * perInst < ((size_t)1 << 32) && pool < batch) pool *= 2;
        if (const char* e = std::getenv("LCQP_SPARSE_POOL")) { const int v = std::atoi(e); if (v >= 1 && v < pool && (v & (v - 1)) == 0) pool = v; }      // test hook: several small pools
        d.poolSize = pool; d.nPools = (batch + pool - 1) / pool;
        ok = ok && (d.state = sp_alloc<SpState>(h, B)) && (d.qring = sp_alloc<unsigned long long>(h, (size_t)d.nPools * PH_NUM * pool)) &&
             (d.qctl = sp_alloc<int>(h, (size_t)d.nPools * (PH_NUM + 1) * QCTL)) && (d.qprof = sp_alloc<unsigned long long>(h, (PH_NUM + 1) * 3));
    }
    if (!ok) { g_sp_err = "device allocation failed"; return nullptr; }      // (the guard destroys the handle)
    guard.h = nullptr;
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
extern "C" int lcqp_hip_sparse_lanes(const lcqp_hip_sparse_t* h) { return h ? h->db.G : -1; }
extern "C" int lcqp_hip_sparse_border(const lcqp_hip_sparse_t* h) { return h ? h->db.kb : -1; }
extern "C" int lcqp_hip_sparse_fronts(const lcqp_hip_sparse_t* h) { return h ? (h->db.general ? h->db.gnF : 0) : -1; }
extern "C" int lcqp_hip_sparse_get_ordering(const lcqp_hip_sparse_t* h, int* perm)
{
    if (!h || !perm) return LCQP_INVALID_ARGUMENT;
    const std::vector<int>& pm = h->ord[h->useB].perm;      // the ordering the loaded Hessians select (sp_choose_ordering)
    memcpy(perm, pm.data(), sizeof(int) * pm.size());
    return 0;
}

extern "C" int lcqp_hip_sparse_set_options(lcqp_hip_sparse_t* h, const lcqp_options_t* opt)
{
    if (!h || !opt) return LCQP_INVALID_ARGUMENT;
    if (opt->nDynamicPenalty > 64) { g_sp_err = "nDynamicPenalty > 64 unsupported"; return LCQP_HIP_UNSUPPORTED; }
    h->db.opt = *opt;
    // per-iterate tracking buffers: kept at the largest trace this handle was asked for (a regrow frees the smaller ones); the kernel records
    // only while storeSteps is on -- traceCap < 0 keeps the buffers of a handle whose tracking has been switched off again
    SpBatch& d = h->db;
    const int want = opt->storeSteps ? std::min(std::max(opt->maxIterations + 1, 1), 4096) : 0;
    const int have = d.traceCap < 0 ? -d.traceCap : d.traceCap;
    if (want > have) {
        if (hipSetDevice(h->device) != hipSuccess) { g_sp_err = "hipSetDevice failed"; return LCQP_HIP_ERROR; }
        (void)hipStreamSynchronize(h->stream);
        for (void* old : {(void*)d.traceS, (void*)d.traceX, (void*)d.traceLen})
            if (old) { (void)hipFree(old); h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), old), h->allocs.end()); }
        d.traceS = d.traceX = nullptr; d.traceLen = nullptr; d.traceCap = 0;
        double *ts = sp_alloc<double>(h, (size_t)d.B * want * 8), *tx = sp_alloc<double>(h, (size_t)d.B * want * d.n);
        int* tl = sp_alloc<int>(h, (size_t)d.B);
        if (!ts || !tx || !tl) { g_sp_err = "out of device memory for the iterate trace"; return LCQP_HIP_ERROR; }
        d.traceS = ts; d.traceX = tx; d.traceLen = tl; d.traceCap = want;
    } else d.traceCap = opt->storeSteps ? have : -have;
    return 0;
}

/* per-iterate trace of one instance of the last run (needs options.storeSteps), as lcqp_hip_batch_get_trace */
extern "C" int lcqp_hip_sparse_get_trace(lcqp_hip_sparse_t* h, int instance, int cap, double* scalars, double* x, int* len)
{
    if (!h || !len) return LCQP_INVALID_ARGUMENT;
    SpBatch& d = h->db;
    *len = 0;
    if (instance < 0 || instance >= d.B) return LCQP_INVALID_ARGUMENT;
    if (d.traceCap <= 0) return 0;      // no buffers, or tracking switched off: an empty trace
    SPCHK(hipSetDevice(h->device));
    SPCHK(hipStreamSynchronize(h->stream));
    int n = 0;
    SPCHK(hipMemcpy(&n, d.traceLen + instance, sizeof(int), hipMemcpyDeviceToHost));
    n = std::min(n, std::min(cap, d.traceCap));
    if (n > 0 && scalars) SPCHK(hipMemcpy(scalars, d.traceS + (size_t)instance * d.traceCap * 8, sizeof(double) * 8 * n, hipMemcpyDeviceToHost));
    if (n > 0 && x) SPCHK(hipMemcpy(x, d.traceX + (size_t)instance * d.traceCap * d.n, sizeof(double) * (size_t)d.n * n, hipMemcpyDeviceToHost));
    *len = n;
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
    // instances with and without lbL / lbR may share a batch: an absent vector is the zero vector, the same arithmetic (lcqp_hip_batch_load)
    if (!h->loaded || first == 0) { d.hasLbL = hasL; d.hasLbR = hasR; }
    else { d.hasLbL |= hasL; d.hasLbR |= hasR; }
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
        double dmin = INFINITY, dmax = 0.0;
        for (int i = 0; i < n; i++) { const double q = h->qdiagHost[i] >= 0 ? Qx[(size_t)k * d.nnzQ + h->qdiagHost[i]] : 0.0; dmin = std::min(dmin, q); dmax = std::max(dmax, std::fabs(q)); }
        h->diagRatio[b] = (dmax > 0.0 && dmin > 0.0) ? dmin / dmax : 0.0;
        SPCHK(hipMemcpy(d.Qx + b * d.nnzQ, Qx + (size_t)k * d.nnzQ, sizeof(double) * d.nnzQ, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.Ex + b * d.nnzE, ex.data(), sizeof(double) * d.nnzE, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.nv + b * NV_NUM * n, nvb.data(), sizeof(double) * nvb.size(), hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.mv + b * MV_NUM * m, mvb.data(), sizeof(double) * mvb.size(), hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.lbL + b * nK, lb.data(), sizeof(double) * nK, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.lbR + b * nK, rb.data(), sizeof(double) * nK, hipMemcpyHostToDevice));
        SPCHK(hipMemcpy(d.info + b, &info, sizeof(info), hipMemcpyHostToDevice));
    }
    h->loaded = true;
    sp_choose_ordering(h);
    return 0;
}
catch (...) { g_sp_err = "out of host memory"; return LCQP_HIP_ERROR; }

/* -DLCQP_SCHED_PROFILE builds: per phase (rows 0 .. PH_NUM-1: start, round, trial, factor, correct, qp end; row PH_NUM: polls without work) the clock
 * ticks (100 MHz), wavefront steps and instances served, summed over the wavefronts of all runs since the handle was created: 3 (PH_NUM + 1) values */
extern "C" int lcqp_hip_sparse_sched_profile(lcqp_hip_sparse_t* h, unsigned long long* out)
{
    if (!h || !out) return LCQP_INVALID_ARGUMENT;
    SPCHK(hipSetDevice(h->device));
    SPCHK(hipStreamSynchronize(h->stream));
    SPCHK(hipMemcpy(out, h->db.qprof, sizeof(unsigned long long) * 3 * (PH_NUM + 1), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lcqp_hip_sparse_run(lcqp_hip_sparse_t* h)
try {
    if (!h || !h->loaded) return LCQP_LCQPOBJECT_NOT_SETUP;
    SPCHK(hipSetDevice(h->device));
    sp_choose_ordering(h);
    SPCHK(hipEventRecord(h->ev0, h->stream));
    switch (h->db.G) {
        case 8: sp_launch<8>(h->db, h->stream, h->ev1); break;
        case 16: sp_launch<16>(h->db, h->stream, h->ev1); break;
        case 32: sp_launch<32>(h->db, h->stream, h->ev1); break;
        default: sp_launch<64>(h->db, h->stream, h->ev1); break;
    }
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

// -DLCQP_PROFILE builds (tools/gpu.py sparse_profile): mean clock ticks per instance and phase of the last run
// (products, assembly, factorisation, forward sweeps, backward sweeps, vector operations, LCQP level, -)
extern "C" int lcqp_hip_sparse_read_profile(lcqp_hip_sparse_t* h, double* out)
try {
#if defined(LCQP_PROFILE) || defined(SP_DEBUG)
    if (!h || !out) return LCQP_INVALID_ARGUMENT;
    SpBatch& d = h->db;
    SPCHK(hipSetDevice(h->device));
    SPCHK(hipStreamSynchronize(h->stream));
    std::vector<SpInfo> info(d.B);
    SPCHK(hipMemcpy(info.data(), d.info, sizeof(SpInfo) * (size_t)d.B, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; k++) { out[k] = 0.0; for (auto& i : info) out[k] += i.prof[k] / d.B; }
    return 0;
#else
    (void)h; (void)out;
    return LCQP_HIP_UNSUPPORTED;
#endif
}
catch (...) { return LCQP_HIP_ERROR; }

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
