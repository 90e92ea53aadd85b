// lcqp_wg.hpp -- workgroup-cooperative fp64 building blocks for gfx950 (CDNA4).
//
// Execution model: ONE 256-thread workgroup (4 wave64) owns ONE LCQP instance for the whole solve
// (DESIGN.md §Mapping).  Every routine here is called by all 256 threads with workgroup-uniform
// arguments, reads its operands from global memory (HBM/L2), uses LDS only as routine-private
// scratch, writes its results back to global memory and ends with a workgroup barrier.
//
// Streaming convention for the HBM-bound passes: a matrix row (np = 128*NCH doubles, row-major, as
// the reference stores it -- src/Utilities.cpp:43) is read by one wave with 16-byte loads,
// lane l taking columns 128k+2l, 128k+2l+1 (1 KiB per wave-instruction, fully coalesced).  Products
// with the transposed / symmetric matrix are lane-local AXPY accumulations (no cross-lane traffic);
// products with the matrix itself use one wavefront shuffle reduction per row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcqp {

constexpr int WG = 256;          // threads per workgroup
constexpr int NWAVE = 4;         // waves per workgroup
// doubles of routine-private LDS: 35.25 KiB (4 workgroups per CU) up to np = 512; the np = 1024 instantiation (NCH = 8) needs room for
// six vectors of length np (wg_symv) and takes 48 KiB (3 workgroups per CU), np = 2048 96 KiB.  np = 4096 (NCH = 32) would need 192 KiB that
// way: its four waves combine their partial sums in TWO copies instead of four (wg_ncopy), four vectors of length np = 128 KiB.
constexpr int wg_ncopy(int nch) { return nch > 16 ? 2 : 4; }
constexpr int arena_doubles(int nch) { return (wg_ncopy(nch) + 2) * 128 * nch > 4512 ? (wg_ncopy(nch) + 2) * 128 * nch : 4512; }
constexpr int ARENA = arena_doubles(4);
constexpr int TILE_LD = 65;      // padded leading dimension of the 64x64 LDS tile
// active rows the subsolver has room for: wg_trsv keeps the vector and four partial copies of it in the arena (896; 1216 at NCH = 8)
constexpr int max_active(int nch) { return 64 * (arena_doubles(nch) / (5 * 64)); }
constexpr int LCQP_MAX_ACTIVE = max_active(4);

enum { ST_INACT = 0, ST_LOWER = 1, ST_UPPER = 2, ST_EQ = 3 };

// Row state of the subsolver kept in LDS by the homotopy kernel (round 3; np <= 256 and at most LDS_ROWS_MAX rows of E): multipliers,
// E x, screening margins and status of every row live in arena[LDS_ROWS_OFF ..) for the whole launch -- the passes over them between
// the sweeps are LDS passes instead of round trips through L2 (with 1024 instances per GPU the 170 KB of vectors per instance do not
// stay in the 4 MiB L2 of an XCD).  Below LDS_ROWS_OFF the arena stays routine-private scratch (6 np doubles for the sweeps, 3 capS for
// the rotations, 2048 for the widest pass over the inverse factor); the one user of the whole arena, the 64 x 64 tile of wg_chol, runs
// between a save and a restore of the row state (qp_polish).
constexpr int LDS_ROWS_MAX = 640;
constexpr int LDS_ROWS_OFF = 2048;
static_assert(LDS_ROWS_OFF + 3 * LDS_ROWS_MAX + LDS_ROWS_MAX / 2 <= 4512, "row state must fit the arena");

struct Lds {
    double* arena;  // arena_doubles(NCH) doubles
    double* red;    // 16 doubles
    int* ired;      // 16 ints
};

// Matrix streams (rows of Q, E, Et, the factor L1): every byte is used once per pass and the matrices of a batch (12 GB) never fit a
// cache, so these loads are non-temporal and leave L2 / the Infinity Cache to what IS re-read: the vectors and the inverse factor T
// (139 KB per instance).  Same-box A/B 40.1 -> 38.3 ms (profiles/round3); -DLCQP_NO_NT_STREAMS restores plain loads.
#ifndef LCQP_NO_NT_STREAMS
typedef double d2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const double2* p) { const d2v_t v = __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(p)); return double2{v.x, v.y}; }
__device__ __forceinline__ double ld_stream(const double* p) { return __builtin_nontemporal_load(p); }
#else
__device__ __forceinline__ double2 ld_stream(const double2* p) { return *p; }
__device__ __forceinline__ double ld_stream(const double* p) { return *p; }
#endif

// Everything a thread computes from its thread number and uniform values is invariant in every loop of a kernel, and the compiler
// hoists it all to the top of the kernel (hundreds of addresses and masks, spilled to scratch at once).  An empty volatile asm cannot
// be hoisted: what is derived from the laundered thread number stays inside the routine that uses it.
__device__ __forceinline__ int tid_here() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ int lane_id() { return tid_here() & 63; }
__device__ __forceinline__ int wave_id() { return tid_here() >> 6; }
#ifndef LCQP_SEQ_WAVE
#define LCQP_SEQ_WAVE 1
#endif

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// broadcast lane `src` (must be wave-uniform) through the scalar unit: 2 v_readlane_b32, no LDS traffic
__device__ __forceinline__ double wave_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// make a value the compiler cannot prove wave-uniform live in scalar registers
__device__ __forceinline__ double uniform_d(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a 64-bit constant the compiler materialises where it is used (scalar registers) instead of hoisting it into a vector register pair
__device__ __forceinline__ uint64_t opaque_u64(uint64_t v) { asm volatile("" : "+s"(v)); return v; }

// Element-wise loop over n entries by the workgroup: all the loads of a tile of U*WG entries (load(i) returns them by value) are
// issued before the first store of the tile (store(i, v)).  Written as load - store - load ... the compiler has to keep the order
// (the arrays may alias) and a loop over 640 rows pays three memory round trips instead of one.
template <int U, class L, class S>
__device__ __forceinline__ void wg_map(int n, L load, S store)
{
    const int t = tid_here();
    for (int i0 = t; i0 < n; i0 += U * WG) {
        decltype(load(0)) v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * WG; v[u] = load(i < n ? i : 0); }
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * WG; if (i < n) store(i, v[u]); }
    }
}
struct MapID { int s; double a; };
struct MapID3 { int s; double a, b, c; };
struct MapD4 { double a, b, c, d; };
struct MapID4 { int s; double a, b, c, d; };

// workgroup reductions; result is uniform across the workgroup (and held in SGPRs). Ends with a barrier.
__device__ __forceinline__ double block_sum(double v, Lds lds)
{
    v = wave_sum(v);
    if (lane_id() == 0) lds.red[wave_id()] = v;
    __syncthreads();
    double r = lds.red[0] + lds.red[1] + lds.red[2] + lds.red[3];
    __syncthreads();
    return uniform_d(r);
}
__device__ __forceinline__ double block_max(double v, Lds lds)
{
    v = wave_max(v);
    if (lane_id() == 0) lds.red[wave_id()] = v;
    __syncthreads();
    double r = fmax(fmax(lds.red[0], lds.red[1]), fmax(lds.red[2], lds.red[3]));
    __syncthreads();
    return uniform_d(r);
}
// two reductions for the price of one barrier pair: (sum, sum) and (max, sum)
__device__ __forceinline__ void block_sum2(double a, double b, double& ra, double& rb, Lds lds)
{
    a = wave_sum(a); b = wave_sum(b);
    if (lane_id() == 0) { lds.red[wave_id()] = a; lds.red[4 + wave_id()] = b; }
    __syncthreads();
    const double sa = lds.red[0] + lds.red[1] + lds.red[2] + lds.red[3];
    const double sb = lds.red[4] + lds.red[5] + lds.red[6] + lds.red[7];
    __syncthreads();
    ra = uniform_d(sa); rb = uniform_d(sb);
}
__device__ __forceinline__ void block_max_sum(double a, double b, double& ra, double& rb, Lds lds)
{
    a = wave_max(a); b = wave_sum(b);
    if (lane_id() == 0) { lds.red[wave_id()] = a; lds.red[4 + wave_id()] = b; }
    __syncthreads();
    const double sa = fmax(fmax(lds.red[0], lds.red[1]), fmax(lds.red[2], lds.red[3]));
    const double sb = lds.red[4] + lds.red[5] + lds.red[6] + lds.red[7];
    __syncthreads();
    ra = uniform_d(sa); rb = uniform_d(sb);
}
__device__ __forceinline__ int block_or(int v, Lds lds)
{
    int any = __any(v) ? 1 : 0;
    if (lane_id() == 0) lds.ired[wave_id()] = any;
    __syncthreads();
    int r = lds.ired[0] | lds.ired[1] | lds.ired[2] | lds.ired[3];
    __syncthreads();
    return uniform_i(r);
}
// bitwise OR over the block of a value in {0,1,2,3} (uniform result)
__device__ __forceinline__ int block_or_bits(int v, Lds lds)
{
    const int any = (__any(v & 1) ? 1 : 0) | (__any(v & 2) ? 2 : 0);
    if (lane_id() == 0) lds.ired[wave_id()] = any;
    __syncthreads();
    int r = lds.ired[0] | lds.ired[1] | lds.ired[2] | lds.ired[3];
    __syncthreads();
    return uniform_i(r);
}
__device__ __forceinline__ int block_sum_i(int v, Lds lds)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane_id() == 0) lds.ired[wave_id()] = v;
    __syncthreads();
    int r = lds.ired[0] + lds.ired[1] + lds.ired[2] + lds.ired[3];
    __syncthreads();
    return uniform_i(r);
}

// Ordered compaction: out[0..min(count, cap)) = the indices i in [0, N) with pred(i), ascending; returns count (uniform).
// Each thread scans a contiguous chunk; exclusive scan over the 256 per-thread counts.  out: global memory.  Ends with a barrier.
template <class Pred>
__device__ __forceinline__ int wg_compact(int N, Pred pred, int* out, Lds lds, int cap = 1 << 30)
{
    const int per = (N + WG - 1) / WG;
    const int i0 = tid_here() * per, i1 = min(N, i0 + per);
    int cnt = 0;
    for (int i = i0; i < i1; i++) cnt += pred(i) ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) { const int v = __shfl_up(incl, ofs, 64); if (lane_id() >= ofs) incl += v; }
    if (lane_id() == 63) lds.ired[8 + wave_id()] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave_id(); w++) base += lds.ired[8 + w];
    const int total = uniform_i(lds.ired[8] + lds.ired[9] + lds.ired[10] + lds.ired[11]);
    int pos = base + incl - cnt;
    for (int i = i0; i < i1; i++) if (pred(i)) { if (pos < cap) out[pos] = i; pos++; }     // entries beyond cap are counted, not stored
    __syncthreads();
    return total;
}

// ---------------------------------------------------------------------------------------------
// small vector helpers (global -> global, length a multiple of nothing in particular)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wg_copy(double* dst, const double* src, int n)
{
    for (int i = tid_here(); i < n; i += WG) dst[i] = src[i];
    __syncthreads();
}
__device__ __forceinline__ void wg_fill(double* dst, double v, int n)
{
    for (int i = tid_here(); i < n; i += WG) dst[i] = v;
    __syncthreads();
}
__device__ __forceinline__ double wg_dot(const double* a, const double* b, int n, Lds lds)
{
    double s = 0;
    for (int i = tid_here(); i < n; i += WG) s += a[i] * b[i];
    return block_sum(s, lds);
}
__device__ __forceinline__ double wg_maxabs(const double* a, int n, Lds lds)
{
    double s = 0;
    for (int i = tid_here(); i < n; i += WG) s = fmax(s, fabs(a[i]));
    return block_max(s, lds);
}

// max|a[0..na)| and max b[0..nb) (b >= 0) with one barrier pair
__device__ __forceinline__ void wg_maxabs2(const double* a, const double* b, int na, int nb, double& ma, double& mb, Lds lds)
{
    double s = 0, u = 0;
    for (int i = tid_here(); i < na; i += WG) { s = fmax(s, fabs(a[i])); if (i < nb) u = fmax(u, b[i]); }
    s = wave_max(s); u = wave_max(u);
    if (lane_id() == 0) { lds.red[wave_id()] = s; lds.red[4 + wave_id()] = u; }
    __syncthreads();
    const double ra = fmax(fmax(lds.red[0], lds.red[1]), fmax(lds.red[2], lds.red[3]));
    const double rb = fmax(fmax(lds.red[4], lds.red[5]), fmax(lds.red[6], lds.red[7]));
    __syncthreads();
    ma = uniform_d(ra); mb = uniform_d(rb);
}

// cross-wave combine of per-lane accumulators acc[2*NCH] (lane l holds columns 128k+2l, +1):
// out[c] = post(c, sum over waves).  Uses arena[0 .. 4*np).
template <int NCH, class Post>
__device__ __forceinline__ void wg_combine(const double (&acc)[2 * NCH], Lds lds, Post post)
{
    constexpr int np = 128 * NCH;
    double* red = lds.arena;
    const int l = lane_id(), w = wave_id();
    if constexpr (wg_ncopy(NCH) == 4) {
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            red[w * np + 128 * k + 2 * l] = acc[2 * k];
            red[w * np + 128 * k + 2 * l + 1] = acc[2 * k + 1];
        }
        __syncthreads();
        for (int c = tid_here(); c < np; c += WG) post(c, red[c] + red[np + c] + red[2 * np + c] + red[3 * np + c]);
    } else {
        // two copies: waves 0 and 1 write, waves 2 and 3 add their partial sums to them (wave 2 to copy 0, wave 3 to copy 1)
        if (w < 2) {
#pragma unroll
            for (int k = 0; k < NCH; k++) { red[w * np + 128 * k + 2 * l] = acc[2 * k]; red[w * np + 128 * k + 2 * l + 1] = acc[2 * k + 1]; }
        }
        __syncthreads();
        if (w >= 2) {
#pragma unroll
            for (int k = 0; k < NCH; k++) { red[(w - 2) * np + 128 * k + 2 * l] += acc[2 * k]; red[(w - 2) * np + 128 * k + 2 * l + 1] += acc[2 * k + 1]; }
        }
        __syncthreads();
        for (int c = tid_here(); c < np; c += WG) post(c, red[c] + red[np + c]);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Symmetric mat-vecs, AXPY form:  out_m_v[c] = sum_{r<n} M_m[r][c] * v[r]   (M symmetric)
// Restates Utilities::AffineLinearTransformation / QuadraticFormProduct for the symmetric Q, C, Qk
// (src/Utilities.cpp:176-186, 214-225) as one sweep over the rows of up to two matrices with up to
// two vectors.  Any of M1, v1, outputs may be nullptr (workgroup-uniform).
// LDS: arena[0..4np) combine, arena[4np..6np) the vectors.
// ---------------------------------------------------------------------------------------------
// Rows in flight: a wave keeps ROWS_IN_FLIGHT rows (2 KiB each at np = 256) of every swept matrix outstanding before it
// consumes the oldest one.  The kernel is bound by the bytes it keeps in flight, not by issue (4 waves per SIMD, 84 % of the
// wave cycles parked on memory with one row per wave outstanding -- profiles/round1/final/pmc_sq.txt), so the sweeps
// load LCQP_DEPTH rows, then consume them.  Rows beyond n are read from the zero padding (np is a multiple of 128).
#ifndef LCQP_DEPTH
#define LCQP_DEPTH 4
#endif
static_assert(LCQP_DEPTH == 1 || LCQP_DEPTH == 2 || LCQP_DEPTH == 4 || LCQP_DEPTH == 8, "LCQP_DEPTH: the sweeps read NWAVE * LCQP_DEPTH rows per step and rely on that dividing the padded sizes (a build with 3, 5 or 6 reads past the matrix)");
// DCAP: at most this many rows in flight (a call site that keeps many registers live across the sweep passes 1)
template <int NCH, bool TWO_M, bool TWO_V, int DCAP = 8>
__device__ __forceinline__ void wg_symv_t(const double* __restrict__ M0, const double* __restrict__ M1, int n,
                        const double* __restrict__ v0, const double* __restrict__ v1,
                        double* o00, double* o10, double* o01, double* o11, Lds lds)
{
    constexpr int np = 128 * NCH;
    constexpr int D0 = (NCH > 8 || (TWO_M && NCH >= 4)) ? 1 : (TWO_M ? (LCQP_DEPTH >= 2 ? LCQP_DEPTH / 2 : 1) : LCQP_DEPTH);
    constexpr int D = D0 < DCAP ? D0 : DCAP;      // (np = 2048: a row is 8 KiB per wave already; the two-matrix sweep runs once per homotopy)
    static_assert((wg_ncopy(NCH) + 2) * np <= arena_doubles(NCH), "wg_symv: the partial copies and two staged vectors must fit the LDS arena");
    double* sv0 = lds.arena + wg_ncopy(NCH) * np;
    double* sv1 = lds.arena + (wg_ncopy(NCH) + 1) * np;
    for (int i = tid_here(); i < np; i += WG) {
        sv0[i] = (i < n) ? v0[i] : 0.0;
        sv1[i] = (TWO_V && i < n) ? v1[i] : 0.0;
    }
    __syncthreads();
    const int l = lane_id(), w = wave_id();
    double a00[2 * NCH], a10[2 * NCH], a01[2 * NCH], a11[2 * NCH];
#pragma unroll
    for (int k = 0; k < 2 * NCH; k++) a00[k] = a10[k] = a01[k] = a11[k] = 0.0;
    const int nR = min(np, ((n + NWAVE * D - 1) / (NWAVE * D)) * (NWAVE * D));   // rows n..nR-1: zero padding times a zero of sv
    for (int r0 = w; r0 < nR; r0 += NWAVE * D) {
        double2 m0[D][NCH], m1[D][NCH];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const double2* row0 = reinterpret_cast<const double2*>(M0 + (size_t)(r0 + NWAVE * d) * np) + l;
#pragma unroll
            for (int k = 0; k < NCH; k++) m0[d][k] = ld_stream(row0 + 64 * k);
            if (TWO_M) {
                const double2* row1 = reinterpret_cast<const double2*>(M1 + (size_t)(r0 + NWAVE * d) * np) + l;
#pragma unroll
                for (int k = 0; k < NCH; k++) m1[d][k] = ld_stream(row1 + 64 * k);
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            const double x0 = sv0[r0 + NWAVE * d], x1 = sv1[r0 + NWAVE * d];
#pragma unroll
            for (int k = 0; k < NCH; k++) {
                a00[2 * k] += m0[d][k].x * x0; a00[2 * k + 1] += m0[d][k].y * x0;
                if (TWO_V) { a01[2 * k] += m0[d][k].x * x1; a01[2 * k + 1] += m0[d][k].y * x1; }
                if (TWO_M) {
                    a10[2 * k] += m1[d][k].x * x0; a10[2 * k + 1] += m1[d][k].y * x0;
                    if (TWO_V) { a11[2 * k] += m1[d][k].x * x1; a11[2 * k + 1] += m1[d][k].y * x1; }
                }
            }
        }
    }
    if (o00) wg_combine<NCH>(a00, lds, [&](int c, double s) { o00[c] = s; });
    if (TWO_M && o10) wg_combine<NCH>(a10, lds, [&](int c, double s) { o10[c] = s; });
    if (TWO_V && o01) wg_combine<NCH>(a01, lds, [&](int c, double s) { o01[c] = s; });
    if (TWO_M && TWO_V && o11) wg_combine<NCH>(a11, lds, [&](int c, double s) { o11[c] = s; });
}

// the one-matrix, one-vector sweep (Q x of the QP residual, C x): any of the outputs may be nullptr
template <int NCH>
__device__ __forceinline__ void wg_symv(const double* __restrict__ M0, const double* __restrict__ M1, int n,
                        const double* __restrict__ v0, const double* __restrict__ v1,
                        double* o00, double* o10, double* o01, double* o11, Lds lds)
{
    if (M1 != nullptr && v1 != nullptr) wg_symv_t<NCH, true, true>(M0, M1, n, v0, v1, o00, o10, o01, o11, lds);
    else if (M1 != nullptr) wg_symv_t<NCH, true, false>(M0, M1, n, v0, nullptr, o00, o10, nullptr, nullptr, lds);
    else if (v1 != nullptr) wg_symv_t<NCH, false, true>(M0, nullptr, n, v0, v1, o00, nullptr, o01, nullptr, lds);
    else wg_symv_t<NCH, false, false>(M0, nullptr, n, v0, nullptr, o00, nullptr, nullptr, nullptr, lds);
}

// Euclidean norms of the m rows of an m x np row-major matrix (one wave per row, four rows in flight).  hotc / hotv (optional): rows with a
// single non-zero -- complementarity selectors, box rows, simple bounds written as constraints -- are noted with their column and value
// (hotc[r] = -1 for all others), so that the sweeps can make such a row up instead of reading np doubles of it (wg_rows).
template <int NCH>
__device__ __forceinline__ void wg_row_norms(const double* __restrict__ Mx, int m, double* out, int* hotc = nullptr, double* hotv = nullptr)
{
    constexpr int np = 128 * NCH;
    const int l = lane_id(), w = wave_id();
    for (int r0 = w; r0 < m; r0 += 4 * NWAVE) {
        double2 mm[4][NCH];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int r = r0 + NWAVE * d;
#pragma unroll
            for (int k = 0; k < NCH; k++) mm[d][k] = (r < m) ? (reinterpret_cast<const double2*>(Mx + (size_t)r * np) + l)[64 * k] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            double s2 = 0.0;
#pragma unroll
            for (int k = 0; k < NCH; k++) s2 += mm[d][k].x * mm[d][k].x + mm[d][k].y * mm[d][k].y;
            s2 = wave_sum(s2);
            const int r = r0 + NWAVE * d;
            if (l == 0 && r < m) out[r] = sqrt(s2);
            if (hotc && r < m) {
                int cnt = 0, col = -1; double val = 0.0;
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    if (mm[d][k].x != 0.0) { cnt++; col = 128 * k + 2 * l; val = mm[d][k].x; }
                    if (mm[d][k].y != 0.0) { cnt++; col = 128 * k + 2 * l + 1; val = mm[d][k].y; }
                }
                const int tot = wave_sum_i(cnt);
                if (tot == 1) { if (cnt == 1) { hotc[r] = col; hotv[r] = val; } }
                else if (l == 0) hotc[r] = -1;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Row sweep over an m x np row-major matrix (optionally through a row-index list):
//   dots[a]  = row_a . x                       (if x != nullptr)
//   out[c]   = post(c, sum_a coef[a] * row_a[c])   (if coef != nullptr)
// Restates Utilities::MatrixMultiplication (p=1) and TransponsedMatrixMultiplication (p=1),
// src/Utilities.cpp:38-47, 62-72 -- the latter without the reference's column-strided walk.
// Rows are dealt to waves in chunks of 16; lane j<16 of a wave carries the scalars of row j.
// LDS: arena[0..4np) combine, arena[4np..5np) x.
// ---------------------------------------------------------------------------------------------
// BYROW: dots[] and coef[] are indexed by the row number (idx[a]) instead of the list position a.
// KEEP: plain loads (the rows may stay in the caches) for a pass whose rows are read again within microseconds: the first of the two
// passes over the working-set rows of Et in a full correction.
template <int NCH, bool BYROW = false, bool KEEP = false, class Post>
__device__ __forceinline__ void wg_rows(const double* __restrict__ Mx, const int* __restrict__ idx, int m,
                        const double* __restrict__ x, double* dots,
                        const double* __restrict__ coef, Lds lds, Post post,
                        const int* __restrict__ hotc = nullptr, const double* __restrict__ hotv = nullptr)
{
    // hotc / hotv (wg_row_norms): a row with a single non-zero is made up from its column and value instead of being read -- the same
    // registers, the same sums, no bytes (bit-identical results; a typical working set is half complementarity selectors and box rows)
    constexpr int np = 128 * NCH;
    static_assert((wg_ncopy(NCH) + 1) * np <= arena_doubles(NCH), "wg_rows: the partial copies and the staged vector must fit the LDS arena");
    double* sx = lds.arena + wg_ncopy(NCH) * np;
    if (x) {
        for (int i = tid_here(); i < np; i += WG) sx[i] = x[i];
    }
    __syncthreads();
    const int l = lane_id(), w = wave_id();
    double xr[2 * NCH], acc[2 * NCH];
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        xr[2 * k] = x ? sx[128 * k + 2 * l] : 0.0;
        xr[2 * k + 1] = x ? sx[128 * k + 2 * l + 1] : 0.0;
        acc[2 * k] = acc[2 * k + 1] = 0.0;
    }
    const int nchunk = (m + 15) >> 4;
    constexpr int D = (NCH > 8) ? 1 : ((NCH == 4 && LCQP_DEPTH > 2) ? 2 : LCQP_DEPTH);     // rows a wave keeps in flight (see wg_symv_t; np = 512 at 128 VGPRs: two rows of 4 KiB, four spill)
    for (int ch = w; ch < nchunk; ch += NWAVE) {
        const int a0 = ch << 4;
        const int mya = a0 + l;
        const bool mine = (l < 16) && (mya < m);
        int myrow = mine ? (idx ? idx[mya] : mya) : -1;
        double mycoef = (mine && coef && (!BYROW || myrow >= 0)) ? coef[BYROW ? myrow : mya] : 0.0;
        const int myhc = (hotc && myrow >= 0) ? hotc[myrow] : -1;
        const double myhv = (myhc >= 0) ? hotv[myrow] : 0.0;
        double mydot = 0.0;
        const int cnt = min(16, m - a0);
        for (int j0 = 0; j0 < cnt; j0 += D) {
            double2 mm[D][NCH];
            int rows[D];
            double cfs[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                // lanes >= 16 hold row -1: entries beyond the chunk and padded list entries load nothing and contribute nothing
                rows[d] = wave_bcast_i(myrow, (j0 + d) & 63);
                cfs[d] = wave_bcast(mycoef, (j0 + d) & 63);
                const int hc = hotc ? wave_bcast_i(myhc, (j0 + d) & 63) : -1;
                if (hc >= 0) {
                    const double hv = wave_bcast(myhv, (j0 + d) & 63);
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
                        mm[d][k] = double2{0.0, 0.0};
                        if ((hc >> 7) == k && ((hc & 127) >> 1) == l) { if (hc & 1) mm[d][k].y = hv; else mm[d][k].x = hv; }
                    }
                } else if (rows[d] >= 0) {
                    const double2* rp = reinterpret_cast<const double2*>(Mx + (size_t)rows[d] * np) + l;
#pragma unroll
                    for (int k = 0; k < NCH; k++) mm[d][k] = KEEP ? rp[64 * k] : ld_stream(rp + 64 * k);
                } else {
#pragma unroll
                    for (int k = 0; k < NCH; k++) mm[d][k] = double2{0.0, 0.0};
                }
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                if (x) {
                    double dsum = 0.0;
#pragma unroll
                    for (int k = 0; k < NCH; k++) dsum += mm[d][k].x * xr[2 * k] + mm[d][k].y * xr[2 * k + 1];
                    dsum = wave_sum(dsum);
                    if (l == j0 + d) mydot = dsum;
                }
                if (coef && cfs[d] != 0.0) {
#pragma unroll
                    for (int k = 0; k < NCH; k++) { acc[2 * k] += cfs[d] * mm[d][k].x; acc[2 * k + 1] += cfs[d] * mm[d][k].y; }
                }
            }
        }
        if (mine && dots && (!BYROW || myrow >= 0)) dots[BYROW ? myrow : mya] = mydot;
    }
    if (coef) wg_combine<NCH>(acc, lds, post);
    else __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Triangular back-solve with a "symmetric-filled" Cholesky factor F (ld doubles per row,
// dimension 64*nblk): off-diagonal 64x64 blocks hold L (below) and L' (above); the diagonal blocks
// hold D = inv(L_II) below and D' above.  Forward (L y = b) and backward (L' x = b) both stream
// 64-wide row segments with lane-local accumulation; no per-row reductions, no sequential 64-step
// chains.   vec: global, in/out.   LDS: arena[0..nn) b, arena[nn..5nn) partials (nn <= LCQP_MAX_ACTIVE = 896).
// Algorithmic HBM bytes: 8*N*(N+2) for forward+backward.
// ---------------------------------------------------------------------------------------------
// TWO: two copies of the partial sums instead of four (the np = 4096 instantiation, wg_ncopy)
template <bool TWO = false>
__device__ __forceinline__ void wg_trsv(const double* __restrict__ F, int ld, int nblk, double* vec, bool forward, Lds lds)
{
    const int nn = 64 * nblk;
    double* b = lds.arena;
    double* red = lds.arena + nn;
    for (int i = tid_here(); i < nn; i += WG) b[i] = vec[i];
    __syncthreads();
    const int l = lane_id(), w = wave_id();
    for (int s = 0; s < nblk; s++) {
        const int I = forward ? s : nblk - 1 - s;
        {   // diagonal block: y_I[l] = sum_c Fd[c][l] * b_I[c], c<=l (forward, D' above) / c>=l (backward, D below)
            const double* Fd = F + (size_t)(64 * I) * ld + 64 * I;
            double f[16];
#pragma unroll
            for (int cc = 0; cc < 16; cc++) {      // only the triangle is fetched: lanes outside it issue no load (LCQP_TRSV_FULL_DIAG: all)
                const int c = 16 * w + cc;
#ifdef LCQP_TRSV_FULL_DIAG
                f[cc] = Fd[(size_t)c * ld + l];
#else
                f[cc] = 0.0;
                if (forward ? (c <= l) : (c >= l)) f[cc] = ld_stream(Fd + (size_t)c * ld + l);
#endif
            }
            double acc = 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; cc++) {
                const int c = 16 * w + cc;
                const bool use = forward ? (c <= l) : (c >= l);
                acc += use ? f[cc] * b[64 * I + c] : 0.0;
            }
            red[w * 64 + l] = acc;
        }
        __syncthreads();
        if (tid_here() < 64) b[64 * I + tid_here()] = red[l] + red[64 + l] + red[128 + l] + red[192 + l];
        __syncthreads();
        const int cb0 = forward ? I + 1 : 0, cb1 = forward ? nblk : I;
        if (cb1 > cb0) {
            // (TWO, the np = 4096 instantiation: two copies of the partial sums instead of four -- waves 0 and 1 write, waves 2 and 3 add)
            constexpr bool two = TWO;
            for (int pass = 0; pass < (two ? 2 : 1); pass++) {
                if (!two || (w >> 1) == pass)
                    for (int cb = cb0; cb < cb1; cb++) {
                        const double* Fp = F + (size_t)(64 * I + 16 * w) * ld + 64 * cb + l;
                        double f[16];
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) f[cc] = ld_stream(Fp + (size_t)cc * ld);
                        double acc = 0.0;
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) acc += f[cc] * b[64 * I + 16 * w + cc];
                        if (!two) red[w * nn + 64 * cb + l] = acc;
                        else if (pass == 0) red[(w & 1) * nn + 64 * cb + l] = acc;
                        else red[(w & 1) * nn + 64 * cb + l] += acc;
                    }
                __syncthreads();
            }
            for (int c = 64 * cb0 + tid_here(); c < 64 * cb1; c += WG)
                b[c] -= two ? (red[c] + red[nn + c]) : (red[c] + red[nn + c] + red[2 * nn + c] + red[3 * nn + c]);
            __syncthreads();
        }
    }
    for (int i = tid_here(); i < nn; i += WG) vec[i] = b[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// 64x64 output tiles are accumulated from k-major LDS panels As[16][TILE_PL], Bs[16][TILE_PL]
// (As[k][i] = A operand of output row i, Bs[k][j] = B operand of output column j).
// Default: fp64 matrix cores, v_mfma_f64_16x16x4_f64 -- wave w owns output rows 16w..16w+15 as four 16x16
// blocks; acc[a][b] is block a (columns 16a..16a+15), accumulator register b:
//     row = 16w + (lane>>4) + 4b,   col = 16a + (lane&15)            (C/D layout of the f64 MFMA)
// A operand: one f64 per lane, A[i = lane&15][k = lane>>4]; B operand: B[k = lane>>4][j = lane&15].
// -DLCQP_TILE_VALU selects the 4x4-per-thread v_fma_f64 micro-kernel instead (same peak rate on gfx950;
// kept as the cross-check of the MFMA lane maps): row = 4*(tid>>4) + a, col = 4*(tid&15) + b.
// ---------------------------------------------------------------------------------------------
constexpr int TILE_PL = 80;   // panel pitch in doubles: 160 dwords = 32 mod 64 banks -> conflict-free MFMA operand reads
typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tile_li(int a, int b)
{
#ifdef LCQP_TILE_VALU
    (void)b; return 4 * (tid_here() >> 4) + a;
#else
    (void)a; return 16 * wave_id() + (lane_id() >> 4) + 4 * b;
#endif
}
__device__ __forceinline__ int tile_lj(int a, int b)
{
#ifdef LCQP_TILE_VALU
    (void)a; return 4 * (tid_here() & 15) + b;
#else
    (void)b; return 16 * a + (lane_id() & 15);
#endif
}

// acc += As' * Bs over the 16 staged k values
__device__ __forceinline__ void tile_panel(double (&acc)[4][4], const double* As, const double* Bs)
{
#ifdef LCQP_TILE_VALU
    const int ty = tid_here() >> 4, tx = tid_here() & 15;
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        const double2 av0 = *reinterpret_cast<const double2*>(As + kk * TILE_PL + 4 * ty);
        const double2 av1 = *reinterpret_cast<const double2*>(As + kk * TILE_PL + 4 * ty + 2);
        const double2 bv0 = *reinterpret_cast<const double2*>(Bs + kk * TILE_PL + 4 * tx);
        const double2 bv1 = *reinterpret_cast<const double2*>(Bs + kk * TILE_PL + 4 * tx + 2);
        const double a[4] = {av0.x, av0.y, av1.x, av1.y};
        const double b[4] = {bv0.x, bv0.y, bv1.x, bv1.y};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] += a[i] * b[j];
    }
#else
    const int il = lane_id() & 15, kl = lane_id() >> 4, w = wave_id();
    d4_t c[4];
#pragma unroll
    for (int a = 0; a < 4; a++) c[a] = d4_t{acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
        const double av = As[(4 * k4 + kl) * TILE_PL + 16 * w + il];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const double bv = Bs[(4 * k4 + kl) * TILE_PL + 16 * a + il];
            c[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c[a], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) { acc[a][0] = c[a][0]; acc[a][1] = c[a][1]; acc[a][2] = c[a][2]; acc[a][3] = c[a][3]; }
#endif
}

// ---------------------------------------------------------------------------------------------
// 64x64 tile product  acc[i][j] = sum_{k<K} A(i)[k] * B(j)[k]   (NT form; K a multiple of 16).
// Row i of the A operand is  A + rowA(i)*lda  (rowA(i) < 0 -> zero row), same for B.
// acc[a][b] holds output (tile_li(a,b), tile_lj(a,b)).   LDS: arena[0 .. 2*16*TILE_PL).
// ---------------------------------------------------------------------------------------------
template <class RowA, class RowB>
__device__ __forceinline__ void wg_tile_nt(double (&acc)[4][4], const double* __restrict__ A, int lda, RowA rowA,
                                           const double* __restrict__ Bm, int ldb, RowB rowB, int K, Lds lds,
                                           int rowsValid = 64)
{
    // waves whose 16 output rows are all padding (>= rowsValid) skip the products (their A rows are zero)
    const bool active = 16 * wave_id() < rowsValid;
    constexpr int PL = TILE_PL;
    double* As = lds.arena;
    double* Bs = lds.arena + 16 * PL;
    const int t = tid_here();
    const int lr = t >> 2, kq = (t & 3) * 4;   // loader: row lr, k-quad kq
    const long ra = rowA(lr), rb = rowB(lr);
    const double* ap = (ra >= 0) ? A + (size_t)ra * lda + kq : nullptr;
    const double* bp = (rb >= 0) ? Bm + (size_t)rb * ldb + kq : nullptr;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    // software-pipelined: the operands of panel k0 + 16 are fetched into registers before the products of panel k0 start
    double2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0}, b0 = {0.0, 0.0}, b1 = {0.0, 0.0};
    if (ap && K > 0) { a0 = *reinterpret_cast<const double2*>(ap); a1 = *reinterpret_cast<const double2*>(ap + 2); }
    if (bp && K > 0) { b0 = *reinterpret_cast<const double2*>(bp); b1 = *reinterpret_cast<const double2*>(bp + 2); }
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();   // previous panel fully consumed
        As[(kq + 0) * PL + lr] = a0.x; As[(kq + 1) * PL + lr] = a0.y; As[(kq + 2) * PL + lr] = a1.x; As[(kq + 3) * PL + lr] = a1.y;
        Bs[(kq + 0) * PL + lr] = b0.x; Bs[(kq + 1) * PL + lr] = b0.y; Bs[(kq + 2) * PL + lr] = b1.x; Bs[(kq + 3) * PL + lr] = b1.y;
        __syncthreads();
        if (k0 + 16 < K) {
            if (ap) { a0 = *reinterpret_cast<const double2*>(ap + k0 + 16); a1 = *reinterpret_cast<const double2*>(ap + k0 + 18); }
            if (bp) { b0 = *reinterpret_cast<const double2*>(bp + k0 + 16); b1 = *reinterpret_cast<const double2*>(bp + k0 + 18); }
        }
        if (active) tile_panel(acc, As, Bs);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky of the SPD matrix held in the lower triangle of F (dimension
// 64*nblk, ld doubles per row), producing the symmetric-filled factor wg_trsv consumes.
//   tau > 0 : pivots <= tau * (original diagonal) mark linearly dependent rows (pivot := 1e150,
//             column := 0) -- the safeguarded factorisation of the active-row Gram matrix.
//   tau == 0: plain Cholesky; *info_fail is set when a pivot is not positive.
// dscr0 + J*dscrStride: 64*64 doubles of global scratch (dense copy of the J-th inverted diagonal block).
// d0  : nn doubles of global scratch (original diagonal), only used when tau > 0.
// src0 (tau == 0 only): the matrix to factor is src0 + shift0 I and F is output only -- the first block column reads src0 where it would read
// F (same values, same operations: the copy F = src0 + shift0 I the callers made before is gone, a pass over the matrix less).
// Returns the smallest pivot over rows < nreal (uniform).
// LDS: arena[0..64*65) tile + arena[4160..4224) flags + arena[4224..4496) 16x17 scratch.
// ---------------------------------------------------------------------------------------------
// -DLCQP_FACTOR_PROFILE (diagnostic build, tools/micro/factor_phases.py): thread 0 adds the shader clocks between the stamps to fp[0..6]
#ifdef LCQP_FACTOR_PROFILE
#define FPROF(k) do { if (fp) { const unsigned long long t_ = clock64(); fp[k] += t_ - fp[15]; fp[15] = t_; } } while (0)
#else
#define FPROF(k) do { } while (0)
#endif
__device__ __forceinline__ double wg_chol(double* F, int ld, int nblk, int nreal, double tau, double* dscr0, double* d0,
                          int* info_fail, Lds lds, int dscrStride, unsigned long long* fp = nullptr,
                          const double* src0 = nullptr, double shift0 = 0.0)
{
    const int t = tid_here();
    (void)fp;
    double* tile = lds.arena;
    double* dl = lds.arena + 64 * TILE_LD;
    // the smallest pivot and the failure flag are kept in LDS (dl[8], dl[9]; thread 0 writes them), not in registers around the loops
    if (t == 0) { dl[8] = INFINITY; dl[9] = 0.0; }
#if LCQP_SEQ_WAVE
    // Which wave runs the sequential stretches (the 16x16 sub-block chains below).  With wave 0 in every workgroup, the stretches of the
    // workgroups that share a CU queue up on ONE SIMD when the dispatcher has put every wave 0 there (k_factor: 0.40 ms with one workgroup
    // per two CUs, 1.30 ms with four per CU).  The waves say where they are (HW_REG_HW_ID: slot [3:0], SIMD [5:4]); the workgroup in slot k
    // of its CU takes its wave on SIMD k mod 4.  Any choice is correct; this one spreads the stretches when slots are handed out in order.
    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (lane_id() == 0) dl[10 + wave_id()] = (double)(hw & 0x3fu);
    }
#endif
    const int nn = 64 * nblk;
    if (tau > 0.0) {
        for (int i = t; i < nn; i += WG) d0[i] = F[(size_t)i * ld + i];
        __syncthreads();
    }
    for (int J = 0; J < nblk; J++) {
        const int t = tid_here();      // per block column: nothing derived from the thread number is carried around the loop
        const int o = 64 * J;
        double* dscr = dscr0 + (size_t)J * dscrStride;
        const bool first = (J == 0 && src0 != nullptr);      // the trailing matrix is still the input
        const double* S = first ? src0 : F;
        for (int e = t; e < 64 * 64; e += WG) {
            const int i = e >> 6, j = e & 63;
            tile[i * TILE_LD + j] = (j <= i) ? S[(size_t)(o + i) * ld + o + j] + ((first && i == j) ? shift0 : 0.0) : 0.0;
        }
        __syncthreads();
        FPROF(0);
        // Factor and invert the 64x64 tile in LDS with 16x16 sub-blocks.  Per sub-block column jb:
        //  (1) wave 0, lanes 0..15: row-per-lane Cholesky of the 16x16 diagonal sub-block in registers
        //      (wave-synchronous, v_readlane broadcasts, no barriers), then its inverse, column-per-lane;
        //  (2) all threads: panel rows below  L_rj = sum_q A_rq * Dd[j][q];
        //  (3) all threads: rank-16 update of the remaining lower triangle of the tile.
        // Afterwards the diagonal sub-blocks hold their inverses, the off-diagonal ones L; step (4) turns the
        // tile into inv(L) block column by block column:  D_ij = -D_ii * sum_{k=j}^{i-1} L_ik D_kj.
        double* tsc = dl + 64;                   // 16 x 17 scratch
        // padded rows (>= nreal, only in the safeguarded Gram factorisation) form an identity block: its
        // 16x16 sub-blocks are their own factors and inverses, nothing to do for them
        const int nsub = (tau > 0.0) ? min(4, max(0, (nreal - o + 15) >> 4)) : 4;
        for (int jb = 0; jb < nsub; jb++) {
            const int c0 = 16 * jb;
#if LCQP_SEQ_WAVE
            int sw = 0;
            { const int want = ((int)dl[10]) & 3;
              for (int w = 1; w < 4; w++) if (((((int)dl[10 + w]) >> 4) & 3) == want) sw = w;
              if (((((int)dl[10]) >> 4) & 3) == want) sw = 0; }
#else
            const int sw = 0;
#endif
            if ((t >> 6) == sw) {
                __builtin_amdgcn_s_setprio(3);   // the only sequential stretch: let it win issue slots from streaming waves
                const int l = t & 63;
                double minpiv = INFINITY;
                int fail = 0;
                double a[16], dcol[16];
                double d0v = (tau > 0.0 && l < 16) ? d0[o + c0 + l] : 0.0;
#pragma unroll
                for (int j = 0; j < 16; j++) a[j] = (l < 16 && j <= l) ? tile[(c0 + l) * TILE_LD + c0 + j] : 0.0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const double dk = wave_bcast(a[k], k);
                    bool dep;
                    if (tau > 0.0) dep = !(dk > tau * wave_bcast(d0v, k)) || !(dk > 0.0);
                    else { dep = false; if (!(dk > 0.0)) { fail = 1; dep = true; } }
                    if (o + c0 + k < nreal) minpiv = fmin(minpiv, dk);
                    const double ljj = dep ? 1e150 : sqrt(dk);
                    if (l == k) a[k] = ljj;
                    else if (l > k) a[k] = dep ? 0.0 : a[k] / ljj;
#pragma unroll
                    for (int j = k + 1; j < 16; j++) {
                        const double ljk = wave_bcast(a[k], j);
                        if (l >= j) a[j] -= a[k] * ljk;
                    }
                }
                // inverse of the 16x16 lower-triangular block: lane c builds column c by forward substitution
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < i; k++) sacc += wave_bcast(a[k], i) * dcol[k];
                    const double lii = wave_bcast(a[i], i);
                    dcol[i] = (l == i) ? 1.0 / lii : ((l < i) ? -sacc / lii : 0.0);
                }
                if (l < 16) {
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (i >= l) tile[(c0 + i) * TILE_LD + c0 + l] = dcol[i];   // Dd[i][l]
                }
                if (l == 0) { dl[8] = fmin(dl[8], minpiv); if (fail) dl[9] = 1.0; }
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
            FPROF(1);
            const int nrem = 16 * nsub - (c0 + 16);     // (non-padded) rows below the diagonal sub-block
            if (nrem > 0) {
                // (2) panel: outputs (r, j), r in [c0+16, 64), j in [0,16)
                double pv[3];
#pragma unroll
                for (int q3 = 0; q3 < 3; q3++) {
                    const int e = t + WG * q3;
                    pv[q3] = 0.0;
                    if (e < nrem * 16) {
                        const int r = c0 + 16 + (e >> 4), j = e & 15;
                        double sacc = 0.0;
                        for (int q = 0; q <= j; q++) sacc += tile[r * TILE_LD + c0 + q] * tile[(c0 + j) * TILE_LD + c0 + q];
                        pv[q3] = sacc;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q3 = 0; q3 < 3; q3++) {
                    const int e = t + WG * q3;
                    if (e < nrem * 16) tile[(c0 + 16 + (e >> 4)) * TILE_LD + c0 + (e & 15)] = pv[q3];
                }
                __syncthreads();
                // (3) trailing update of rows/cols >= c0+16 (lower triangle)
                for (int e = t; e < nrem * nrem; e += WG) {
                    const int ri = e / nrem, ci = e - ri * nrem;
                    if (ci <= ri) {
                        const int r = c0 + 16 + ri, cc = c0 + 16 + ci;
                        double sacc = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; q++) sacc += tile[r * TILE_LD + c0 + q] * tile[cc * TILE_LD + c0 + q];
                        tile[r * TILE_LD + cc] -= sacc;
                    }
                }
                __syncthreads();
                FPROF(2);
            }
        }
        // (4) blocked inversion: thread (r, c) of a 16x16 block
        {
            const int r = t >> 4, cidx = t & 15;
            for (int jb = 0; jb + 1 < nsub; jb++)
                for (int ib = jb + 1; ib < nsub; ib++) {
                    double tv = 0.0;
                    for (int kb = jb; kb < ib; kb++) {
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            tv += tile[(16 * ib + r) * TILE_LD + 16 * kb + q] * tile[(16 * kb + q) * TILE_LD + 16 * jb + cidx];
                    }
                    tsc[r * 17 + cidx] = tv;
                    __syncthreads();
                    double dv = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; q++) dv -= tile[(16 * ib + r) * TILE_LD + 16 * ib + q] * tsc[q * 17 + cidx];
                    __syncthreads();
                    tile[(16 * ib + r) * TILE_LD + 16 * jb + cidx] = dv;
                    __syncthreads();
                }
        }
        FPROF(3);
        // write D: symmetric fill into F_JJ, dense lower copy into dscr
        for (int e = t; e < 64 * 64; e += WG) {
            const int i = e >> 6, j = e & 63;
            const double v = (j <= i) ? tile[i * TILE_LD + j] : tile[j * TILE_LD + i];
            F[(size_t)(o + i) * ld + o + j] = v;
            dscr[e] = (j <= i) ? v : 0.0;
        }
        __syncthreads();
        FPROF(4);
        // panel:  L_IJ = A_IJ * D'   (rows of block I, k over block J)
        for (int I = J + 1; I < nblk; I++) {
            double acc[4][4];
            wg_tile_nt(acc, S + (size_t)(64 * I) * ld + o, ld, [](int r) { return (long)r; },
                       dscr, 64, [](int r) { return (long)r; }, 64, lds, tau > 0.0 ? nreal - 64 * I : 64);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int gi = 64 * I + tile_li(i, j), gj = o + tile_lj(i, j);
                    F[(size_t)gi * ld + gj] = acc[i][j];
                    F[(size_t)gj * ld + gi] = acc[i][j];
                }
            __syncthreads();
        }
        FPROF(5);
        // trailing update:  A_IK -= L_IJ * L_KJ'   for I >= K > J
        for (int I = J + 1; I < nblk; I++)
            for (int Kb = J + 1; Kb <= I; Kb++) {
                double acc[4][4];
                wg_tile_nt(acc, F + (size_t)(64 * I) * ld + o, ld, [](int r) { return (long)r; },
                           F + (size_t)(64 * Kb) * ld + o, ld, [](int r) { return (long)r; }, 64, lds,
                           tau > 0.0 ? nreal - 64 * I : 64);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int gi = 64 * I + tile_li(i, j), gj = 64 * Kb + tile_lj(i, j);
                        if (I != Kb || gj <= gi) {
                            if (first) F[(size_t)gi * ld + gj] = (S[(size_t)gi * ld + gj] + (gi == gj ? shift0 : 0.0)) - acc[i][j];
                            else F[(size_t)gi * ld + gj] -= acc[i][j];
                        }
                    }
                __syncthreads();
            }
        FPROF(6);
    }
    __syncthreads();
    const double minpiv = dl[8];
    if (info_fail && dl[9] != 0.0) *info_fail = 1;   // benign same-value race
    __syncthreads();
    return uniform_d(minpiv);
}

// ---------------------------------------------------------------------------------------------
// 64x64 tile product, TN form:  acc[i][j] = sum_{r<nrows} wgt(r) * A[r][ca+i] * B[r][cb+j]
template <class Wgt>
__device__ __forceinline__ void wg_tile_tn(double (&acc)[4][4], const double* __restrict__ A, int lda, int ca,
                                           const double* __restrict__ Bm, int ldb, int cb, int nrows, Wgt wgt, Lds lds)
{
    constexpr int PL = TILE_PL;
    double* As = lds.arena;
    double* Bs = lds.arena + 16 * PL;
    const int t = tid_here();
    const int kk = t >> 4, c4 = (t & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    // software-pipelined: the operands of panel k0 + 16 are fetched into registers before the products of panel k0 start
    double2 a0, a1, b0, b1;
    double wv = 0.0;
    auto fetch = [&](int r) {
        a0 = a1 = b0 = b1 = double2{0.0, 0.0};
        wv = 0.0;
        if (r < nrows) {
            wv = wgt(r);
            const double* ap = A + (size_t)r * lda + ca + c4;
            const double* bp = Bm + (size_t)r * ldb + cb + c4;
            a0 = *reinterpret_cast<const double2*>(ap); a1 = *reinterpret_cast<const double2*>(ap + 2);
            b0 = *reinterpret_cast<const double2*>(bp); b1 = *reinterpret_cast<const double2*>(bp + 2);
        }
    };
    fetch(kk);
    for (int k0 = 0; k0 < nrows; k0 += 16) {
        a0.x *= wv; a0.y *= wv; a1.x *= wv; a1.y *= wv;
        __syncthreads();
        *reinterpret_cast<double2*>(As + kk * PL + c4) = a0; *reinterpret_cast<double2*>(As + kk * PL + c4 + 2) = a1;
        *reinterpret_cast<double2*>(Bs + kk * PL + c4) = b0; *reinterpret_cast<double2*>(Bs + kk * PL + c4 + 2) = b1;
        __syncthreads();
        if (k0 + 16 < nrows) fetch(k0 + 16 + kk);
        tile_panel(acc, As, Bs);
    }
    __syncthreads();
}

// Workgroup ids are handed to the 8 XCDs round robin (id % 8 labels the workgroups that share an L2): the tiles of ONE instance read the same
// operands, so every XCD gets a contiguous range of logical ids (bijective for any grid; cdna_hip_programming.md §5.5 T1).  A speed choice only.
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg)
{
#ifdef LCQP_NO_XCD_REMAP
    (void)nwg; return bid;
#else
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
#endif
}

// lower-triangular tile index -> (I, J), I >= J
__device__ __forceinline__ void tri_tile(int tIdx, int& I, int& J)
{
    I = 0;
    while ((I + 1) * (I + 2) / 2 <= tIdx) I++;
    J = tIdx - I * (I + 1) / 2;
}

}  // namespace lcqp
