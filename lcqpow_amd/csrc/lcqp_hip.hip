// lcqp_hip.hip -- kernels and C-ABI of liblcqpow_hip.so (gfx950 only; see include/lcqp_hip.h).
#include "lcqp_dev.hpp"
#include "lcqp_launch.hpp"
#include "../../include/lcqp_synth.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace lcqp;


// =================================================================================================
// device kernels: the per-size ones live in lcqp_kernels.hpp / lcqp_nch.hip; here only the two that are not templated
// =================================================================================================
#define LCQP_LDS_N(NCHV)                                    \
    __shared__ double sh_arena[arena_doubles(NCHV)];        \
    __shared__ double sh_red[16];                           \
    __shared__ int sh_ired[16];                             \
    Lds lds{sh_arena, sh_red, sh_ired};
#define LCQP_LDS LCQP_LDS_N(4)

__global__ __launch_bounds__(WG) void k_chol(int np, int nblk, int n, double* F, double* dscr, int* fail)
{
    LCQP_LDS
    const int b = blockIdx.x;
    wg_chol(F + (size_t)b * np * np, np, nblk, n, 0.0, dscr + (size_t)b * 4096, nullptr, fail + b, lds, 0);
}

__global__ __launch_bounds__(WG, 4) void k_backsolve(int np, int nblk, const double* F, const double* rhs, double* x)
{
    LCQP_LDS
    const int b = blockIdx.x;
    double* xv = x + (size_t)b * np;
    for (int i = threadIdx.x; i < np; i += WG) xv[i] = rhs[(size_t)b * np + i];
    __syncthreads();
    wg_trsv(F + (size_t)b * np * np, np, nblk, xv, true, lds);
    wg_trsv(F + (size_t)b * np * np, np, nblk, xv, false, lds);
}

// =================================================================================================
// host side
// =================================================================================================
// HIP maps the streams of a process onto a few hardware queues -- 4 unless GPU_MAX_HW_QUEUES says otherwise -- and every batch object has
// two streams: a process that keeps three batch objects alive can find both slots of a BatchPipeline on ONE queue, and its batches then
// run one after the other (tools/micro/pipeline_check.py: 35 200 LCQPs/s with two objects alive, 30 750 with an idle third one, 34 800
// again with eight queues).  The library does NOT touch the environment by itself (round 6; it used to, from a constructor: that changed
// the queue count of every GPU user of the host process).  A program that wants the queues asks for them explicitly, before the first
// HIP call of the process -- the runtime reads the variable once, when it initialises: bench.py and the examples do.
// Returns 0 when the variable was set, 1 when the caller's environment already holds a value (left alone), LCQP_HIP_ERROR on a bad count.
extern "C" int lcqp_hip_request_hw_queues(int n)
{
    if (n < 1 || n > 64) return LCQP_HIP_ERROR;
    if (getenv("GPU_MAX_HW_QUEUES")) return 1;
    char buf[16];
    snprintf(buf, sizeof buf, "%d", n);
    return setenv("GPU_MAX_HW_QUEUES", buf, /*overwrite=*/0) == 0 ? 0 : LCQP_HIP_ERROR;
}

static thread_local std::string g_err;
static int set_err(const char* what, hipError_t e)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return LCQP_HIP_ERROR;
}
#define HIPCHK(call)                                               \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return set_err(#call, e_);           \
    } while (0)
#define HIPCHKN(call)                                              \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) { set_err(#call, e_); return nullptr; } \
    } while (0)

extern "C" const char* lcqp_hip_last_error(void) { return g_err.c_str(); }
extern "C" int lcqp_hip_device_count(void)
try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" void lcqp_hip_options_default(lcqp_options_t* o)
{   // src/Options.cpp:296-333
    memset(o, 0, sizeof(*o));
    const double EPS = 2.221e-16;   // include/Utilities.hpp:350
    o->complementarityTolerance = 1.0e3 * EPS;
    o->stationarityTolerance = 1.0e6 * EPS;
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
    o->admmRho = 0.1; o->admmSigma = 1e-6; o->admmAlpha = 1.6; o->rhoEqMult = 1e3;
    o->proxSmall = 1e-12; o->proxBig = 1e-8; o->pivotThreshold = 1e-7; o->depTau = 1e-12;
    o->feasTol = 1e-9; o->resTol = 1e-12;
    o->admmFirst = 0; o->admmHot = 0; o->maxTrials = 16; o->maxRounds = 40;      // maxTrials: 12 until round 3 -- cold starts of the synthetic workload need up to 14 trials, and a polish that runs out of trials costs an ADMM round (the factor L_K, ten iterations, a second cold polish): those instances were the tail of the launch
}

struct lcqp_hip_batch {
    DevBatch db;
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1, ev2;
    // the setup has two independent branches (C = L'R + R'L and its compression; L1 -> Et -> M): the short one runs beside the long one
    hipStream_t side;
    hipEvent_t evFork, evJoin;
    int numCU;
    bool overlapped;      // lcqp_hip_batch_set_overlapped
    // two pinned staging slots for loadLCQP: instance k is packed into slot k&1 while slot (k-1)&1 is in flight
    void* stage[2];
    hipEvent_t stageDone[2];
    size_t stageBytes;
    std::vector<void*> allocs;
    bool setupDone, ran, anyLoaded;
    int nch;
    size_t bytesTotal;
};

// -DLCQP_ONLY_NCH=k (experiment builds, tools/gpu_ab.py): link only the kernels of one padded size
// padded size of a problem with n variables in units of 128: 1, 2, 3, 4, then 8 (np = 1024), 16 (np = 2048) and 32 (np = 4096)
static inline int padded_nch(int n) { const int k = (n + 127) / 128; return k > 16 ? 32 : (k > 8 ? 16 : (k > 4 ? 8 : k)); }

static void lcqp_dispatch(int nch, int kid, int grid, hipStream_t s, const LaunchArgs& a)
{
#ifdef LCQP_ONLY_NCH
    (void)nch;
#define LCQP_CAT2(a, b) a##b
#define LCQP_CAT(a, b) LCQP_CAT2(a, b)
    LCQP_CAT(lcqp_launch_, LCQP_ONLY_NCH)(kid, grid, s, a);
#else
    switch (nch) {
        case 1: lcqp_launch_1(kid, grid, s, a); break;
        case 2: lcqp_launch_2(kid, grid, s, a); break;
        case 3: lcqp_launch_3(kid, grid, s, a); break;
        case 4: lcqp_launch_4(kid, grid, s, a); break;
        case 8: lcqp_launch_8(kid, grid, s, a); break;
        case 16: lcqp_launch_16(kid, grid, s, a); break;
        default: lcqp_launch_32(kid, grid, s, a); break;
    }
#endif
}
// the second build of the persistent kernels (256 registers): np <= 512 (36 KB of LDS per workgroup) and at most three workgroups per CU
static bool lcqp_dispatch_few(int nch, int kid, int grid, hipStream_t s, const LaunchArgs& a)
{
#ifdef LCQP_ONLY_NCH
#if LCQP_ONLY_NCH <= 4
    (void)nch;
    LCQP_CAT(lcqp_launch_few_, LCQP_ONLY_NCH)(kid, grid, s, a);
    return true;
#else
    (void)nch; (void)kid; (void)grid; (void)s; (void)a;
    return false;
#endif
#else
    if (nch == 1) lcqp_launch_few_1(kid, grid, s, a);
    else if (nch == 2) lcqp_launch_few_2(kid, grid, s, a);
    else if (nch == 3) lcqp_launch_few_3(kid, grid, s, a);
    else if (nch == 4) lcqp_launch_few_4(kid, grid, s, a);
    else return false;
    return true;
#endif
}

static void dispatch_db(lcqp_hip_batch* h, int kid, int grid, const int* list = nullptr, int initial = 0, uint64_t seed0 = 0, uint64_t first = 0, hipStream_t on = nullptr)
{
    LaunchArgs a;
    a.db = h->db; a.list = list; a.initial = initial; a.seed0 = seed0; a.first = first;
    if ((kid == ID_k_lcqp_run || kid == ID_k_qp_solve) && h->nch <= 4 && h->db.B <= 3 * h->numCU && !h->overlapped
        && lcqp_dispatch_few(h->nch, kid, grid, on ? on : h->stream, a)) return;
    lcqp_dispatch(h->nch, kid, grid, on ? on : h->stream, a);
}

template <class T>
static int dev_alloc(lcqp_hip_batch* h, T** p, size_t count, bool zero)
{
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIPCHK(hipMalloc(&q, bytes));
    h->allocs.push_back(q);
    h->bytesTotal += bytes;
    if (zero) HIPCHK(hipMemsetAsync(q, 0, bytes, h->stream));
    *p = (T*)q;
    return 0;
}

extern "C" lcqp_hip_batch_t* lcqp_hip_batch_create(int batch, int nV, int nC, int nComp, int withBox, int device)
try {
    if (batch <= 0 || nV <= 0 || nC < 0 || nComp < 0) { g_err = "invalid dimensions"; return nullptr; }
    if (nV > 4096) { g_err = "nV > 4096 is not supported by the dense kernels of this build (padded sizes 128 ... 4096; the sparse engine takes larger banded / bordered problems)"; return nullptr; }
    HIPCHKN(hipSetDevice(device));
    lcqp_hip_batch* h = new (std::nothrow) lcqp_hip_batch();
    if (!h) { g_err = "out of host memory"; return nullptr; }
    h->device = device; h->setupDone = false; h->ran = false; h->anyLoaded = false; h->bytesTotal = 0;
    h->stage[0] = h->stage[1] = nullptr; h->stageBytes = 0;
    h->stream = nullptr; h->ev0 = h->ev1 = h->ev2 = nullptr; h->side = nullptr; h->evFork = h->evJoin = nullptr;
    h->numCU = 256; h->overlapped = false;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) h->numCU = cu; }
    DevBatch& d = h->db;
    memset(&d, 0, sizeof(d));
    hipError_t e0 = hipStreamCreate(&h->stream);
    if (e0 == hipSuccess) e0 = hipEventCreate(&h->ev0);
    if (e0 == hipSuccess) e0 = hipEventCreate(&h->ev1);
    if (e0 == hipSuccess) e0 = hipEventCreate(&h->ev2);
    if (e0 == hipSuccess) e0 = hipStreamCreate(&h->side);
    if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming);
    if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming);
    if (e0 != hipSuccess) { set_err("stream/event creation", e0); lcqp_hip_batch_destroy(h); return nullptr; }
    d.B = batch; d.n = nV; d.nC = nC; d.nComp = nComp; d.mA = nC + 2 * nComp;
    h->nch = padded_nch(nV);      // 512 < nV <= 1024 runs the np = 1024 instantiation; 1024 < nV <= 2048: np = 2048 (96 KiB of LDS, one workgroup per CU, one row in flight per wave)
    d.np = 128 * h->nch;
    d.nblk = d.np / 64;
    d.boxcap = withBox ? nV : 0;
    d.mEcap = d.mA + d.boxcap;
    if (d.mEcap < 1) d.mEcap = 1;
    int capNa = 2 * nV > 64 ? 2 * nV : 64;      // active rows the Gram factor has room for (qp_polish)
    if (capNa > d.mEcap) capNa = d.mEcap;
    if (capNa > max_active(h->nch)) capNa = max_active(h->nch);
    d.capS = ((capNa + 63) / 64) * 64;
    if (d.capS < 64) d.capS = 64;
    d.nd = nV + d.mA;
    lcqp_hip_options_default(&d.opt);
    const size_t B = batch, np = d.np, mE = d.mEcap;
    int rc = 0;
    rc |= dev_alloc(h, &d.Q, B * np * np, true);
    rc |= dev_alloc(h, &d.C, B * np * np, true);
    rc |= dev_alloc(h, &d.E, B * mE * np, true);
    rc |= dev_alloc(h, &d.Et, B * mE * np, true);
    rc |= dev_alloc(h, &d.F1, B * np * np, true);
    rc |= dev_alloc(h, &d.FK, B * np * np, true);
    rc |= dev_alloc(h, &d.S, B * (size_t)d.capS * d.capS, true);
    d.mMld = ((d.mEcap + 63) / 64) * 64;
    rc |= dev_alloc(h, &d.MM, B * (size_t)d.mMld * d.mMld, true);
    rc |= dev_alloc(h, &d.crow, B * (size_t)d.capS, true);
    d.capC = 8 * d.np;                                     // C goes into compressed rows when it has at most 8 non-zeros per row on average
    rc |= dev_alloc(h, &d.Cp, B * (np + 1), true);
    rc |= dev_alloc(h, &d.Ci, B * (size_t)d.capC, true);
    rc |= dev_alloc(h, &d.Cv, B * (size_t)d.capC, true);
    rc |= dev_alloc(h, &d.S2, B * (size_t)d.capS * d.capS, true);
    rc |= dev_alloc(h, &d.DS, B * (size_t)(d.capS / 64) * 4096, true);
    rc |= dev_alloc(h, &d.D1, B * (size_t)d.nblk * 4096, true);
    rc |= dev_alloc(h, &d.dscr, B * 4096, true);
    rc |= dev_alloc(h, &d.nv, B * V_NUM * np, true);
    rc |= dev_alloc(h, &d.mv, B * M_NUM * mE, true);
    rc |= dev_alloc(h, &d.sv, B * S_NUM * (size_t)d.capS, true);
    rc |= dev_alloc(h, &d.mi, B * I_NUM * mE, true);
    rc |= dev_alloc(h, &d.idx, B * (size_t)d.capS, true);
    rc |= dev_alloc(h, &d.boxidx, B * np, true);
    rc |= dev_alloc(h, &d.lbL, B * (size_t)(nComp ? nComp : 1), true);
    rc |= dev_alloc(h, &d.lbR, B * (size_t)(nComp ? nComp : 1), true);
    rc |= dev_alloc(h, &d.yk, B * (size_t)d.nd, true);
    rc |= dev_alloc(h, &d.y0, B * (size_t)d.nd, true);
    rc |= dev_alloc(h, &d.xout, B * (size_t)nV, true);
    rc |= dev_alloc(h, &d.yout, B * (size_t)d.nd, true);
    rc |= dev_alloc(h, &d.stats, B, true);
    rc |= dev_alloc(h, &d.info, B, true);
    rc |= dev_alloc(h, &d.prof, B * 16, true);
    if (rc != 0 || hipStreamSynchronize(h->stream) != hipSuccess) { lcqp_hip_batch_destroy(h); return nullptr; }
    return h;
}
catch (...) { g_err = "out of host memory"; return nullptr; }   // nothing throws across the C boundary

extern "C" void lcqp_hip_batch_destroy(lcqp_hip_batch_t* h)
try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->allocs) (void)hipFree(p);
    for (int k = 0; k < 2; k++) if (h->stage[k]) { (void)hipHostFree(h->stage[k]); (void)hipEventDestroy(h->stageDone[k]); }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}
catch (...) { }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_set_options(lcqp_hip_batch_t* h, const lcqp_options_t* opt)
try {
    if (!h || !opt) return LCQP_INVALID_ARGUMENT;
    if (opt->nDynamicPenalty > 64) { g_err = "nDynamicPenalty > 64 unsupported"; return LCQP_HIP_UNSUPPORTED; }
    {
        // tracking vectors of OutputStatistics (src/OutputStatistics.cpp:131-164), first 1024 iterates; the buffers are sized
        // for the largest maxIterations seen with storeSteps on and grow when a later setOptions raises it
        DevBatch& d = h->db;
        const int want = opt->maxIterations + 1 < 1024 ? opt->maxIterations + 1 : 1024;
        const int have = d.traceCap < 0 ? -d.traceCap : d.traceCap;
        if (opt->storeSteps && have < want) {
            HIPCHK(hipSetDevice(h->device));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (void* old : {(void*)d.traceS, (void*)d.traceX})      // a regrow frees the smaller buffers
                if (old) { (void)hipFree(old); h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), old), h->allocs.end()); }
            d.traceS = d.traceX = nullptr; d.traceCap = 0;
            if (dev_alloc(h, &d.traceS, (size_t)d.B * want * 8, true) || dev_alloc(h, &d.traceX, (size_t)d.B * want * d.n, true)) return LCQP_HIP_ERROR;
            if (!d.traceLen && dev_alloc(h, &d.traceLen, (size_t)d.B, true)) return LCQP_HIP_ERROR;
            d.traceCap = want;
        } else if (opt->storeSteps) d.traceCap = have;
        else if (have > 0) d.traceCap = -have;                    // keep the buffers, stop recording
    }
    h->db.opt = *opt;
    h->setupDone = false;   // rho / sigma / prox weights enter the factorisations
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_get_trace(lcqp_hip_batch_t* h, int instance, int cap, double* scalars, double* x, int* len)
try {
    if (!h || instance < 0 || instance >= h->db.B || !len) return LCQP_INVALID_ARGUMENT;
    DevBatch& d = h->db;
    *len = 0;
    if (d.traceCap <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int n = 0;
    HIPCHK(hipMemcpy(&n, d.traceLen + instance, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    if (scalars && n) HIPCHK(hipMemcpy(scalars, d.traceS + (size_t)instance * d.traceCap * 8, sizeof(double) * 8 * n, hipMemcpyDeviceToHost));
    if (x && n) HIPCHK(hipMemcpy(x, d.traceX + (size_t)instance * d.traceCap * d.n, sizeof(double) * (size_t)d.n * n, hipMemcpyDeviceToHost));
    *len = n;
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// diagnostic builds (-DLCQP_PROFILE): per-instance cycle counters of the megakernel's phases, [B][16]
extern "C" int lcqp_hip_batch_read_profile(lcqp_hip_batch_t* h, unsigned long long* out)
try {
    if (!h || !out) return LCQP_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->db.prof, sizeof(unsigned long long) * (size_t)h->db.B * 16, hipMemcpyDeviceToHost));
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" void* lcqp_hip_batch_stream(lcqp_hip_batch_t* h) { return h ? (void*)h->stream : nullptr; }

static inline double bnd(const double* p, size_t i, double dflt) { return p ? p[i] : dflt; }

extern "C" int lcqp_hip_batch_load(lcqp_hip_batch_t* h, int first, int count,
                                   const double* Q, const double* g, const double* L, const double* R,
                                   const double* lbL, const double* ubL, const double* lbR, const double* ubR,
                                   const double* A, const double* lbA, const double* ubA,
                                   const double* lb, const double* ub, const double* x0, const double* y0)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    DevBatch& d = h->db;
    const int n = d.n, nC = d.nC, nComp = d.nComp, mA = d.mA, np = d.np, mE = d.mEcap;
    if (first < 0 || count <= 0 || first + count > d.B) return LCQP_INVALID_ARGUMENT;
    if (!Q) return LCQP_INVALID_ARGUMENT;
    if (!g) return LCQP_INVALID_OBJECTIVE_LINEAR_TERM;               // include/LCQProblem.ipp:44-45
    if (!A && nC > 0) return LCQP_INVALID_CONSTRAINT_MATRIX;         // src/LCQProblem.cpp:569-570
    if (!L || !R) return LCQP_INVALID_COMPLEMENTARITY_MATRIX;        // :611-612
    if ((lb || ub) && d.boxcap == 0) { g_err = "batch was created without box-bound capacity"; return LCQP_INVALID_ARGUMENT; }
    HIPCHK(hipSetDevice(h->device));
    // A batch may mix instances with and without lbL / lbR (round 6): an absent bound vector is the zero vector (setComplementarityBounds
    // :726-785), and the phi expressions of :969-996 with zeros -- phi_const = 0, g_phi = 0, g_tilde = g + rho 0 -- are the arithmetic of an
    // instance loaded without them, bit for bit.  The batch-wide flag only says whether ANY instance carries bounds, i.e. whether the
    // kernels read the (zero-filled) arrays at all; a (re)load starting at instance 0, or the first load of the object, starts it over.
    const int hasL = lbL ? 1 : 0, hasR = lbR ? 1 : 0;
    if (!h->anyLoaded || first == 0) { d.hasLbL = hasL; d.hasLbR = hasR; h->anyLoaded = true; }
    else { d.hasLbL |= hasL; d.hasLbR |= hasR; }
    // pinned staging: [Qp | Ep | nvb | mvb | ybuf | lbuf | rbuf | info | bidx]
    const size_t nQ = (size_t)np * np, nE = (size_t)mE * np, nNV = (size_t)V_NUM * np, nMV = (size_t)M_NUM * mE;
    const size_t nY = (size_t)d.nd, nLR = (size_t)(nComp ? nComp : 1);
    const size_t infoDbl = (sizeof(InstInfo) + 7) / 8, bidxDbl = ((size_t)np * sizeof(int) + 7) / 8;
    const size_t slotBytes = sizeof(double) * (nQ + nE + nNV + nMV + nY + 2 * nLR + infoDbl + bidxDbl);
    if (h->stageBytes < slotBytes) {
        for (int k = 0; k < 2; k++) {
            if (h->stage[k]) { (void)hipHostFree(h->stage[k]); (void)hipEventDestroy(h->stageDone[k]); h->stage[k] = nullptr; }
            HIPCHK(hipHostMalloc(&h->stage[k], slotBytes, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&h->stageDone[k], hipEventDisableTiming));
            HIPCHK(hipEventRecord(h->stageDone[k], h->stream));
        }
        h->stageBytes = slotBytes;
    }
    for (int k = 0; k < count; k++) {
        const size_t b = (size_t)first + k;
        const int slot = k & 1;
        HIPCHK(hipEventSynchronize(h->stageDone[slot]));            // the copies that last used this slot are done
        double* Qp = (double*)h->stage[slot];
        double* Ep = Qp + nQ; double* nvb = Ep + nE; double* mvb = nvb + nNV; double* ybuf = mvb + nMV;
        double* lbuf = ybuf + nY; double* rbuf = lbuf + nLR;
        InstInfo* info = (InstInfo*)(rbuf + nLR);
        int* bidx = (int*)((double*)info + infoDbl);
        memset(Qp, 0, slotBytes);
        for (int i = 0; i < n; i++) memcpy(&Qp[(size_t)i * np], Q + ((size_t)k * n + i) * n, sizeof(double) * n);       // setQ .ipp:27-36
        // setConstraints :563-626: stack [A; L; R]
        for (int r = 0; r < nC; r++) memcpy(&Ep[(size_t)r * np], A + ((size_t)k * nC + r) * n, sizeof(double) * n);
        for (int r = 0; r < nComp; r++) {
            memcpy(&Ep[(size_t)(nC + r) * np], L + ((size_t)k * nComp + r) * n, sizeof(double) * n);
            memcpy(&Ep[(size_t)(nC + nComp + r) * np], R + ((size_t)k * nComp + r) * n, sizeof(double) * n);
        }
        double* lE = &mvb[(size_t)M_L * mE];
        double* uE = &mvb[(size_t)M_U * mE];
        for (int r = 0; r < nC; r++) { lE[r] = bnd(lbA, (size_t)k * nC + r, -INFINITY); uE[r] = bnd(ubA, (size_t)k * nC + r, INFINITY); }
        // setComplementarityBounds :726-785
        for (int i = 0; i < nComp; i++) {
            if (lbL && lbL[(size_t)k * nComp + i] <= -INFINITY) return LCQP_INVALID_LOWER_COMPLEMENTARITY_BOUND;
            if (lbR && lbR[(size_t)k * nComp + i] <= -INFINITY) return LCQP_INVALID_LOWER_COMPLEMENTARITY_BOUND;
            lE[nC + i] = bnd(lbL, (size_t)k * nComp + i, 0.0);
            uE[nC + i] = bnd(ubL, (size_t)k * nComp + i, INFINITY);
            lE[nC + nComp + i] = bnd(lbR, (size_t)k * nComp + i, 0.0);
            uE[nC + nComp + i] = bnd(ubR, (size_t)k * nComp + i, INFINITY);
            lbuf[i] = bnd(lbL, (size_t)k * nComp + i, 0.0);
            rbuf[i] = bnd(lbR, (size_t)k * nComp + i, 0.0);
        }
        double* vg = &nvb[(size_t)V_G * np];
        double* vlb = &nvb[(size_t)V_LB * np];
        double* vub = &nvb[(size_t)V_UB * np];
        double* vx0 = &nvb[(size_t)V_X0 * np];
        int nfin = 0;
        for (int i = 0; i < np; i++) { vlb[i] = -INFINITY; vub[i] = INFINITY; }
        for (int i = 0; i < n; i++) {
            vg[i] = g[(size_t)k * n + i];
            vlb[i] = bnd(lb, (size_t)k * n + i, -INFINITY);     // setLB/setUB .ipp:54-112
            vub[i] = bnd(ub, (size_t)k * n + i, INFINITY);
            vx0[i] = x0 ? x0[(size_t)k * n + i] : 0.0;          // setInitialGuess .ipp:133-158
            if (std::isfinite(vlb[i]) || std::isfinite(vub[i])) bidx[nfin++] = i;
        }
        info->nfin = nfin; info->mE = mA + nfin; info->hasY0 = y0 ? 1 : 0;
        HIPCHK(hipMemcpyAsync(d.Q + b * nQ, Qp, sizeof(double) * nQ, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d.E + b * nE, Ep, sizeof(double) * nE, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d.nv + b * nNV, nvb, sizeof(double) * nNV, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d.mv + b * nMV, mvb, sizeof(double) * nMV, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d.boxidx + b * np, bidx, sizeof(int) * np, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d.info + b, info, sizeof(InstInfo), hipMemcpyHostToDevice, h->stream));
        if (nComp) {
            HIPCHK(hipMemcpyAsync(d.lbL + b * nComp, lbuf, sizeof(double) * nComp, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(d.lbR + b * nComp, rbuf, sizeof(double) * nComp, hipMemcpyHostToDevice, h->stream));
        }
        if (y0) {
            memcpy(ybuf, y0 + (size_t)k * d.nd, sizeof(double) * d.nd);
            HIPCHK(hipMemcpyAsync(d.y0 + b * nY, ybuf, sizeof(double) * nY, hipMemcpyHostToDevice, h->stream));
        }
        HIPCHK(hipEventRecord(h->stageDone[slot], h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->setupDone = false;
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_generate_synthetic(lcqp_hip_batch_t* h, uint64_t seed0, uint64_t firstInstance)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    HIPCHK(hipSetDevice(h->device));
    DevBatch& d = h->db;
    if (d.nComp * 2 > d.n) { g_err = "synthetic generator needs 2*nComp <= nV"; return LCQP_INVALID_ARGUMENT; }
    d.hasLbL = d.hasLbR = 0; h->anyLoaded = true;
    dispatch_db(h, ID_k_synth_fill, d.B, nullptr, 0, seed0, firstInstance);
    dispatch_db(h, ID_k_synth_Q, d.B * (d.nblk * (d.nblk + 1) / 2));
    HIPCHK(hipGetLastError());
    h->setupDone = false;
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_read_problem(lcqp_hip_batch_t* h, int b, double* Q, double* g, double* L, double* R,
                                           double* A, double* lbA, double* ubA)
try {
    if (!h || b < 0 || b >= h->db.B) return LCQP_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(h->device));
    DevBatch& d = h->db;
    const int n = d.n, nC = d.nC, nComp = d.nComp, np = d.np, mE = d.mEcap;
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> Qp((size_t)np * np), Ep((size_t)mE * np), nvb((size_t)V_NUM * np), mvb((size_t)M_NUM * mE);
    HIPCHK(hipMemcpy(Qp.data(), d.Q + (size_t)b * np * np, sizeof(double) * np * np, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(Ep.data(), d.E + (size_t)b * mE * np, sizeof(double) * mE * np, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nvb.data(), d.nv + (size_t)b * V_NUM * np, sizeof(double) * V_NUM * np, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mvb.data(), d.mv + (size_t)b * M_NUM * mE, sizeof(double) * M_NUM * mE, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        if (Q) memcpy(Q + (size_t)i * n, &Qp[(size_t)i * np], sizeof(double) * n);
        if (g) g[i] = nvb[(size_t)V_G * np + i];
    }
    for (int r = 0; r < nC; r++) {
        if (A) memcpy(A + (size_t)r * n, &Ep[(size_t)r * np], sizeof(double) * n);
        if (lbA) lbA[r] = mvb[(size_t)M_L * mE + r];
        if (ubA) ubA[r] = mvb[(size_t)M_U * mE + r];
    }
    for (int r = 0; r < nComp; r++) {
        if (L) memcpy(L + (size_t)r * n, &Ep[(size_t)(nC + r) * np], sizeof(double) * n);
        if (R) memcpy(R + (size_t)r * n, &Ep[(size_t)(nC + nComp + r) * np], sizeof(double) * n);
    }
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// the setup kernels of the batch on its stream, the C branch on the side stream
static int launch_setup(lcqp_hip_batch* h)
{
    const DevBatch& d = h->db;
    hipStream_t on = h->stream;
    const int ntile = d.nblk * (d.nblk + 1) / 2;
    const int nrb = (d.mEcap + 63) / 64, nb = (d.mMld + 127) / 128, nmt = nb * (nb + 1);      // k_build_M: 128 x 64 tiles of the lower triangle
    dispatch_db(h, ID_k_prepare, d.B);
    // C and its compressed rows depend on L and R only, the chain L1 -> Et -> M on Q and E: two branches.  The short one goes to the side
    // stream and runs in the gaps of k_factor (one workgroup per instance, a life of dependent chains).  Measured alternatives, round 6
    // (profiles/round6/README.md): the side branch beside k_trsm, or beside k_trsm and k_build_M -- both 0.2 ms slower.
    const bool fork = d.nComp > 0;
    if (fork) {
        HIPCHK(hipEventRecord(h->evFork, on));
        HIPCHK(hipStreamWaitEvent(h->side, h->evFork, 0));
        dispatch_db(h, ID_k_build_C, d.B * ntile, nullptr, 0, 0, 0, h->side);
        dispatch_db(h, ID_k_compress_C, d.B, nullptr, 0, 0, 0, h->side);
        HIPCHK(hipEventRecord(h->evJoin, h->side));
    }
    // more than three workgroups per CU (np <= 256: 36 KB of LDS each): the instantiation held to 128 registers, so that four are resident and
    // the batch needs one round
    dispatch_db(h, (h->nch <= 2 && d.B > 3 * h->numCU) ? ID_k_factor_full : ID_k_factor, d.B);
    dispatch_db(h, h->overlapped ? ID_k_trsm_streamed : ID_k_trsm, d.B * nrb);
    // the join sits in front of the last setup kernel, not behind it: an event recorded right after a stream wait carried a late time stamp
    // (the homotopy kernel appeared 2 ms shorter than rocprofv3 and the wall clock say), and the side branch has long finished by then
    if (fork) HIPCHK(hipStreamWaitEvent(on, h->evJoin, 0));
    dispatch_db(h, ID_k_build_M, d.B * nmt);
    HIPCHK(hipGetLastError());
    h->setupDone = true;
    return 0;
}

extern "C" int lcqp_hip_batch_set_overlapped(lcqp_hip_batch_t* h, int overlapped)
{
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    h->overlapped = overlapped != 0;
    return 0;
}

extern "C" int lcqp_hip_batch_setup(lcqp_hip_batch_t* h)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    HIPCHK(hipSetDevice(h->device));
    return launch_setup(h);
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// One launch for the whole batch.  (Round 5 measured the setup of one slice of the batch beside the homotopy of the slice before, as a
// switch of this call: slower at every split -- profiles/round5/run_chunks_ab.log: 30 200 LCQPs/s in one piece, 26 800 / 27 100 / 23 100 in
// two / three / four slices, the setup kernels crawl beside a homotopy launch that saturates HBM.  The switch is gone; overlap across
// BATCHES is the product's BatchPipeline.)
extern "C" int lcqp_hip_batch_run(lcqp_hip_batch_t* h)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    int rc = launch_setup(h);   // the factorisations are part of runSolver's cost (initializeSolver :885)
    if (rc) return rc;
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    dispatch_db(h, ID_k_lcqp_run, h->db.B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev2, h->stream));
    h->ran = true;
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_synchronize(lcqp_hip_batch_t* h)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_last_timing(lcqp_hip_batch_t* h, float* setup_ms, float* solve_ms)
try {
    if (!h || !h->ran) return LCQP_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev2));
    if (setup_ms) HIPCHK(hipEventElapsedTime(setup_ms, h->ev0, h->ev1));
    if (solve_ms) HIPCHK(hipEventElapsedTime(solve_ms, h->ev1, h->ev2));
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_get_solution(lcqp_hip_batch_t* h, double* x, double* y, lcqp_stats_t* stats)
try {
    if (!h) return LCQP_LCQPOBJECT_NOT_SETUP;
    HIPCHK(hipSetDevice(h->device));
    DevBatch& d = h->db;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (x) HIPCHK(hipMemcpy(x, d.xout, sizeof(double) * (size_t)d.B * d.n, hipMemcpyDeviceToHost));
    if (y) HIPCHK(hipMemcpy(y, d.yout, sizeof(double) * (size_t)d.B * d.nd, hipMemcpyDeviceToHost));
    if (stats) HIPCHK(hipMemcpy(stats, d.stats, sizeof(lcqp_stats_t) * (size_t)d.B, hipMemcpyDeviceToHost));
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// Algorithmic HBM bytes of the last run, from the per-instance work counters:
//   residual evaluation (trial with sweeps, stats.reserved): Q + E once   8*(n*n + m*n)
//   correction                  : L1 fwd+bwd + 2 sweeps over the nT active rows of Et + one fused pass over the inverse factor Ti (nT x ns)
//   working-set update          : the bytes of Ti read and written by row appends and rotations, and the entries of M read (summed
//                                 exactly by the kernel, InstInfo::work[2])
//   ADMM iteration              : LK fwd+bwd + two sweeps over E
//   LCQP                        : one sweep over Q and C (Q x0, C x0); per iterate C pk from compressed rows (12 B per non-zero) or by a
//                                 sweep over C; Q pk comes from the subsolver's verified residual
extern "C" double lcqp_hip_batch_algorithmic_bytes(lcqp_hip_batch_t* h)
try {
    if (!h) return 0.0;
    DevBatch& d = h->db;
    std::vector<lcqp_stats_t> st(d.B);
    if (hipSetDevice(h->device) != hipSuccess) return 0.0;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return 0.0;
    if (hipMemcpy(st.data(), d.stats, sizeof(lcqp_stats_t) * (size_t)d.B, hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
    std::vector<InstInfo> info(d.B);
    if (hipMemcpy(info.data(), d.info, sizeof(InstInfo) * (size_t)d.B, hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
    const double n = d.n, m = d.mA, N = d.n;
    const double bs = 8.0 * N * (N + 2.0);
    double total = 0.0;
    for (int b = 0; b < d.B; b++) {
        // the active rows of each correction and the bytes of each working-set update are summed by the kernel (InstInfo::work), not estimated
        const double rowsEt = info[b].work[0], tiC = info[b].work[1], updBytes = info[b].work[2], nTrsv = info[b].work[5];
        total += st[b].reserved * 8.0 * n * n + 8.0 * n * info[b].work[4];   // true-residual sweeps over Q; rows of E read by both stages of the trials (hot-start trials reuse the last residual)
        total += nTrsv * 0.5 * bs + 8.0 * rowsEt * n + 8.0 * (tiC + rowsEt);   // corrections: triangular solves with L1 (two per full, one per predicted correction), rows of Et, pass over Ti
        total += updBytes;
        total += st[b].admmIter * (bs + 2.0 * 8.0 * m * n);
        total += 2.0 * 8.0 * n * n;                                                   // Q x0, C x0: the one sweep over Q and C
        total += (st[b].iterTotal + 1) * (info[b].cNnz >= 0 ? 12.0 * info[b].cNnz : 8.0 * n * n);   // C pk per LCQP iterate: compressed rows or a sweep
    }
    return total;
}
catch (...) { return 0.0; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_batch_work_sums(lcqp_hip_batch_t* h, double out[6])
try {
    if (!h || !out) return LCQP_INVALID_ARGUMENT;
    DevBatch& d = h->db;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<InstInfo> info(d.B);
    HIPCHK(hipMemcpy(info.data(), d.info, sizeof(InstInfo) * (size_t)d.B, hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; k++) out[k] = 0.0;
    for (int b = 0; b < d.B; b++) for (int k = 0; k < 6; k++) out[k] += info[b].work[k];
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// =================================================================================================
// QP object (SubsolverBase semantics): a batch of one with nComp = 0 whose rows are the nC stacked rows
// =================================================================================================
struct lcqp_hip_qp {
    lcqp_hip_batch* hb;
    int nV, nC;
    std::vector<double> Q, A;          // host copies (deep copy, src/SubsolverQPOASES.cpp:41-45)
    std::vector<double> lbA, ubA, lb, ub;
    bool haveBounds, withBox;
    lcqp_options_t opt;
    int device;
    std::vector<double> xsol, ysol;
    int cAdmm, cTrials, cFact, cCorr;
};

extern "C" lcqp_hip_qp_t* lcqp_hip_qp_create(int nV, int nC, const double* Q, const double* A, const lcqp_options_t* opt, int device)
try {
    if (nV <= 0 || nC < 0 || !Q || (nC > 0 && !A)) { g_err = "invalid arguments"; return nullptr; }
    lcqp_hip_qp* q = new lcqp_hip_qp();
    q->hb = nullptr; q->nV = nV; q->nC = nC; q->device = device; q->haveBounds = false; q->withBox = false;
    q->Q.assign(Q, Q + (size_t)nV * nV);
    if (nC) q->A.assign(A, A + (size_t)nC * nV);
    if (opt) q->opt = *opt; else lcqp_hip_options_default(&q->opt);
    q->xsol.assign(nV, 0.0); q->ysol.assign((size_t)nV + nC, 0.0);
    q->cAdmm = q->cTrials = q->cFact = q->cCorr = 0;
    return q;
}
catch (...) { g_err = "out of host memory"; return nullptr; }   // nothing throws across the C boundary

extern "C" lcqp_hip_qp_t* lcqp_hip_qp_clone(const lcqp_hip_qp_t* s)
try {
    if (!s) return nullptr;
    // The reference copies subsolvers only before their first use (src/Subsolver.cpp:125-136,
    // src/LCQProblem.cpp:906-907): the clone carries the problem data and options; device state is
    // rebuilt by its own first solve.
    lcqp_hip_qp* q = new lcqp_hip_qp(*s);
    q->hb = nullptr;
    q->haveBounds = false;
    return q;
}
catch (...) { g_err = "out of host memory"; return nullptr; }   // nothing throws across the C boundary

extern "C" void lcqp_hip_qp_destroy(lcqp_hip_qp_t* q)
try {
    if (!q) return;
    if (q->hb) lcqp_hip_batch_destroy(q->hb);
    delete q;
}
catch (...) { }   // nothing throws across the C boundary

extern "C" int lcqp_hip_qp_set_options(lcqp_hip_qp_t* q, const lcqp_options_t* opt)
try {
    if (!q || !opt) return LCQP_INVALID_ARGUMENT;
    q->opt = *opt;
    q->haveBounds = false;   // forces a fresh setup (rho / sigma / prox weights enter the factorisations)
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

static bool same_pattern(const std::vector<double>& a0, const std::vector<double>& b0, const double* a1, const double* b1, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        const double lo1 = a1 ? a1[i] : -INFINITY, hi1 = b1 ? b1[i] : INFINITY;
        const bool fin0 = std::isfinite(a0[i]) || std::isfinite(b0[i]), fin1 = std::isfinite(lo1) || std::isfinite(hi1);
        const bool eq0 = a0[i] == b0[i], eq1 = lo1 == hi1;
        if (fin0 != fin1 || eq0 != eq1) return false;
    }
    return true;
}

extern "C" int lcqp_hip_qp_solve(lcqp_hip_qp_t* q, int initialSolve, int* iterations, int* exit_flag,
                                 const double* g, const double* lbA, const double* ubA,
                                 const double* x0, const double* y0, const double* lb, const double* ub)
try {
    if (!q || !g || !iterations || !exit_flag) return LCQP_INVALID_ARGUMENT;
    const int n = q->nV, nC = q->nC;
    *iterations = 0; *exit_flag = 0;
    const bool needBox = (lb != nullptr) || (ub != nullptr);
    bool fresh = initialSolve || !q->hb || !q->haveBounds;
    if (!fresh) {
        if (needBox && !q->withBox) fresh = true;
        else if (!same_pattern(q->lbA, q->ubA, lbA, ubA, nC) || !same_pattern(q->lb, q->ub, lb, ub, n)) fresh = true;
    }
    if (q->hb && (fresh && (needBox && !q->withBox))) { lcqp_hip_batch_destroy(q->hb); q->hb = nullptr; }
    if (!q->hb) {
        q->hb = lcqp_hip_batch_create(1, n, nC, 0, needBox ? 1 : 0, q->device);
        if (!q->hb) { *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
        q->withBox = needBox;
        fresh = true;
    }
    lcqp_hip_batch* h = q->hb;
    DevBatch& d = h->db;
    if (hipSetDevice(h->device) != hipSuccess) { *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
    q->lbA.assign(nC, -INFINITY); q->ubA.assign(nC, INFINITY); q->lb.assign(n, -INFINITY); q->ub.assign(n, INFINITY);
    for (int i = 0; i < nC; i++) { if (lbA) q->lbA[i] = lbA[i]; if (ubA) q->ubA[i] = ubA[i]; }
    for (int i = 0; i < n; i++) { if (lb) q->lb[i] = lb[i]; if (ub) q->ub[i] = ub[i]; }
    q->haveBounds = true;
    int rc;
    if (fresh) {
        lcqp_hip_batch_set_options(h, &q->opt);
        // batch of one, nComp = 0: the "A" block carries all stacked rows; L/R are empty
        double dummy = 0.0;
        rc = lcqp_hip_batch_load(h, 0, 1, q->Q.data(), g, &dummy, &dummy, nullptr, nullptr, nullptr, nullptr,
                                 nC ? q->A.data() : nullptr, q->lbA.data(), q->ubA.data(),
                                 q->withBox ? q->lb.data() : nullptr, q->withBox ? q->ub.data() : nullptr, x0, y0);
        if (rc) { *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
        rc = launch_setup(h);
        if (rc) { *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
        initialSolve = 1;
    } else {
        // same pattern: refresh bound values (finite/equality pattern unchanged, factorisations stay valid)
        std::vector<double> l(d.mEcap, 0.0), u(d.mEcap, 0.0);
        for (int r = 0; r < nC; r++) { l[r] = q->lbA[r]; u[r] = q->ubA[r]; }
        int k = 0;
        for (int i = 0; i < n; i++)
            if (std::isfinite(q->lb[i]) || std::isfinite(q->ub[i])) { l[nC + k] = q->lb[i]; u[nC + k] = q->ub[i]; k++; }
        if (hipMemcpyAsync(d.mv + (size_t)M_L * d.mEcap, l.data(), sizeof(double) * d.mEcap, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(d.mv + (size_t)M_U * d.mEcap, u.data(), sizeof(double) * d.mEcap, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            // new bound values: the safe margins of the row screening (M_MG, relative to the old bounds) are void -- NaN margins make
            // the next residual sweep read every row
            hipMemsetAsync(d.mv + (size_t)M_MG * d.mEcap, 0xFF, sizeof(double) * d.mEcap, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
    }
    // linear term of this call
    std::vector<double> gp(d.np, 0.0);
    memcpy(gp.data(), g, sizeof(double) * n);
    if (hipMemcpyAsync(d.nv + (size_t)V_GK * d.np, gp.data(), sizeof(double) * d.np, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
        *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR;
    }
    dispatch_db(h, ID_k_qp_solve, 1, nullptr, initialSolve ? 1 : 0);
    lcqp_stats_t st;
    if (hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(&st, d.stats, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) { *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
    *iterations = st.subproblemIter;
    *exit_flag = st.qpSolverExitFlag;
    q->cAdmm += st.admmIter; q->cTrials += st.trials; q->cFact += st.factorizations; q->cCorr += st.corrections;
    if (st.qpSolverExitFlag != 0) return LCQP_SUBPROBLEM_SOLVER_ERROR;
    if (hipMemcpy(q->xsol.data(), d.xout, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(q->ysol.data(), d.yout, sizeof(double) * ((size_t)n + nC), hipMemcpyDeviceToHost) != hipSuccess) {
        *exit_flag = -1; return LCQP_SUBPROBLEM_SOLVER_ERROR;
    }
    return LCQP_SUCCESSFUL_RETURN;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" void lcqp_hip_qp_get_solution(lcqp_hip_qp_t* q, double* x, double* y)
try {
    if (!q) return;
    if (x) memcpy(x, q->xsol.data(), sizeof(double) * q->nV);
    if (y) memcpy(y, q->ysol.data(), sizeof(double) * ((size_t)q->nV + q->nC));
}
catch (...) { }   // nothing throws across the C boundary

extern "C" void lcqp_hip_qp_get_counters(lcqp_hip_qp_t* q, int* admm, int* trials, int* factorizations, int* corrections)
try {
    if (!q) return;
    if (admm) *admm = q->cAdmm;
    if (trials) *trials = q->cTrials;
    if (factorizations) *factorizations = q->cFact;
    if (corrections) *corrections = q->cCorr;
}
catch (...) { }   // nothing throws across the C boundary

// =================================================================================================
// building blocks (tests, micro-benchmarks)
// =================================================================================================
struct TmpBuf {
    std::vector<void*> p;
    ~TmpBuf() { for (void* q : p) (void)hipFree(q); }
    double* get(size_t count, bool zero = true)
    {
        void* q = nullptr;
        if (hipMalloc(&q, (count ? count : 1) * sizeof(double)) != hipSuccess) return nullptr;
        if (zero) (void)hipMemset(q, 0, (count ? count : 1) * sizeof(double));
        p.push_back(q);
        return (double*)q;
    }
};

static int upload_padded(double* dst, const double* src, int batch, int rows, int cols, int ld, int rowsPad)
{
    // src: [batch][rows][cols] -> dst: [batch][rowsPad][ld]
    std::vector<double> buf((size_t)rowsPad * ld);
    for (int b = 0; b < batch; b++) {
        std::fill(buf.begin(), buf.end(), 0.0);
        for (int r = 0; r < rows; r++) memcpy(&buf[(size_t)r * ld], src + ((size_t)b * rows + r) * cols, sizeof(double) * cols);
        HIPCHK(hipMemcpy(dst + (size_t)b * rowsPad * ld, buf.data(), sizeof(double) * rowsPad * ld, hipMemcpyHostToDevice));
    }
    return 0;
}
static int download_padded(double* dst, const double* src, int batch, int rows, int cols, int ld, int rowsPad)
{
    std::vector<double> buf((size_t)rowsPad * ld);
    for (int b = 0; b < batch; b++) {
        HIPCHK(hipMemcpy(buf.data(), src + (size_t)b * rowsPad * ld, sizeof(double) * rowsPad * ld, hipMemcpyDeviceToHost));
        for (int r = 0; r < rows; r++) memcpy(dst + ((size_t)b * rows + r) * cols, &buf[(size_t)r * ld], sizeof(double) * cols);
    }
    return 0;
}

extern "C" int lcqp_hip_util_symv(int batch, int n, double alpha, const double* A, const double* bv, const double* cv, double* dv)
try {
    if (n <= 0 || n > 4096 || batch <= 0) return LCQP_HIP_UNSUPPORTED;
    const int nch = padded_nch(n), np = 128 * nch;
    TmpBuf tb;
    double *dA = tb.get((size_t)batch * np * np), *db_ = tb.get((size_t)batch * np), *dc = tb.get((size_t)batch * np), *dd = tb.get((size_t)batch * np);
    if (!dA || !db_ || !dc || !dd) return set_err("hipMalloc", hipErrorOutOfMemory);
    int rc = upload_padded(dA, A, batch, n, n, np, np); if (rc) return rc;
    rc = upload_padded(db_, bv, batch, 1, n, np, 1); if (rc) return rc;
    rc = upload_padded(dc, cv, batch, 1, n, np, 1); if (rc) return rc;
    { LaunchArgs la; la.n = n; la.alpha = alpha; la.A = dA; la.b = db_; la.c = dc; la.d = dd; lcqp_dispatch(nch, ID_k_util_symv, batch, 0, la); }
    HIPCHK(hipDeviceSynchronize());
    return download_padded(dv, dd, batch, 1, n, np, 1);
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

static int util_rows(int batch, int m, int n, const double* A, const double* x, double* dots, const double* coef, double* outT)
{
    if (n <= 0 || n > 4096 || batch <= 0 || m <= 0) return LCQP_HIP_UNSUPPORTED;
    const int nch = padded_nch(n), np = 128 * nch;
    TmpBuf tb;
    double* dA = tb.get((size_t)batch * m * np);
    double* dx = x ? tb.get((size_t)batch * np) : nullptr;
    double* dd = dots ? tb.get((size_t)batch * m) : nullptr;
    double* dcf = coef ? tb.get((size_t)batch * m) : nullptr;
    double* dout = outT ? tb.get((size_t)batch * np) : nullptr;
    if (!dA) return set_err("hipMalloc", hipErrorOutOfMemory);
    int rc = upload_padded(dA, A, batch, m, n, np, m); if (rc) return rc;
    if (x) { rc = upload_padded(dx, x, batch, 1, n, np, 1); if (rc) return rc; }
    if (coef) HIPCHK(hipMemcpy(dcf, coef, sizeof(double) * (size_t)batch * m, hipMemcpyHostToDevice));
    { LaunchArgs la; la.m = m; la.A = dA; la.x = dx; la.dots = dd; la.coef = dcf; la.outT = dout; lcqp_dispatch(nch, ID_k_util_rows, batch, 0, la); }
    HIPCHK(hipDeviceSynchronize());
    if (dots) HIPCHK(hipMemcpy(dots, dd, sizeof(double) * (size_t)batch * m, hipMemcpyDeviceToHost));
    if (outT) return download_padded(outT, dout, batch, 1, n, np, 1);
    return 0;
}

extern "C" int lcqp_hip_util_rows_list(int batch, int m, int n, const double* A, const int* list, int nlist, const double* x, const double* coef,
                                       double* dots, double* outT)
try {
    if (n <= 0 || n > 4096 || batch <= 0 || m <= 0 || nlist < 0 || nlist > m || !list) return LCQP_HIP_UNSUPPORTED;
    const int nch = padded_nch(n), np = 128 * nch;
    TmpBuf tb;
    double* dA = tb.get((size_t)batch * m * np);
    double* dx = x ? tb.get((size_t)batch * np) : nullptr;
    double* dd = dots ? tb.get((size_t)batch * m) : nullptr;
    double* dcf = coef ? tb.get((size_t)batch * m) : nullptr;
    double* dout = outT ? tb.get((size_t)batch * np) : nullptr;
    int* dl = reinterpret_cast<int*>(tb.get(((size_t)batch * std::max(nlist, 1) + 1) / 2 + 1));
    if (!dA || !dl) return set_err("hipMalloc", hipErrorOutOfMemory);
    int rc = upload_padded(dA, A, batch, m, n, np, m); if (rc) return rc;
    if (x) { rc = upload_padded(dx, x, batch, 1, n, np, 1); if (rc) return rc; }
    if (coef) HIPCHK(hipMemcpy(dcf, coef, sizeof(double) * (size_t)batch * m, hipMemcpyHostToDevice));
    if (dots) HIPCHK(hipMemcpy(dd, dots, sizeof(double) * (size_t)batch * m, hipMemcpyHostToDevice));      // rows outside the list keep the caller's values
    if (nlist > 0) HIPCHK(hipMemcpy(dl, list, sizeof(int) * (size_t)batch * nlist, hipMemcpyHostToDevice));
    { LaunchArgs la; la.m = m; la.n = nlist; la.A = dA; la.list = dl; la.x = dx; la.dots = dd; la.coef = dcf; la.outT = dout; lcqp_dispatch(nch, ID_k_util_rows_list, batch, 0, la); }
    HIPCHK(hipDeviceSynchronize());
    if (dots) HIPCHK(hipMemcpy(dots, dd, sizeof(double) * (size_t)batch * m, hipMemcpyDeviceToHost));
    if (outT) return download_padded(outT, dout, batch, 1, n, np, 1);
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_util_gemv(int batch, int m, int n, const double* A, const double* b, double* c)
try {
    return util_rows(batch, m, n, A, b, c, nullptr, nullptr);
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary
extern "C" int lcqp_hip_util_gemv_t(int batch, int m, int n, const double* A, const double* b, double* c)
try {
    return util_rows(batch, m, n, A, nullptr, nullptr, b, c);
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_util_symm_product(int batch, int m, int n, const double* A, const double* Bm, double* C)
try {
    // goes through the batch object so that the production kernel k_build_C is what is tested
    lcqp_hip_batch* h = lcqp_hip_batch_create(batch, n, 0, m, 0, 0);
    if (!h) return LCQP_HIP_ERROR;
    DevBatch& d = h->db;
    int rc = upload_padded(d.E, A, batch, m, n, d.np, d.mEcap);
    if (!rc) {
        // second block (R) starts at row m of each instance
        std::vector<double> buf((size_t)d.mEcap * d.np);
        for (int b = 0; b < batch && !rc; b++) {
            if (hipMemcpy(buf.data(), d.E + (size_t)b * d.mEcap * d.np, sizeof(double) * buf.size(), hipMemcpyDeviceToHost) != hipSuccess) { rc = LCQP_HIP_ERROR; break; }
            for (int r = 0; r < m; r++) memcpy(&buf[(size_t)(m + r) * d.np], Bm + ((size_t)b * m + r) * n, sizeof(double) * n);
            if (hipMemcpy(d.E + (size_t)b * d.mEcap * d.np, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice) != hipSuccess) rc = LCQP_HIP_ERROR;
        }
    }
    if (!rc) {
        dispatch_db(h, ID_k_build_C, d.B * (d.nblk * (d.nblk + 1) / 2));
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = LCQP_HIP_ERROR;
    }
    if (!rc) rc = download_padded(C, d.C, batch, n, n, d.np, d.np);
    lcqp_hip_batch_destroy(h);
    return rc;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// =================================================================================================
// CSC utilities on the device: compressed-segment gather products.
// A CSC matrix is uploaded together with its transpose (the CSC of A' is the CSR of A), so both
// MatrixMultiplication (A b) and TransponsedMatrixMultiplication (A'b) are gathers over compressed segments --
// no atomics, deterministic, the same summation order as the reference's inner loops
// (src/Utilities.cpp:49-59,75-82,189-199,228-241).
// =================================================================================================
// out[s] = alpha * sum_{k in [ptr[s], ptr[s+1])} val[k] * v[idx[k]] + (add ? add[s] : 0); 16 lanes per segment
__global__ __launch_bounds__(256) void k_seg_gather(int nseg, const int* __restrict__ ptr, const int* __restrict__ idx,
                                                    const double* __restrict__ val, const double* __restrict__ v, double alpha,
                                                    const double* __restrict__ add, double* __restrict__ out)
{
    const int sub = threadIdx.x & 15;
    const int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    double s = 0.0;
    if (seg < nseg) {
        const int k0 = ptr[seg], k1 = ptr[seg + 1];
        for (int k = k0 + sub; k < k1; k += 16) s += val[k] * v[idx[k]];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
    if (seg < nseg && sub == 0) out[seg] = alpha * s + (add ? add[seg] : 0.0);
}

struct lcqp_hip_csc {
    int m, n, nnz, device;
    int *p, *i, *tp, *ti;        // CSC of A and CSC of A' (device)
    double *x, *tx;
    double *vin, *vout, *vadd;   // staging vectors of length max(m, n)
};

extern "C" lcqp_hip_csc_t* lcqp_hip_csc_create(int m, int n, int nnz, const int* p, const int* i, const double* x, int device)
try {
    if (m <= 0 || n <= 0 || nnz < 0 || !p || (nnz && (!i || !x))) { g_err = "invalid CSC arguments"; return nullptr; }
    HIPCHKN(hipSetDevice(device));
    // transpose on the host: counting sort by row index (stable, so columns stay ascending inside a row)
    std::vector<int> tp(m + 1, 0), ti(nnz ? nnz : 1);
    std::vector<double> tx(nnz ? nnz : 1);
    for (int k = 0; k < nnz; k++) { if (i[k] < 0 || i[k] >= m) { g_err = "CSC row index out of bounds"; return nullptr; } tp[i[k] + 1]++; }
    for (int r = 0; r < m; r++) tp[r + 1] += tp[r];
    std::vector<int> cur(tp.begin(), tp.end() - 1);
    for (int c = 0; c < n; c++)
        for (int k = p[c]; k < p[c + 1]; k++) { const int d = cur[i[k]]++; ti[d] = c; tx[d] = x[k]; }
    lcqp_hip_csc* h = new lcqp_hip_csc();
    memset(h, 0, sizeof(*h));
    h->m = m; h->n = n; h->nnz = nnz; h->device = device;
    const size_t nz = nnz ? nnz : 1, mx = (size_t)(m > n ? m : n);
    if (hipMalloc((void**)&h->p, sizeof(int) * (n + 1)) != hipSuccess || hipMalloc((void**)&h->i, sizeof(int) * nz) != hipSuccess ||
        hipMalloc((void**)&h->x, sizeof(double) * nz) != hipSuccess || hipMalloc((void**)&h->tp, sizeof(int) * (m + 1)) != hipSuccess ||
        hipMalloc((void**)&h->ti, sizeof(int) * nz) != hipSuccess || hipMalloc((void**)&h->tx, sizeof(double) * nz) != hipSuccess ||
        hipMalloc((void**)&h->vin, sizeof(double) * mx) != hipSuccess || hipMalloc((void**)&h->vout, sizeof(double) * mx) != hipSuccess ||
        hipMalloc((void**)&h->vadd, sizeof(double) * mx) != hipSuccess) { g_err = "hipMalloc failed"; lcqp_hip_csc_destroy(h); return nullptr; }
    HIPCHKN(hipMemcpy(h->p, p, sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    HIPCHKN(hipMemcpy(h->tp, tp.data(), sizeof(int) * (m + 1), hipMemcpyHostToDevice));
    if (nnz) {
        HIPCHKN(hipMemcpy(h->i, i, sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIPCHKN(hipMemcpy(h->x, x, sizeof(double) * nnz, hipMemcpyHostToDevice));
        HIPCHKN(hipMemcpy(h->ti, ti.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIPCHKN(hipMemcpy(h->tx, tx.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    }
    return h;
}
catch (...) { g_err = "out of host memory"; return nullptr; }   // nothing throws across the C boundary

extern "C" void lcqp_hip_csc_destroy(lcqp_hip_csc_t* h)
try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->p); (void)hipFree(h->i); (void)hipFree(h->x); (void)hipFree(h->tp); (void)hipFree(h->ti); (void)hipFree(h->tx);
    (void)hipFree(h->vin); (void)hipFree(h->vout); (void)hipFree(h->vadd);
    delete h;
}
catch (...) { }   // nothing throws across the C boundary

// d = alpha * op(A) * b + (c ? c : 0);  transposed != 0: op(A) = A' (b has m entries, d has n), else op(A) = A.
// repeat > 1 re-launches the product for timing; *ms = time per launch.
extern "C" int lcqp_hip_csc_apply(lcqp_hip_csc_t* h, int transposed, double alpha, const double* b, const double* c, double* d,
                                  int repeat, float* ms)
try {
    if (!h || !b || !d) return LCQP_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(h->device));
    const int nin = transposed ? h->m : h->n, nout = transposed ? h->n : h->m;
    HIPCHK(hipMemcpy(h->vin, b, sizeof(double) * nin, hipMemcpyHostToDevice));
    if (c) HIPCHK(hipMemcpy(h->vadd, c, sizeof(double) * nout, hipMemcpyHostToDevice));
    const int* ptr = transposed ? h->p : h->tp;     // A'b gathers over the columns of A, A b over the columns of A'
    const int* idx = transposed ? h->i : h->ti;
    const double* val = transposed ? h->x : h->tx;
    const int grid = (nout * 16 + 255) / 256;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    if (repeat < 1) repeat = 1;
    hipLaunchKernelGGL(k_seg_gather, dim3(grid), dim3(256), 0, 0, nout, ptr, idx, val, h->vin, alpha, c ? h->vadd : nullptr, h->vout);
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < repeat; r++)
        hipLaunchKernelGGL(k_seg_gather, dim3(grid), dim3(256), 0, 0, nout, ptr, idx, val, h->vin, alpha, c ? h->vadd : nullptr, h->vout);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / repeat;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(hipMemcpy(d, h->vout, sizeof(double) * nout, hipMemcpyDeviceToHost));
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

// micro-benchmark of the row sweep (wg_rows) on device-resident random data: mode 1 = dots only (A x),
// 2 = axpy only (A'y), 3 = both in one sweep; *ms = time per launch
__global__ void k_fill_random(double* p, size_t n, uint64_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = 2.0 * lcqp_u01(seed, i) - 1.0;
}

extern "C" int lcqp_hip_bench_rows(int batch, int m, int n, int mode, int repeat, float* ms)
try {
    if (n <= 0 || n > 4096 || batch <= 0 || m <= 0) return LCQP_HIP_UNSUPPORTED;
    const int nch = padded_nch(n), np = 128 * nch;
    TmpBuf tb;
    double *dA = tb.get((size_t)batch * m * np, false), *dx = tb.get((size_t)batch * np, false), *dd = tb.get((size_t)batch * m, false);
    double *dcf = tb.get((size_t)batch * m, false), *dout = tb.get((size_t)batch * np, false);
    if (!dA || !dx || !dd || !dcf || !dout) return set_err("hipMalloc", hipErrorOutOfMemory);
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, 0, dA, (size_t)batch * m * np, 1ULL);
    hipLaunchKernelGGL(k_fill_random, dim3(256), dim3(256), 0, 0, dx, (size_t)batch * np, 2ULL);
    hipLaunchKernelGGL(k_fill_random, dim3(256), dim3(256), 0, 0, dcf, (size_t)batch * m, 3ULL);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const double* px = (mode & 1) ? dx : nullptr;
    double* pd = (mode & 1) ? dd : nullptr;
    const double* pc = (mode & 2) ? dcf : nullptr;
    double* po = (mode & 2) ? dout : nullptr;
    for (int r = 0; r <= repeat; r++) {
        if (r == 1) HIPCHK(hipEventRecord(e0, 0));
        { LaunchArgs la; la.m = m; la.A = dA; la.x = px; la.dots = pd; la.coef = pc; la.outT = po; lcqp_dispatch(nch, ID_k_util_rows, batch, 0, la); }
    }
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / (repeat > 0 ? repeat : 1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary

extern "C" int lcqp_hip_chol_solve(int batch, int n, const double* K, const double* b, double* x, int repeat, float* ms)
try {
    if (n <= 0 || n > LCQP_MAX_ACTIVE || batch <= 0) return LCQP_HIP_UNSUPPORTED;   // k_chol / k_backsolve use the 35 KiB arena
    const int np = ((n + 63) / 64) * 64, nblk = np / 64;
    TmpBuf tb;
    double *dF = tb.get((size_t)batch * np * np), *dscr = tb.get((size_t)batch * 4096), *drhs = tb.get((size_t)batch * np), *dx = tb.get((size_t)batch * np);
    int* dfail = nullptr;
    HIPCHK(hipMalloc((void**)&dfail, sizeof(int) * batch));
    tb.p.push_back(dfail);
    HIPCHK(hipMemset(dfail, 0, sizeof(int) * batch));
    if (!dF || !dscr || !drhs || !dx) return set_err("hipMalloc", hipErrorOutOfMemory);
    // pad with a unit diagonal
    {
        std::vector<double> buf((size_t)np * np);
        for (int bb = 0; bb < batch; bb++) {
            std::fill(buf.begin(), buf.end(), 0.0);
            for (int i = 0; i < n; i++) memcpy(&buf[(size_t)i * np], K + ((size_t)bb * n + i) * n, sizeof(double) * n);
            for (int i = n; i < np; i++) buf[(size_t)i * np + i] = 1.0;
            HIPCHK(hipMemcpy(dF + (size_t)bb * np * np, buf.data(), sizeof(double) * np * np, hipMemcpyHostToDevice));
        }
    }
    int rc = upload_padded(drhs, b, batch, 1, n, np, 1); if (rc) return rc;
    hipLaunchKernelGGL(k_chol, dim3(batch), dim3(WG), 0, 0, np, nblk, n, dF, dscr, dfail);
    HIPCHK(hipDeviceSynchronize());
    std::vector<int> fail(batch);
    HIPCHK(hipMemcpy(fail.data(), dfail, sizeof(int) * batch, hipMemcpyDeviceToHost));
    for (int i = 0; i < batch; i++) if (fail[i]) { g_err = "matrix not positive definite"; return LCQP_SUBPROBLEM_SOLVER_ERROR; }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    if (repeat < 1) repeat = 1;
    hipLaunchKernelGGL(k_backsolve, dim3(batch), dim3(WG), 0, 0, np, nblk, dF, drhs, dx);   // warm-up
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < repeat; r++) hipLaunchKernelGGL(k_backsolve, dim3(batch), dim3(WG), 0, 0, np, nblk, dF, drhs, dx);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / repeat;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return download_padded(x, dx, batch, 1, n, np, 1);
}
catch (...) { g_err = "out of host memory"; return LCQP_HIP_ERROR; }   // nothing throws across the C boundary
