/*
 * lcqp_synth.h -- counter-based synthetic dense LCQP instance generator.
 *
 * Shared by the HIP product (device-side generation, lcqpow_amd/csrc) and by the CPU oracle
 * (oracle/), so both sides see bit-identical inputs: M, g, A, x*, slacks are single expressions, and the two
 * reductions (Q = M'M/n + I, A x* for the bounds) are summed in the same order with separately rounded
 * products on both sides (k_synth_Q / k_synth_fill, orc_synth_generate; tests compare them bit for bit).
 *
 * Structure follows the reference's own examples: one-hot complementarity selectors on disjoint
 * variables (examples/warm_up.cpp:34-35, examples/OptimizeOnCircle.cpp:86-87), zero lower
 * complementarity bounds (src/LCQProblem.cpp:753,773), no box bounds, x0 = 0, y0 = NULL.
 *
 * Stream: SplitMix64 with state0 = seed0 ^ instance_id; the k-th output is
 *   mix(state0 + (k+1)*0x9E3779B97F4A7C15), so any element can be generated independently.
 * Field offsets inside one instance stream (n = nV):
 *   M      [0, n*n)                 U(-1,1)
 *   g      [n*n, n*n+n)             U(-1,1)
 *   coin   next nComp               U(0,1)   (<0.5: L side is the zero side)
 *   xs     next n                   U(-1,1)  (pairs remapped to U(0,1))
 *   A      next nC*n                U(-1,1)/sqrt(n)
 *   slo    next nC                  U(0.1,1)
 *   shi    next nC                  U(0.1,1)
 */
#ifndef LCQP_SYNTH_H
#define LCQP_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LCQP_SYNTH_FN static __host__ __device__ __forceinline__
#else
#define LCQP_SYNTH_FN static inline
#endif

#define LCQP_SYNTH_SEED0 0x4C43515000000001ULL

LCQP_SYNTH_FN uint64_t lcqp_sm64(uint64_t state0, uint64_t k)
{
    uint64_t z = state0 + (k + 1ULL) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* uniform double in [0,1) from the top 53 bits */
LCQP_SYNTH_FN double lcqp_u01(uint64_t state0, uint64_t k)
{
    return (double)(lcqp_sm64(state0, k) >> 11) * (1.0 / 9007199254740992.0);
}

LCQP_SYNTH_FN uint64_t lcqp_synth_state(uint64_t seed0, uint64_t instance) { return seed0 ^ instance; }

/* stream offsets */
LCQP_SYNTH_FN uint64_t lcqp_synth_off_M(int n, int nC, int nComp) { (void)n; (void)nC; (void)nComp; return 0; }
LCQP_SYNTH_FN uint64_t lcqp_synth_off_g(int n, int nC, int nComp) { (void)nC; (void)nComp; return (uint64_t)n * n; }
LCQP_SYNTH_FN uint64_t lcqp_synth_off_coin(int n, int nC, int nComp) { return lcqp_synth_off_g(n, nC, nComp) + (uint64_t)n; }
LCQP_SYNTH_FN uint64_t lcqp_synth_off_xs(int n, int nC, int nComp) { return lcqp_synth_off_coin(n, nC, nComp) + (uint64_t)nComp; }
LCQP_SYNTH_FN uint64_t lcqp_synth_off_A(int n, int nC, int nComp) { return lcqp_synth_off_xs(n, nC, nComp) + (uint64_t)n; }
LCQP_SYNTH_FN uint64_t lcqp_synth_off_slo(int n, int nC, int nComp) { return lcqp_synth_off_A(n, nC, nComp) + (uint64_t)nC * n; }
LCQP_SYNTH_FN uint64_t lcqp_synth_off_shi(int n, int nC, int nComp) { return lcqp_synth_off_slo(n, nC, nComp) + (uint64_t)nC; }

LCQP_SYNTH_FN double lcqp_synth_M(uint64_t st, int n, int i, int j) { return 2.0 * lcqp_u01(st, (uint64_t)i * n + j) - 1.0; }
LCQP_SYNTH_FN double lcqp_synth_g(uint64_t st, int n, int nC, int nComp, int i)
{
    return 2.0 * lcqp_u01(st, lcqp_synth_off_g(n, nC, nComp) + i) - 1.0;
}
/* feasible point x*: variables [0,nComp) are the L side, [nComp,2nComp) the R side of pair i */
LCQP_SYNTH_FN double lcqp_synth_xstar(uint64_t st, int n, int nC, int nComp, int i)
{
    double v = 2.0 * lcqp_u01(st, lcqp_synth_off_xs(n, nC, nComp) + i) - 1.0;
    if (i < 2 * nComp) {
        int pair = (i < nComp) ? i : i - nComp;
        double coin = lcqp_u01(st, lcqp_synth_off_coin(n, nC, nComp) + pair);
        int lzero = coin < 0.5;
        double pos = 0.5 * (v + 1.0);
        if (i < nComp) return lzero ? 0.0 : pos;
        return lzero ? pos : 0.0;
    }
    return v;
}
/* A[r][c] without the 1/sqrt(n) factor (callers multiply by rsn = 1/sqrt(n)) */
LCQP_SYNTH_FN double lcqp_synth_Araw(uint64_t st, int n, int nC, int nComp, int r, int c)
{
    return 2.0 * lcqp_u01(st, lcqp_synth_off_A(n, nC, nComp) + (uint64_t)r * n + c) - 1.0;
}
/* 0.1 + 0.9 u: product and sum rounded separately on both sides (hipcc would otherwise fuse them on the device) */
LCQP_SYNTH_FN double lcqp_synth_slack(double u)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double p = 0.9 * u;
    return 0.1 + p;
}
LCQP_SYNTH_FN double lcqp_synth_slo(uint64_t st, int n, int nC, int nComp, int r)
{
    return lcqp_synth_slack(lcqp_u01(st, lcqp_synth_off_slo(n, nC, nComp) + r));
}
LCQP_SYNTH_FN double lcqp_synth_shi(uint64_t st, int n, int nC, int nComp, int r)
{
    return lcqp_synth_slack(lcqp_u01(st, lcqp_synth_off_shi(n, nC, nComp) + r));
}

#endif /* LCQP_SYNTH_H */
