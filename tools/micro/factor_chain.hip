// Micro-benchmark: what does one step of the band factorisation's chain cost?  One wavefront, 8 lane groups of 8, N steps.
// Variants: 0 full step (pivot row through LDS, IEEE division, 8 FMAs); 1 no division (v_rcp_f64 + one Newton step); 2 no LDS exchange;
// 3 division only; 4 LDS exchange only.   build: hipcc -O3 --offload-arch=gfx950 -o factor_chain factor_chain.hip ; run: ./factor_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
constexpr int G = 8;
#ifndef STRIDE
#define STRIDE 16
#endif
__shared__ double buf_s[8 * STRIDE];
template <int VAR>
__global__ void k_chain(double* out, int N, unsigned long long* ticks)
{
    const int l = threadIdx.x & (G - 1);
    double* buf = buf_s + (threadIdx.x / G) * STRIDE;
    double wr[G];
    for (int k = 0; k < G; k++) wr[k] = (k == 0) ? 4.0 + 0.01 * l : 0.1 / (1 + k);
    buf[G + l] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j0 = 0; j0 < N; j0 += G) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int ag = (l - u) & (G - 1);
            double p[G], d;
            if (VAR == 2 || VAR == 3) {
#pragma unroll
                for (int k = 0; k < G; k++) p[k] = wr[k] * 0.5;
                d = wr[0] + 1.0;
            } else {
                if (ag == 0) {
#pragma unroll
                    for (int k = 0; k < G; k += 2) { dv2 v; v.x = wr[k]; v.y = wr[k + 1]; *reinterpret_cast<dv2*>(buf + k) = v; }
                }
                asm volatile("" ::: "memory");
                d = buf[0];
#pragma unroll
                for (int k = 0; k < G; k++) p[k] = buf[ag + k];
                asm volatile("" ::: "memory");
            }
            double ri;
            if (VAR == 0 || VAR == 2 || VAR == 3) ri = 1.0 / d;
            else if (VAR == 1) { double r = __builtin_amdgcn_rcp(d); r = r + r * (1.0 - d * r); ri = r; }
            else ri = d * 0.24;
            const double la = (ag != 0) ? p[0] * ri : 0.0;
            if (VAR == 3 || VAR == 4) { wr[0] += 1e-9 * la; }
            else if (ag == 0) {
#pragma unroll
                for (int k = 0; k < G; k++) wr[k] = (k == 0) ? 4.0 + 0.01 * l : 0.1 / (1 + k);
            } else {
#pragma unroll
                for (int k = 0; k < G; k++) wr[k] -= la * p[k];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int k = 0; k < G; k++) s += wr[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
template <int VAR> static void run(const char* name)
{
    double* out; unsigned long long* tk; hipMalloc(&out, 64 * 8); hipMalloc(&tk, 8);
    const int N = 7168;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_chain<VAR>, dim3(1), dim3(64), 0, 0, out, N, tk);
    hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.1f ticks per step\n", name, (double)t / N);
    hipFree(out); hipFree(tk);
}
int main()
{
    run<0>("full step (LDS row, IEEE division, 8 FMA)");
    run<1>("rcp + Newton instead of the division");
    run<2>("no LDS exchange");
    run<3>("division chain only");
    run<4>("LDS exchange only");
    return 0;
}
