// Micro-benchmark: HBM read bandwidth of the access pattern of the sparse engine -- a wavefront reads 8 streams (one per instance,
// far apart) 64 bytes at a time -- against one contiguous 512-byte stream per wavefront.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" {
// K > 1: the wavefront alternates between K arrays that lie far apart (the vectors, factors and values of the sparse engine are
// separate allocations of B instances each); blocked != 0: the K pieces of a wavefront lie next to each other instead
__global__ void k_read_multi(const double* __restrict__ a, size_t perInst, int nIter, int K, int blocked, int waves, double* out)
{
    const int lane = threadIdx.x & 63, gi = lane >> 3, gl = lane & 7;
    const size_t wave = blockIdx.x;
    double s = 0.0;
    const size_t per = perInst / K;                        // doubles per instance and array
    for (int k = 0; k < nIter; k += 4) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int it = k + u, arr = it % K, pos = it / K;
            const size_t base = blocked ? ((wave * K + arr) * 8 + gi) * per : (((size_t)arr * waves + wave) * 8 + gi) * per;
            v[u] = a[base + (size_t)pos * 8 + gl];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) s += v[u];
    }
    if (s == 123.456) out[0] = s;
}
__global__ void k_read(const double* __restrict__ a, size_t perInst, int nIter, int interleaved, double* out)
{
    const int lane = threadIdx.x & 63, gi = lane >> 3, gl = lane & 7;
    const size_t wave = blockIdx.x;
    double s = 0.0;
    if (interleaved) {
        const double* p = a + wave * 8 * perInst;            // the 8 instances of the wave interleaved at 64 bytes
        for (int k = 0; k < nIter; k += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = p[(size_t)(k + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 4; u++) s += v[u];
        }
    } else {
        const double* p = a + (wave * 8 + gi) * perInst;      // each lane group its own array
        for (int k = 0; k < nIter; k += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = p[(size_t)(k + u) * 8 + gl];
#pragma unroll
            for (int u = 0; u < 4; u++) s += v[u];
        }
    }
    if (s == 123.456) out[0] = s;
}
int run_multi(int waves, long perInst, int K, int blocked, int reps, float* ms)
{
    double *a, *out;
    const size_t total = (size_t)waves * 8 * perInst;
    if (hipMalloc(&a, total * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    (void)hipMemset(a, 0, total * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int nIter = (int)(perInst / 8);
    hipLaunchKernelGGL(k_read_multi, dim3(waves), dim3(64), 0, 0, a, (size_t)perInst, nIter, K, blocked, waves, out);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read_multi, dim3(waves), dim3(64), 0, 0, a, (size_t)perInst, nIter, K, blocked, waves, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(ms, e0, e1); *ms /= reps;
    (void)hipFree(a); (void)hipFree(out);
    return 0;
}
int run(int waves, long perInst, int interleaved, int reps, float* ms)
{
    double *a, *out;
    const size_t total = (size_t)waves * 8 * perInst;
    if (hipMalloc(&a, total * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    hipMemset(a, 0, total * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nIter = (int)(perInst / 8);
    hipLaunchKernelGGL(k_read, dim3(waves), dim3(64), 0, 0, a, (size_t)perInst, nIter, interleaved, out);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3(waves), dim3(64), 0, 0, a, (size_t)perInst, nIter, interleaved, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(ms, e0, e1); *ms /= reps;
    hipFree(a); hipFree(out);
    return 0;
}
}
