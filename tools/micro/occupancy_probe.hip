// How many 256-thread workgroups does a CU of this GPU hold at once, as a function of their LDS size?  Every workgroup notes the wall clock when it
// starts and spins 300 microseconds; the workgroups of the first residency round are the ones that start before the first one ends.
//   hipcc --offload-arch=gfx950 -O2 -o occupancy_probe tools/micro/occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void probe(unsigned long long* start, int touch)
{
    extern __shared__ double lds[];
    const unsigned long long t0 = wall_clock64();
    if (touch >= 0) lds[touch + threadIdx.x] = (double)t0;      // the allocation is used
    while (wall_clock64() - t0 < 30000) { }                     // 100 MHz: 300 microseconds
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
}
int main()
{
    const int G = 4096;
    unsigned long long* d;
    hipMalloc(&d, G * sizeof(unsigned long long));
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %zu B of LDS per workgroup at most, %zu B per CU reported\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
    const int sizes[] = {8192, 16384, 24576, 30720, 32768, 33792, 36352, 36864, 40960, 49152, 53248, 65536};
    for (int s : sizes) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, s);
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), s, 0, d, 0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(G);
        hipMemcpy(h.data(), d, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        int first = 0;
        for (auto v : h) if (v - t0 < 25000) first++;
        printf("LDS %6d B per workgroup: %5d workgroups in the first round = %.2f per CU\n", s, first, (double)first / p.multiProcessorCount);
    }
    return 0;
}
