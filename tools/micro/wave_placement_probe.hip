// On which SIMDs do the four waves of ONE 256-thread workgroup land, as a function of its register count?  (128 registers allow four waves per SIMD,
// 136 allow three.)  Every wave reads HW_REG_HW_ID (SIMD id in bits 5:4, CU id in 11:8).
//   hipcc --offload-arch=gfx950 -O2 -o wave_placement_probe tools/micro/wave_placement_probe.hip && ./wave_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TOP>
__global__ __launch_bounds__(256) void probe(unsigned* out)
{
    if (TOP == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");      // the allocation reaches this register
    else asm volatile("v_mov_b32 v135, 0" ::: "v135");
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
}
int main()
{
    unsigned* d; hipMalloc(&d, 64 * sizeof(unsigned));
    unsigned h[64];
    for (int variant = 0; variant < 2; variant++)
        for (int grid : {1, 2, 4}) {
            if (variant == 0) hipLaunchKernelGGL(probe<127>, dim3(grid), dim3(256), 0, 0, d);
            else hipLaunchKernelGGL(probe<135>, dim3(grid), dim3(256), 0, 0, d);
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(unsigned) * 4 * grid, hipMemcpyDeviceToHost);
            printf("%s registers, %d workgroup(s):", variant == 0 ? "128" : "136", grid);
            for (int b = 0; b < grid; b++) {
                printf("  [CU %u:", (h[4 * b] >> 8) & 15);
                for (int w = 0; w < 4; w++) printf(" SIMD %u", (h[4 * b + w] >> 4) & 3);
                printf("]");
            }
            printf("\n");
        }
    return 0;
}
