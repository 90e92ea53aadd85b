// Where do the workgroups of a small launch land, and on which SIMDs their four waves, as a function of the register count?  (128 registers allow
// four 256-thread workgroups per CU, 136 allow three.)  Every wave reads HW_REG_HW_ID (SIMD id in bits 5:4, CU id in 11:8, shader-array and
// shader-engine ids above) and HW_REG_XCC_ID; the program prints how many CUs a launch of G workgroups touches and the largest number sharing one.
//   hipcc --offload-arch=gfx950 -O2 -o wave_placement_probe tools/micro/wave_placement_probe.hip && ./wave_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
template <int TOP>
__global__ __launch_bounds__(256) void probe(unsigned* out)
{
    if (TOP == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");      // the allocation reaches this register
    else asm volatile("v_mov_b32 v135, 0" ::: "v135");
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) { }      // 200 microseconds: every workgroup of the launch is resident at the same time
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main()
{
    const int GMAX = 1024;
    unsigned* d; hipMalloc(&d, GMAX * 8 * sizeof(unsigned));
    std::vector<unsigned> h(GMAX * 8);
    for (int variant = 0; variant < 2; variant++)
        for (int grid : {1, 4, 16, 64, 128, 256, 512, 768}) {
            if (variant == 0) hipLaunchKernelGGL(probe<127>, dim3(grid), dim3(256), 0, 0, d);
            else hipLaunchKernelGGL(probe<135>, dim3(grid), dim3(256), 0, 0, d);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, sizeof(unsigned) * 8 * grid, hipMemcpyDeviceToHost);
            std::map<unsigned, int> perCU;
            int spread = 0;
            for (int b = 0; b < grid; b++) {
                const unsigned hw = h[8 * b], xcc = h[8 * b + 1] & 15;
                perCU[(xcc << 16) | (hw & 0xff00)]++;      // XCC, then everything of HW_ID above the SIMD and wave ids up to bit 15 (CU, SH, SE)
                unsigned simds = 0;
                for (int w = 0; w < 4; w++) simds |= 1u << ((h[8 * b + 2 * w] >> 4) & 3);
                spread += __builtin_popcount(simds) == 4;
            }
            int mx = 0;
            for (auto& kv : perCU) mx = kv.second > mx ? kv.second : mx;
            printf("%s registers, %4d workgroups: %3zu CUs used, at most %d on one CU; %d workgroups with their four waves on four SIMDs\n",
                   variant == 0 ? "128" : "136", grid, perCU.size(), mx, spread);
        }
    return 0;
}
