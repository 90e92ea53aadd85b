// lcqp_launch.hpp -- the seam between the host translation unit (lcqp_hip.hip) and the per-size kernel translation units
// (lcqp_nch.hip, one per NCH in {1,2,3,4,8}).
#pragma once
#include "lcqp_dev.hpp"

namespace lcqp {

enum KernelId { ID_k_prepare, ID_k_build_C, ID_k_compress_C, ID_k_factor, ID_k_factor_full, ID_k_trsm, ID_k_trsm_streamed, ID_k_build_M, ID_k_lcqp_run, ID_k_qp_solve,
                ID_k_synth_fill, ID_k_synth_Q, ID_k_util_symv, ID_k_util_rows, ID_k_util_rows_list };

struct LaunchArgs {
    DevBatch db;
    const int* list = nullptr;
    int initial = 0;
    uint64_t seed0 = 0, first = 0;
    // building-block kernels
    int n = 0, m = 0;
    double alpha = 0.0;
    const double *A = nullptr, *b = nullptr, *c = nullptr, *x = nullptr, *coef = nullptr;
    double *d = nullptr, *dots = nullptr, *outT = nullptr;
};

// defined in lcqp_nch.hip for NCH = LCQP_TU_NCH
void lcqp_launch_1(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_2(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_3(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_4(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_8(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_16(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_32(int kid, int grid, hipStream_t s, const LaunchArgs& a);
// k_lcqp_run and k_qp_solve for batches of at most three workgroups per CU (lcqp_nch.hip with -DLCQP_TU_FEW), np <= 512
void lcqp_launch_few_1(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_few_2(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_few_3(int kid, int grid, hipStream_t s, const LaunchArgs& a);
void lcqp_launch_few_4(int kid, int grid, hipStream_t s, const LaunchArgs& a);

}  // namespace lcqp
