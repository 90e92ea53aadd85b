// lcqp_nch.hip -- one instantiation of the per-size kernels: compile with -DLCQP_TU_NCH=k, k in {1,2,3,4,8,16,32}.
#include "lcqp_kernels.hpp"

#ifndef LCQP_TU_NCH
#error "compile lcqp_nch.hip with -DLCQP_TU_NCH=1|2|3|4|8|16|32"
#endif
#define LCQP_CAT2(a, b) a##b
#define LCQP_CAT(a, b) LCQP_CAT2(a, b)

namespace lcqp {
#ifdef LCQP_TU_FEW      // the second build of the persistent kernels (batches of at most three workgroups per CU; lcqp_kernels.hpp): with
                        // -DLCQP_VARIANT=1 -DLCQP_MINWAVES=2
void LCQP_CAT(lcqp_launch_few_, LCQP_TU_NCH)(int kid, int grid, hipStream_t s, const LaunchArgs& a) { launch_impl<LCQP_TU_NCH>(kid, grid, s, a); }
#else
void LCQP_CAT(lcqp_launch_, LCQP_TU_NCH)(int kid, int grid, hipStream_t s, const LaunchArgs& a) { launch_impl<LCQP_TU_NCH>(kid, grid, s, a); }
#endif
}
