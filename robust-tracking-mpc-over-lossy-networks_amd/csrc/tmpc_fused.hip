// The fused closed-loop instantiations of the wave kernel (solve_kernel<..., MC = true>, tmpc_kernels.hip) as a translation unit of
// their own: `make -j` compiles them next to the plain ones instead of doubling the longest compile of the build.
#define TMPC_FUSED_TU 1
#include "tmpc_kernels.hip"
