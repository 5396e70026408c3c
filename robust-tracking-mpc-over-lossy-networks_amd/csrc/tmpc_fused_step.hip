// The step-fused closed-loop instantiations of the wave kernel (closed_loop_step_kernel, tmpc_kernels.hip) as a translation unit of
// their own, like tmpc_fused.hip.
#define TMPC_FUSED_TU 2
#include "tmpc_kernels.hip"
