// Device-resident closed loop around the solve kernels: the body of the reference's Monte-Carlo loop
// (Results/results_linear_system.py:209-259 and results_linear_system_with_extendedMPC.py:247-378)
// as per-trajectory state machines, one thread per trajectory, launched once per time step right
// after the QP solve on the same stream.  Nothing returns to the host between the steps.
//
// What one thread does for its trajectory at time t, in the reference's order:
//   controller packet  U_t = [u_nom | u_bar + K x_bar]                    (TubeTrackingMPC.py:211-227)
//   theta_t            packet controller -> plant lost?                    (results_linear_system.py:218-221)
//   consistent actuator: Theta_t, s_t, buffer, x_nom_0 adoption, u_t       (SmartActuator.py:57-107,146-231)
//   statistics: tracking error, x_t - x_nom_t in Z                         (results_linear_system.py:258,291)
//   plant              x+ = A x + B u + w                                  (:248)
//   gamma_t            packet plant -> controller lost?                    (:223-226)
//   estimator / robust estimator update, q_t                               (Estimator.py:43-98,113-156)
// The estimator's "sequence sent at time s_t" is, by construction, the sequence the actuator
// buffered at time s_t, so the simulated pair shares one buffer instead of the reference's growing list.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "tmpc_device.hpp"
#include "tmpc_mc_step.hpp"

namespace tmpc {

namespace {

using namespace mcstep;

// reference handed to the first solve: ref = [ref_0, 0, ..] (:240); the later ones are written by mc_step_kernel
__global__ void mc_pre_kernel(const McModel m, const McState st, const int64_t B, const double ref_0) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int i = 0; i < m.nx; ++i) st.ref_k[b * m.nx + i] = (i == 0) ? ref_0 : 0.0;
}

constexpr int MC_WPB = 4;           // trajectories (waves) per workgroup

// One WAVE per trajectory and time step: mcstep::mc_step_wave (tmpc_mc_step.hpp) in one launch behind the solve launch(es) of the step
// (round 3: mc_pre + mc_post with a thread per trajectory and 1168 B of private arrays, + mc_tube).
__global__ __launch_bounds__(WAVE_MC * MC_WPB) void mc_step_kernel(const McModel m, const McState st, const int t, const int T, const int64_t B,
                                                                   const double ref_t, const double ref_next,
                                                                   const double *__restrict__ u_nom, const double *__restrict__ x_nom0,
                                                                   const double *__restrict__ xu_ss, const int32_t *__restrict__ status,
                                                                   const int32_t *__restrict__ iters) {
    __shared__ double sh[MC_WPB][V_COUNT][MAXN];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = static_cast<int64_t>(blockIdx.x) * MC_WPB + wave;
    if (b >= B) return;
    (void)mc_step_wave(m, st, t, T, b, ref_t, ref_next, u_nom, x_nom0, xu_ss, status, iters, sh[wave], lane);
}

}  // namespace

// Instances whose variant id names no problem of the handle are solved by no kernel: they get status NUMERICAL and NaN
// outputs instead of whatever the output buffers held.
__global__ void mark_invalid_variants_kernel(const uint8_t *__restrict__ variant, const int nvariants, const int64_t B, const int nun,
                                             const int nx, const int nxu, const int nxn, double *__restrict__ u_nom,
                                             double *__restrict__ x_nom0, double *__restrict__ xu_ss, double *__restrict__ x_nom,
                                             int32_t *__restrict__ status, int32_t *__restrict__ iters) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= B || variant[b] < nvariants) return;
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    for (int i = 0; i < nun; ++i) u_nom[b * nun + i] = nanv;
    if (x_nom0) for (int i = 0; i < nx; ++i) x_nom0[b * nx + i] = nanv;
    if (xu_ss) for (int i = 0; i < nxu; ++i) xu_ss[b * nxu + i] = nanv;
    if (x_nom) for (int i = 0; i < nxn; ++i) x_nom[b * nxn + i] = nanv;
    status[b] = TMPC_STATUS_NUMERICAL;
    iters[b] = 0;
}
hipError_t launch_mark_invalid_variants(const uint8_t *variant, int nvariants, int64_t B, int nx, int nu, int N, double *u_nom,
                                        double *x_nom0, double *xu_ss, double *x_nom, int32_t *status, int32_t *iters,
                                        hipStream_t stream) {
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((B + threads - 1) / threads);
    hipLaunchKernelGGL(mark_invalid_variants_kernel, dim3(blocks), dim3(threads), 0, stream, variant, nvariants, B, N * nu, nx, nx + nu,
                       (N + 1) * nx, u_nom, x_nom0, xu_ss, x_nom, status, iters);
    return hipGetLastError();
}

hipError_t launch_mc_pre(const McModel &m, const McState &st, int64_t B, double ref_0, hipStream_t stream) {
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((B + threads - 1) / threads);
    hipLaunchKernelGGL(mc_pre_kernel, dim3(blocks), dim3(threads), 0, stream, m, st, B, ref_0);
    return hipGetLastError();
}

hipError_t launch_mc_step(const McModel &m, const McState &st, int t, int T, int64_t B, double ref_t, double ref_next, const double *u_nom,
                          const double *x_nom0, const double *xu_ss, const int32_t *status, const int32_t *iters, hipStream_t stream) {
    if (m.nx > MAXN || m.nu > MAXN) return hipErrorInvalidValue;
    const unsigned blocks = static_cast<unsigned>((B + MC_WPB - 1) / MC_WPB);
    hipLaunchKernelGGL(mc_step_kernel, dim3(blocks), dim3(WAVE_MC * MC_WPB), 0, stream, m, st, t, T, B, ref_t, ref_next, u_nom, x_nom0, xu_ss,
                       status, iters);
    return hipGetLastError();
}

}  // namespace tmpc
