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

namespace tmpc {

namespace {

constexpr int MAXN = 16;      // nx, nu <= 16 (tmpc_create enforces nx <= 16; nu checked by the launcher)

// cart-pole about the upright position, x = [pos, vel, angle, angular velocity] (LinearMPCOverNetworks/workloads.py:
// cartpole_rhs has the derivation and the numpy twin)
__device__ __forceinline__ void cartpole_rhs(const double *par, const double (&x)[4], double F, double (&dx)[4]) {
    const double M = par[0], m = par[1], b = par[2], I = par[3], g = par[4], l = par[5];
    const double s = sin(x[2]), c = cos(x[2]);
    const double a11 = M + m, a12 = m * l * c, a22 = I + m * l * l;
    const double r1 = F - b * x[1] + m * l * x[3] * x[3] * s;
    const double r2 = m * g * l * s;
    const double det = a11 * a22 - a12 * a12;
    dx[0] = x[1];
    dx[1] = (r1 * a22 - a12 * r2) / det;
    dx[2] = x[3];
    dx[3] = (a11 * r2 - a12 * r1) / det;
}

// Philox4x64-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind
// numpy.random.Philox, against which montecarlo.philox4x64 -- the numpy twin of this function -- is pinned in the tests).
__device__ __forceinline__ void philox4x64(unsigned long long c0, unsigned long long c1, unsigned long long k0, unsigned long long k1,
                                           unsigned long long (&out)[4]) {
    unsigned long long c[4] = {c0, c1, 0ull, 0ull};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = 0xD2E7470EE14C6C93ull, m1 = 0xCA5A826395121157ull;
        const unsigned long long hi0 = __umul64hi(m0, c[0]), lo0 = m0 * c[0];
        const unsigned long long hi1 = __umul64hi(m1, c[2]), lo1 = m1 * c[2];
        c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
        k0 += 0x9E3779B97F4A7C15ull; k1 += 0xBB67AE8584CAA73Bull;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}
__device__ __forceinline__ double u01(unsigned long long x) { return static_cast<double>(x >> 11) * 0x1.0p-53; }   // [0, 1), 53 bits
// the step's draws of trajectory b: theta and gamma uniforms, disturbance w (nx <= 16)
__device__ __forceinline__ void mc_draws(const McState &st, int64_t b, int t, int T, int nx, double &th, double &ga, double *w) {
    if (!st.rng_on) {
        th = st.th_u[b * T + t];
        ga = st.ga_u[b * T + t];
        for (int i = 0; i < nx; ++i) w[i] = st.w[(b * T + t) * nx + i];
        return;
    }
    const unsigned long long key1 = static_cast<unsigned long long>(st.rng_first + b);
    unsigned long long r[4];
    philox4x64(static_cast<unsigned long long>(t), 0ull, st.rng_seed, key1, r);
    th = u01(r[0]);
    ga = u01(r[1]);
    for (int i = 0; i < nx; ++i) {
        const int idx = i + 2;
        if (idx >= 4 && (idx & 3) == 0) philox4x64(static_cast<unsigned long long>(t), static_cast<unsigned long long>(idx >> 2), st.rng_seed, key1, r);
        w[i] = st.w_bound[i] * (2.0 * u01(r[idx & 3]) - 1.0);
    }
}

__global__ void mc_pre_kernel(const McModel m, const McState st, const int t, const int64_t B, const double ref_t) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int i = 0; i < m.nx; ++i) st.ref_k[b * m.nx + i] = (i == 0) ? ref_t : 0.0;     // ref = [ref_t, 0, ..] (:240)
}

__global__ void mc_post_kernel(const McModel m, const McState st, const int t, const int T, const int64_t B, const double ref_t,
                               const double *__restrict__ u_nom, const double *__restrict__ x_nom0,
                               const double *__restrict__ xu_ss, const int32_t *__restrict__ status,
                               const int32_t *__restrict__ iters) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nx = m.nx, nu = m.nu, N = m.N;
    if (st.dead[b]) return;                                                              // results_linear_system.py:262
    const double p = st.p_loss[b];
    double th_draw, ga_draw, w_draw[MAXN];
    mc_draws(st, b, t, T, nx, th_draw, ga_draw, w_draw);
    int theta = (t > 0 && th_draw < p) ? 0 : 1;                                          // strict <, first packet always arrives
    const int stat = status[b];
    const bool bad = stat >= 2;
    if (stat != 0) st.not_optimal[b] += 1;
    st.iters_sum[b] += iters[b];
    if (st.ticks) {
        const long long tk = st.ticks[b];
        st.tick_sum[b] += tk;
        if (tk > st.tick_max[b]) st.tick_max[b] = tk;
    }                                                          // results_linear_system.py:305-315 report solve effort
    if (bad) theta = 0;            // a failed solve sends nothing (the reference's tube branch would raise here)
    if (m.smart && bad) {          // R-MPC branch: the trajectory ends here (:268-270), its tracking error is NaN (:297)
        st.dead[b] = 1;
        st.err2[b] = __longlong_as_double(0x7ff8000000000000ll);
        if (st.err2_phys) st.err2_phys[b] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }

    double x[MAXN], xn[MAXN], e[MAXN];
    for (int i = 0; i < nx; ++i) x[i] = st.x[b * nx + i];
    // The tube statistic of the scripts is x_traj[:, t] - x_nom_traj[:, t] (results_linear_system.py:258,
    // results_linear_system_with_extendedMPC.py:331): the nominal state appended after the PREVIOUS step's process_packet,
    // i.e. before this step's adoption of x_nom_0 by the extended controller's actuator.  (The plain smart actuator has no
    // nominal model: its "nominal" state is the measured one.)
    for (int i = 0; i < nx; ++i) st.e_buf[b * nx + i] = m.smart ? 0.0 : x[i] - st.x_nom[b * nx + i];
    if (!bad) {
        for (int j = 0; j < nu; ++j) st.u_latest0[b * nu + j] = u_nom[b * N * nu + j];
        for (int i = 0; i < nx; ++i) st.x_nom0_latest[b * nx + i] = x_nom0[b * nx + i];
    }
    // ---- consistent actuator (SmartActuator.py:57-107, 174-231)
    if (theta == 0) st.last_lost[b] = t;
    if (theta == 1) st.q_act[b] = st.q_est[b];
    const int Theta = (theta == 1 && st.last_lost[b] <= st.q_act[b]) ? 1 : 0;
    st.Theta[b] = Theta;
    double *Ub = st.Ubuf + b * (N + 1) * nu;                  // [i][j], i = 0..N
    if (Theta) {
        st.s[b] = t;
        for (int i = 0; i < N * nu; ++i) Ub[i] = u_nom[b * N * nu + i];
        for (int j = 0; j < nu; ++j) {                         // u_bar + K x_bar (TubeTrackingMPC.py:217)
            double v = xu_ss[b * (nx + nu) + nx + j];
            for (int i = 0; i < nx; ++i) v += m.K[j * nx + i] * xu_ss[b * (nx + nu) + i];
            Ub[N * nu + j] = v;
        }
        if (m.extended)
            for (int i = 0; i < nx; ++i) st.x_nom[b * nx + i] = x_nom0[b * nx + i];
    }
    // the plain smart actuator has no nominal model: its terminal law and its packet use the measured state
    for (int i = 0; i < nx; ++i) { xn[i] = m.smart ? x[i] : st.x_nom[b * nx + i]; e[i] = x[i] - xn[i]; }
    const int d = t - st.s[b];
    const bool inside = d < N;
    double un[MAXN], u[MAXN];
    for (int j = 0; j < nu; ++j) {
        double v = Ub[(inside ? d : N) * nu + j];
        if (!inside)
            for (int i = 0; i < nx; ++i) v -= m.K[j * nx + i] * xn[i];
        un[j] = v;
        double w2 = v;
        for (int i = 0; i < nx; ++i) w2 -= m.K_anc[j * nx + i] * e[i];
        u[j] = w2;
    }
    if (b == st.cap_index) {
        // sample run of the scripts (results_linear_system.py:298-301: x_traj, x_nom_traj of one run per loss rate)
        double *c = st.cap + static_cast<size_t>(t) * (2 * nx + nu);
        for (int i = 0; i < nx; ++i) { c[i] = x[i]; c[nx + i] = x[i] - st.e_buf[b * nx + i]; }
        for (int j = 0; j < nu; ++j) c[2 * nx + j] = u[j];
    }
    // ---- statistics (results_linear_system.py:258, 291)
    {
        double a = (x[0] - ref_t) * (x[0] - ref_t);
        for (int i = 1; i < nx; ++i) a += x[i] * x[i];
        st.err2[b] += a;
        // x_t - x_nom_t in Z (:258) is checked by mc_tube_kernel on e_buf (rZ rows x nx columns per trajectory: too long a loop
        // for one thread of this kernel, which only has a wave's worth of parallelism per 64 trajectories)
    }
    // ---- plant and nominal model
    double xp[MAXN], xnp[MAXN];
    for (int i = 0; i < nx; ++i) {
        double v = w_draw[i], vn = 0.0;
        for (int k = 0; k < nx; ++k) { v += m.A[i * nx + k] * x[k]; vn += m.A[i * nx + k] * xn[k]; }
        for (int j = 0; j < nu; ++j) { v += m.B[i * nu + j] * u[j]; vn += m.B[i * nu + j] * un[j]; }
        xp[i] = v;
        xnp[i] = vn;
    }
    if (m.plant == TMPC_PLANT_CARTPOLE) {
        // zero-order hold of u over the sampling period, RK4 at the physics rate; the nominal model stays linear
        double y[4] = {x[0], x[1], x[2], x[3]};
        const double dt = m.par[6] / m.substeps;
        double aphys = 0.0;       // tracking error at the physics rate (results_nonlinear_system.py:361: x_traj[:, 0:-1], 500 Hz)
        for (int sstep = 0; sstep < m.substeps; ++sstep) {
            double k1[4], k2[4], k3[4], k4[4], yt[4];
            aphys += (y[0] - ref_t) * (y[0] - ref_t) + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
            cartpole_rhs(m.par, y, u[0], k1);
            for (int i = 0; i < 4; ++i) yt[i] = y[i] + 0.5 * dt * k1[i];
            cartpole_rhs(m.par, yt, u[0], k2);
            for (int i = 0; i < 4; ++i) yt[i] = y[i] + 0.5 * dt * k2[i];
            cartpole_rhs(m.par, yt, u[0], k3);
            for (int i = 0; i < 4; ++i) yt[i] = y[i] + dt * k3[i];
            cartpole_rhs(m.par, yt, u[0], k4);
            for (int i = 0; i < 4; ++i) y[i] += dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        }
        for (int i = 0; i < 4; ++i) xp[i] = y[i] + w_draw[i];
        if (st.err2_phys) st.err2_phys[b] += aphys;
    }
    for (int i = 0; i < nx; ++i) { st.x[b * nx + i] = xp[i]; st.x_nom[b * nx + i] = xnp[i]; }
    // ---- estimator (Estimator.py:43-98; robust: :113-156)
    const int gamma = (t > 0 && ga_draw < p) ? 0 : 1;
    double xh[MAXN];
    if (gamma) {
        // packet {'x_t', 's_t'[, 'x_nom_t']}: x_t = nominal state (consistent actuator) or plant state (extended)
        const double *xpk = m.extended ? x : xn;
        const double *ue = m.extended ? u : un;               // u_hat(k|k): the input the plant applied / its nominal part
        for (int i = 0; i < nx; ++i) {
            double v = 0.0;
            for (int k = 0; k < nx; ++k) v += m.A[i * nx + k] * xpk[k];
            for (int j = 0; j < nu; ++j) v += m.B[i * nu + j] * ue[j];
            xh[i] = v;
        }
        st.q_est[b] = t;
    } else {
        const double *base = m.extended ? st.x_nom0_latest + b * nx : st.x_hat + b * nx;
        double bs[MAXN];
        for (int i = 0; i < nx; ++i) bs[i] = base[i];
        for (int i = 0; i < nx; ++i) {
            double v = 0.0;
            for (int k = 0; k < nx; ++k) v += m.A[i * nx + k] * bs[k];
            for (int j = 0; j < nu; ++j) v += m.B[i * nu + j] * st.u_latest0[b * nu + j];
            xh[i] = v;
        }
    }
    double ce = 0.0;
    for (int i = 0; i < nx; ++i) { st.x_hat[b * nx + i] = xh[i]; ce = fmax(ce, fabs(xh[i] - xnp[i])); }
    if (Theta && gamma && !m.extended) st.consistent[b] = fmax(st.consistent[b], ce);     // Proposition 1
    st.gamma[b] = static_cast<uint8_t>(gamma);
}

// one workgroup per trajectory: the rZ rows of Z are spread over the threads
__global__ void mc_tube_kernel(const McModel m, const McState st, const int64_t B) {
    const int64_t b = blockIdx.x;
    if (b >= B || st.dead[b]) return;
    const int nx = m.nx;
    double e[MAXN];
    for (int i = 0; i < nx; ++i) e[i] = st.e_buf[b * nx + i];
    int out = 0;
    for (int r = threadIdx.x; r < m.rZ; r += blockDim.x) {
        double v = -m.hZ[r];
        for (int i = 0; i < nx; ++i) v += m.HZ[r * nx + i] * e[i];
        out |= (v > 1e-7);                                  // polytope's abs_tol
    }
    out = __syncthreads_or(out);
    if (threadIdx.x == 0 && out) st.tube_viol[b] += 1;
}

}  // namespace

hipError_t launch_mc_tube(const McModel &m, const McState &st, int64_t B, hipStream_t stream) {
    if (m.rZ <= 0) return hipSuccess;
    hipLaunchKernelGGL(mc_tube_kernel, dim3(static_cast<unsigned>(B)), dim3(128), 0, stream, m, st, B);
    return hipGetLastError();
}

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

hipError_t launch_mc_pre(const McModel &m, const McState &st, int t, int64_t B, double ref_t, hipStream_t stream) {
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((B + threads - 1) / threads);
    hipLaunchKernelGGL(mc_pre_kernel, dim3(blocks), dim3(threads), 0, stream, m, st, t, B, ref_t);
    return hipGetLastError();
}

hipError_t launch_mc_post(const McModel &m, const McState &st, int t, int T, int64_t B, double ref_t, const double *u_nom,
                          const double *x_nom0, const double *xu_ss, const int32_t *status, const int32_t *iters, hipStream_t stream) {
    if (m.nx > MAXN || m.nu > MAXN) return hipErrorInvalidValue;
    const int threads = 64;
    const unsigned blocks = static_cast<unsigned>((B + threads - 1) / threads);
    hipLaunchKernelGGL(mc_post_kernel, dim3(blocks), dim3(threads), 0, stream, m, st, t, T, B, ref_t, u_nom, x_nom0, xu_ss, status, iters);
    return hipGetLastError();
}

}  // namespace tmpc
