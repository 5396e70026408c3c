// The per-trajectory state machines of the closed loop as ONE device function for one wave: everything between two solves of a
// trajectory (see tmpc_mc.hip for the reference lines).  Called by mc_step_kernel (tmpc_mc.hip: one launch per time step behind the
// solve launch) and, inlined, by the fused closed-loop kernel (tmpc_fused.hip: a wave keeps its trajectory for all T steps).
#pragma once
#ifdef TMPC_HOST_SIM
#include "hip_sim.hpp"
#else
#include <hip/hip_runtime.h>
#endif

#include <cmath>
#include <cstdint>

#include "tmpc_device.hpp"

namespace tmpc {
namespace mcstep {

constexpr int MAXN = 16;      // nx, nu <= 16 (tmpc_create enforces nx <= 16; nu checked by the launcher)
constexpr int WAVE_MC = 64;

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
// The step's draws of trajectory b as seen by one lane of its wave: theta and gamma uniforms (every lane), component
// `comp` of the disturbance w (lanes comp < nx; nx <= 16).  Device generator: block j = 0 of the step holds
// [theta, gamma, w_0, w_1], block j >= 1 holds w_{4j-2} .. w_{4j+1} (include/tmpc.h: tmpc_mc_set_device_rng).
template <class StateRec>
__device__ __forceinline__ void mc_draws(const StateRec &st, int64_t b, int t, int T, int nx, int comp, double &th, double &ga, double &w) {
    if (!st.rng_on) {
        th = st.th_u[b * T + t];
        ga = st.ga_u[b * T + t];
        w = comp < nx ? st.w[(b * T + t) * nx + comp] : 0.0;
        return;
    }
    const unsigned long long key1 = static_cast<unsigned long long>(st.rng_first + b);
    unsigned long long r[4];
    philox4x64(static_cast<unsigned long long>(t), 0ull, st.rng_seed, key1, r);
    th = u01(r[0]);
    ga = u01(r[1]);
    const int idx = comp + 2;
    if (idx >= 4) philox4x64(static_cast<unsigned long long>(t), static_cast<unsigned long long>(idx >> 2), st.rng_seed, key1, r);
    const unsigned long long rc = (idx & 3) == 0 ? r[0] : ((idx & 3) == 1 ? r[1] : ((idx & 3) == 2 ? r[2] : r[3]));
    w = comp < nx ? st.w_bound[comp] * (2.0 * u01(rc) - 1.0) : 0.0;
}

// Orders one wave's LDS traffic for the compiler (the hardware runs the DS instructions of a wave in issue order).
#ifdef TMPC_HOST_SIM
__device__ __forceinline__ void mc_fence() { sim::wave_fence(); }      // (tests/wavesim: an LDS hand-over between lanes is a rendezvous)
#else
__device__ __forceinline__ void mc_fence() { asm volatile("" ::: "memory"); }
#endif

enum { V_X = 0, V_XN, V_E, V_ET, V_U, V_UN, V_BASE, V_TMP, V_COUNT };

// One WAVE per trajectory and time step: everything between two solves -- packet, losses, actuator, statistics, tube
// membership, plant, estimator, and the reference of the next solve -- in one launch (round 3: mc_pre + mc_post with a thread
// per trajectory and 1168 B of private arrays, + mc_tube).  Component i of every state-sized vector lives on lane i, the
// vectors a matrix-vector product reads are handed round through LDS (broadcast reads), the rZ rows of the tube
// cross-section are spread over the 64 lanes; the state machine's scalars are wave-uniform.  No private memory.
// The sums run in the order of the round-3 kernels (and of the numpy twins in montecarlo.py).
// (ModelRec, StateRec: McModel / McState, or the same records read through the constant address space -- the fused kernel, where every field is then a
// scalar load at its use instead of a kernel argument that lives in registers for the whole solve)
template <class ModelRec, class StateRec>
__device__ __forceinline__ bool mc_step_wave(const ModelRec &m, const StateRec &st, const int t, const int T, const int64_t b,
                                             const double ref_t, const double ref_next,
                                             const double *u_nom, const double *x_nom0, const double *xu_ss, const int32_t *status,
                                             const int32_t *iters, double (&S)[V_COUNT][MAXN], const int lane,
                                             uint8_t *gamma_out = nullptr /* where this step's arrival flag goes instead of st.gamma */) {
    const int nx = m.nx, nu = m.nu, N = m.N;
    if (st.dead[b]) return false;                                                        // results_linear_system.py:262
    const bool replay = st.rp_U != nullptr;      // packets injected by the caller instead of solved (tmpc_mc_replay)
    const bool lx = lane < nx, lu = lane < nu;
    const double p = st.p_loss[b];
    double th_draw, ga_draw, w_l;
    mc_draws(st, b, t, T, nx, lane, th_draw, ga_draw, w_l);
    int theta = (t > 0 && th_draw < p) ? 0 : 1;                                          // strict <, first packet always arrives
    const int stat = replay ? 0 : status[b];
    const bool bad = stat >= 2;
    if (lane == 0) {
        if (stat != 0) st.not_optimal[b] += 1;
        if (!replay) st.iters_sum[b] += iters[b];
        if (st.ticks) {
            const long long tk = st.ticks[b];
            st.tick_sum[b] += tk;
            if (tk > st.tick_max[b]) st.tick_max[b] = tk;
        }                                                      // results_linear_system.py:305-315 report solve effort
    }
    if (bad) theta = 0;            // a failed solve sends nothing (the reference's tube branch would raise here)
    if (m.smart && bad) {          // R-MPC branch: the trajectory ends here (:268-270), its tracking error is NaN (:297)
        if (lane == 0) {
            st.dead[b] = 1;
            st.err2[b] = __longlong_as_double(0x7ff8000000000000ll);
            if (st.err2_phys) st.err2_phys[b] = __longlong_as_double(0x7ff8000000000000ll);
        }
        return false;
    }
    // the controller's packet of this step: u_0 .. u_{N-1}, the terminal column u_bar + K x_bar (TubeTrackingMPC.py:217), x_nom_0
    const double *pk_u = replay ? st.rp_U + (b * T + t) * static_cast<int64_t>((N + 1) * nu) : u_nom + b * N * nu;
    const double *pk_x0 = replay ? st.rp_xn0 + (b * T + t) * static_cast<int64_t>(nx) : x_nom0 + b * nx;

    const double x_l = lx ? st.x[b * nx + lane] : 0.0;
    double xnom_l = lx ? st.x_nom[b * nx + lane] : 0.0;
    // The tube statistic of the scripts is x_traj[:, t] - x_nom_traj[:, t] (results_linear_system.py:258,
    // results_linear_system_with_extendedMPC.py:331): the nominal state appended after the PREVIOUS step's process_packet,
    // i.e. before this step's adoption of x_nom_0 by the extended controller's actuator.  (The plain smart actuator has no
    // nominal model: its "nominal" state is the measured one.)
    const double et_l = m.smart ? 0.0 : x_l - xnom_l;
    // first input and x_nom_0 of the last sequence SENT (Estimator.py:41: stored whether the packet arrives or not)
    double ul0_l = 0.0, xn0l_l = 0.0;
    if (lu) ul0_l = bad ? st.u_latest0[b * nu + lane] : pk_u[lane];
    if (lx) xn0l_l = bad ? st.x_nom0_latest[b * nx + lane] : pk_x0[lane];
    if (!bad) {
        if (lu) st.u_latest0[b * nu + lane] = ul0_l;
        if (lx) st.x_nom0_latest[b * nx + lane] = xn0l_l;
    }
    // ---- consistent actuator (SmartActuator.py:57-107, 174-231)
    const int q_before = st.q_est[b];
    int last_lost = st.last_lost[b], q_act = st.q_act[b], s_t = st.s[b];
    if (theta == 0) last_lost = t;
    if (theta == 1) q_act = q_before;
    const int Theta = (theta == 1 && last_lost <= q_act) ? 1 : 0;
    if (Theta) s_t = t;
    if (lane == 0) { st.last_lost[b] = last_lost; st.q_act[b] = q_act; st.Theta[b] = Theta; st.s[b] = s_t; }
    double *Ub = st.Ubuf + b * (N + 1) * nu;                  // [i][j], i = 0..N
    if (Theta) {
        for (int i = lane; i < N * nu; i += WAVE_MC) Ub[i] = pk_u[i];
        if (lu) {
            double v;
            if (replay) {
                v = pk_u[N * nu + lane];
            } else {                                           // u_bar + K x_bar (TubeTrackingMPC.py:217)
                v = xu_ss[b * (nx + nu) + nx + lane];
                for (int i = 0; i < nx; ++i) v += m.K[lane * nx + i] * xu_ss[b * (nx + nu) + i];
            }
            Ub[N * nu + lane] = v;
        }
        if (m.extended && lx) xnom_l = pk_x0[lane];
    }
    // the plain smart actuator has no nominal model: its terminal law and its packet use the measured state
    const double xn_l = m.smart ? x_l : xnom_l;
    const double e_l = x_l - xn_l;
    if (lane < MAXN) { S[V_X][lane] = x_l; S[V_XN][lane] = xn_l; S[V_E][lane] = e_l; S[V_ET][lane] = et_l; }
    mc_fence();
    const int d = t - s_t;
    const bool inside = d < N;
    double un_l = 0.0, u_l = 0.0;
    if (lu) {
        // (Theta = 1 means s_t = t: the input is column 0 of the packet just adopted -- read from the packet, not back from the buffer)
        double v = Theta ? pk_u[lane] : Ub[(inside ? d : N) * nu + lane];
        if (!inside)
            for (int i = 0; i < nx; ++i) v -= m.K[lane * nx + i] * S[V_XN][i];
        un_l = v;
        double w2 = v;
        for (int i = 0; i < nx; ++i) w2 -= m.K_anc[lane * nx + i] * S[V_E][i];
        u_l = w2;
    }
    if (lane < MAXN) { S[V_U][lane] = u_l; S[V_UN][lane] = un_l; }
    mc_fence();
    if (b == st.cap_index) {
        // sample run of the scripts (results_linear_system.py:298-301: x_traj, x_nom_traj of one run per loss rate)
        double *c = st.cap + static_cast<size_t>(t) * (2 * nx + nu);
        if (lx) { c[lane] = x_l; c[nx + lane] = x_l - et_l; }
        if (lu) c[2 * nx + lane] = u_l;
    }
    // ---- statistics (results_linear_system.py:258, 291)
    if (lane == 0) {
        double a = (S[V_X][0] - ref_t) * (S[V_X][0] - ref_t);
        for (int i = 1; i < nx; ++i) a += S[V_X][i] * S[V_X][i];
        st.err2[b] += a;
    }
    if (m.rZ > 0) {
        // x_t - x_nom_t in Z (:258): the rZ rows of Z over the lanes
        int out = 0;
        for (int r = lane; r < m.rZ; r += WAVE_MC) {
            double v = -m.hZ[r];
            for (int i = 0; i < nx; ++i) v += m.HZ[r * nx + i] * S[V_ET][i];
            out |= (v > 1e-7);                                  // polytope's abs_tol
        }
        if (__any(out) && lane == 0) st.tube_viol[b] += 1;
    }
    // ---- plant and nominal model
    double xp_l = 0.0, xnp_l = 0.0;
    if (lx) {
        double v = w_l, vn = 0.0;
        for (int k = 0; k < nx; ++k) { v += m.A[lane * nx + k] * S[V_X][k]; vn += m.A[lane * nx + k] * S[V_XN][k]; }
        for (int j = 0; j < nu; ++j) { v += m.B[lane * nu + j] * S[V_U][j]; vn += m.B[lane * nu + j] * S[V_UN][j]; }
        xp_l = v;
        xnp_l = vn;
    }
    if (m.plant == TMPC_PLANT_CARTPOLE) {
        // zero-order hold of u over the sampling period, RK4 at the physics rate (every lane integrates the same four states);
        // the nominal model stays linear
        double y[4] = {S[V_X][0], S[V_X][1], S[V_X][2], S[V_X][3]};
        const double u0 = S[V_U][0];
        double par[7];
        for (int i = 0; i < 7; ++i) par[i] = m.par[i];
        const double dt = par[6] / m.substeps;
        double aphys = 0.0;       // tracking error at the physics rate (results_nonlinear_system.py:361: x_traj[:, 0:-1], 500 Hz)
        for (int sstep = 0; sstep < m.substeps; ++sstep) {
            double k1[4], k2[4], k3[4], k4[4], yt[4];
            aphys += (y[0] - ref_t) * (y[0] - ref_t) + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
            cartpole_rhs(par, y, u0, k1);
            for (int i = 0; i < 4; ++i) yt[i] = y[i] + 0.5 * dt * k1[i];
            cartpole_rhs(par, yt, u0, k2);
            for (int i = 0; i < 4; ++i) yt[i] = y[i] + 0.5 * dt * k2[i];
            cartpole_rhs(par, yt, u0, k3);
            for (int i = 0; i < 4; ++i) yt[i] = y[i] + dt * k3[i];
            cartpole_rhs(par, yt, u0, k4);
            for (int i = 0; i < 4; ++i) y[i] += dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        }
        const double yl = lane == 0 ? y[0] : (lane == 1 ? y[1] : (lane == 2 ? y[2] : y[3]));
        if (lx) xp_l = yl + w_l;
        if (st.err2_phys && lane == 0) st.err2_phys[b] += aphys;
    }
    if (lx) { st.x[b * nx + lane] = xp_l; st.x_nom[b * nx + lane] = xnp_l; }
    // ---- estimator (Estimator.py:43-98; robust: :113-156)
    const int gamma = (t > 0 && ga_draw < p) ? 0 : 1;
    double xh_l = 0.0;
    if (gamma) {
        // packet {'x_t', 's_t'[, 'x_nom_t']}: x_t = nominal state (consistent actuator) or plant state (extended);
        // u_hat(k|k): the input the plant applied / its nominal part
        const int vx = m.extended ? V_X : V_XN, vu = m.extended ? V_U : V_UN;
        if (lx) {
            double v = 0.0;
            for (int k = 0; k < nx; ++k) v += m.A[lane * nx + k] * S[vx][k];
            for (int j = 0; j < nu; ++j) v += m.B[lane * nu + j] * S[vu][j];
            xh_l = v;
        }
        if (lane == 0) st.q_est[b] = t;
    } else {
        const double base_l = lx ? (m.extended ? xn0l_l : st.x_hat[b * nx + lane]) : 0.0;
        if (lane < MAXN) { S[V_BASE][lane] = base_l; S[V_TMP][lane] = ul0_l; }
        mc_fence();
        if (lx) {
            double v = 0.0;
            for (int k = 0; k < nx; ++k) v += m.A[lane * nx + k] * S[V_BASE][k];
            for (int j = 0; j < nu; ++j) v += m.B[lane * nu + j] * S[V_TMP][j];
            xh_l = v;
        }
        mc_fence();
    }
    if (lx) st.x_hat[b * nx + lane] = xh_l;
    if (Theta && gamma && !m.extended) {                       // Proposition 1
        if (lane < MAXN) S[V_TMP][lane] = lx ? fabs(xh_l - xnp_l) : 0.0;
        mc_fence();
        if (lane == 0) {
            double ce = 0.0;
            for (int i = 0; i < nx; ++i) ce = fmax(ce, S[V_TMP][i]);
            st.consistent[b] = fmax(st.consistent[b], ce);
        }
    }
    if (lane == 0) (gamma_out ? gamma_out : st.gamma)[b] = static_cast<uint8_t>(gamma);
    if (lx) st.ref_k[b * nx + lane] = (lane == 0) ? ref_next : 0.0;      // ref = [ref_{t+1}, 0, ..] of the next solve (:240)
    if (st.trace_f) {
        // every step of every trajectory (tmpc_mc_replay): x_{t+1}, x_hat_{t+1}, the nominal state of the plant's packet, u_t;
        // s_t, Theta_t and the q_t the controller put into its packet
        double *tf = st.trace_f + (b * T + t) * static_cast<int64_t>(3 * nx + nu);
        if (lx) { tf[lane] = xp_l; tf[nx + lane] = xh_l; tf[2 * nx + lane] = xn_l; }
        if (lu) tf[3 * nx + lane] = u_l;
        if (lane == 0) { int32_t *ti = st.trace_i + (b * T + t) * 3; ti[0] = s_t; ti[1] = Theta; ti[2] = q_before; }
    }
    return true;
}


}  // namespace mcstep
}  // namespace tmpc
