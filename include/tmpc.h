/*
 * tmpc.h -- C ABI of the MI355X tube-tracking-MPC solve engine (libtmpc_hip.so).
 *
 * Drop-in boundary for ONE hot path of
 * EricssonResearch/Robust-Tracking-MPC-over-Lossy-Networks: the per-timestep QP of
 * TubeTrackingMPC / ExtendedTubeTrackingMPC, batched over independent
 * Monte-Carlo trajectories.  Each entry point names the reference interface it
 * replaces (paths relative to the reference's src/LinearMPCOverNetworks/).
 *
 * Conventions
 *   - plain C, no C++/torch types; every matrix is float64, row-major, dense;
 *   - the caller owns all buffers passed in; the library owns the opaque handle,
 *     its device copies of the problem, its scratch and its HIP stream;
 *   - every function returns 0 on success and a negative TMPC_E_* code otherwise;
 *     tmpc_last_error() gives the message; no C++ exception crosses the boundary;
 *   - a handle is not thread-safe; use one handle per host thread / per GPU.
 */
#ifndef TMPC_H
#define TMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: first cut; 2: projected terminal rows of the packet-received problem (HTP, hTP, rTP); 3: tmpc_lp_batch, TMPC_STATUS_UNBOUNDED;
 * 4: terminal_equality, tmpc_kernel_name; 5: an iterate that hits the iteration cap keeps TMPC_STATUS_MAX_ITER whatever its
 * constraint violation (INFEASIBLE only with a Farkas-type certificate), host-only handles report their kernel path,
 * tmpc_debug_dump_layout (tmpc_debug_dump_lp_layout was added later without a bump: a new export, nothing else changed) */
#define TMPC_ABI_VERSION 5

/* error codes (function return values) */
#define TMPC_OK            0
#define TMPC_E_INVALID    -1   /* bad argument / inconsistent problem description      */
#define TMPC_E_UNSUPPORTED -2  /* problem size outside what the compiled kernels cover */
#define TMPC_E_DEVICE     -3   /* HIP runtime error                                     */
#define TMPC_E_NOMEM      -4

/* per-instance solver status (status[] output), mirroring what the reference reads
 * from cvxpy: "optimal" / "optimal_inaccurate" / "infeasible" / failure -> None
 * (TubeTrackingMPC.py:185-194) */
#define TMPC_STATUS_OPTIMAL     0
#define TMPC_STATUS_MAX_ITER    1   /* iteration cap hit; last iterate returned          */
#define TMPC_STATUS_INFEASIBLE  2   /* no x satisfies the constraints for this (x_k)     */
#define TMPC_STATUS_NUMERICAL   3
#define TMPC_STATUS_UNBOUNDED   4   /* tmpc_lp_batch only: the objective is unbounded on the set */

/*
 * Problem description: everything `TubeTrackingMPC.generate_optimization_problem`
 * (TubeTrackingMPC.py:104-156) and, when `extended` is set,
 * `ExtendedTubeTrackingMPC.generate_optimization_problem_when_packet_received`
 * (TubeTrackingMPC.py:253-299) close over.
 *
 *   model     x+ = A x + B u                      A: nx*nx   B: nx*nu     (RegulatorMPC.py:13-14)
 *   weights   Q: nx*nx  R: nu*nu                                          (RegulatorMPC.py:24-25)
 *             P: nx*nx  terminal weight                                   (TubeRegulatorMPC.py:23)
 *             T: nx*nx  steady-state offset weight = 10 P                 (TubeTrackingMPC.py:27)
 *   gains     K: nu*nx steady-state LQR gain, K_anc: nu*nx ancillary gain (TubeTrackingMPC.py:229-240)
 *   sets      {Hx x <= hx}   rx rows, tightened state set  Xc            (TubeTrackingMPC.py:112)
 *             {Hu u <= hu}   ru rows, tightened input set  Uc            (TubeTrackingMPC.py:110)
 *             {HT [x_N; x_bar; u_bar] <= hT}  rT rows, terminal set Xf in R^(2nx+nu) (TubeTrackingMPC.py:114,149)
 *             {HZ e <= hZ}   rZ rows, mRPI set Z; used iff fixed_x0 == 0  (TubeTrackingMPC.py:130-132)
 *             {HZW e <= hZW} rZW rows, Z (-) W; used iff extended != 0    (TubeTrackingMPC.py:266-278)
 *             {HTP [x_bar; u_bar] <= hTP} rTP rows (optional, extended only): the terminal row block of the
 *                            packet-received problem (TubeTrackingMPC.py:293) with its free auxiliaries
 *                            eliminated, i.e. proj(Xf) on the steady-state subspace -- see below
 *
 * fixed_x0 : 1 -> x_0 == x_k (TubeTrackingMPC.py:127); 0 -> HZ (x_k - x_0) <= hZ (TubeTrackingMPC.py:132).
 * extended : 1 -> a second QP ("variant 1") is prepared for instances whose
 *            previous plant packet arrived (gamma_t == 1, TubeTrackingMPC.py:312).
 * literal_terminal_row : 1 (default) reproduces TubeTrackingMPC.py:293 literally: in
 *            variant 1 the terminal inequality is written on x_mpc[:,N] and u_bar of the
 *            *base* problem, i.e. on free auxiliary variables; 0 uses the variant's own
 *            x_N and u_bar instead.
 *            The auxiliaries are not priced, so line :293 only says x_bar in proj_xbar(Xf).  When the
 *            caller supplies that projection (HTP, hTP; LinearMPCOverNetworks/utils_polytope.py:
 *            eliminate_terminal_auxiliaries computes it exactly at set-up time) the auxiliaries are
 *            dropped and the QP stays strictly convex: the minimiser in (x, u, x_bar, u_bar) is the
 *            same.  Without it (rTP == 0) the auxiliaries are kept with a vanishing weight
 *            2e-6 min(diag R) |aux|^2; that problem is badly conditioned and its solution is only
 *            reliable to ~1e-5.
 * tol      : relative primal-residual / duality-gap level at which the interior-point
 *            phase hands over to the exact active-set refinement; <= 0 selects the
 *            default 1e-7 (tightened by 1e-2 and retried whenever the refinement
 *            cannot certify its active set).
 * max_iter : interior-point iteration cap; <= 0 selects the default 60.
 */
typedef struct tmpc_problem {
    int32_t nx, nu, N;
    int32_t rx, ru, rT, rZ, rZW;
    int32_t fixed_x0, extended, literal_terminal_row;
    int32_t max_iter;
    double  tol;
    const double *A, *B, *Q, *R, *P, *T, *K, *K_anc;
    const double *Hx, *hx, *Hu, *hu, *HT, *hT, *HZ, *hZ, *HZW, *hZW;
    const double *HTP, *hTP;    /* rTP x (nx+nu), rTP; may be NULL */
    int32_t rTP;
    int32_t terminal_equality;  /* 1: x_N == x_bar instead of a terminal set (TrackingMPC.py:105-107: the tracking MPC
                                 * before setup_optimization()); needs rT == 0.  The nx equalities are eliminated at set-up
                                 * like the dynamics and the steady-state equation. */
} tmpc_problem;

typedef struct tmpc_handle tmpc_handle;

/* ABI version of the loaded library (compare with TMPC_ABI_VERSION). */
int tmpc_abi_version(void);

/* Last error message of `h`, or of the last failed tmpc_create when h == NULL. */
const char *tmpc_last_error(const tmpc_handle *h);

/*
 * Replaces generate_optimization_problem(fixed_initial_state)
 * (TubeTrackingMPC.py:104-156) [+ :253-299 when p->extended]: condenses the QP(s)
 * once, uploads them to HIP device `device`, allocates scratch.
 * device < 0 builds a host-only handle: the condensed problem can be inspected
 * (tmpc_get_dims / tmpc_get_condensed) but every solve call fails with TMPC_E_DEVICE --
 * there is no CPU solve path in this library.
 */
int tmpc_create(const tmpc_problem *p, int device, tmpc_handle **out);

void tmpc_destroy(tmpc_handle *h);

/*
 * Replaces solve_optimization_problem(x_init, ref[, gamma_t])
 * (TubeTrackingMPC.py:170-194, :307-349) for B independent instances.
 *
 *   in   x_k     B*nx     state estimate handed to the controller (x_init)
 *        ref     B*nx     reference state                          (ref)
 *        variant B or NULL  0 = base problem, 1 = packet-received problem (gamma_t)
 *   out  u_nom   B*N*nu   nominal inputs u_0..u_{N-1}   (reference returns its transpose, nu*N)
 *        x_nom0  B*nx     x_nom[:,0]                     (TubeTrackingMPC.py:364)
 *        xu_ss   B*(nx+nu) [x_bar | u_bar]               (TubeTrackingMPC.py:191-192)
 *        x_nom   B*(N+1)*nx or NULL  full nominal state trajectory
 *        status  B        TMPC_STATUS_*
 *        iters   B        interior-point iterations used
 *
 * All pointers are HOST pointers; the call copies in, launches, copies out and
 * returns when the results are in place.  Outputs of instances with
 * status >= TMPC_STATUS_INFEASIBLE are NaN (the reference returns None).
 */
int tmpc_solve_batch(tmpc_handle *h, int64_t B,
                     const double *x_k, const double *ref, const uint8_t *variant,
                     double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                     int32_t *status, int32_t *iters);

/*
 * Same, with every pointer a DEVICE pointer on the handle's device (e.g. a torch
 * tensor's data_ptr()).  The kernels are enqueued on the handle's stream and the
 * call returns without synchronising; use tmpc_synchronize() or
 * tmpc_last_kernel_ms().
 *
 * Ordering contract: the handle's stream is its own non-blocking stream; it is NOT ordered against
 * the stream that produced the inputs or will consume the outputs (e.g. torch's current stream).
 * The caller synchronises on both sides: the producers of x_k / ref / variant must have completed
 * before this call (torch.cuda.synchronize() or an event wait), and the outputs may be read only
 * after tmpc_synchronize().  Instances whose variant id is >= the handle's number of problems get
 * status TMPC_STATUS_NUMERICAL and NaN outputs.
 */
int tmpc_solve_batch_device(tmpc_handle *h, int64_t B,
                            const double *x_k, const double *ref, const uint8_t *variant,
                            double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                            int32_t *status, int32_t *iters);

/*
 * Which kernel solves a variant.  TMPC_PATH_AUTO (default): the one-wave-per-QP kernel
 * (csrc/tmpc_kernels.hip) when one of its compiled shapes covers the condensed problem, otherwise
 * the workgroup-per-QP kernel (csrc/tmpc_block.hip: nv <= 128, any number of rows, G'DG on the
 * FP64 matrix cores).  TMPC_PATH_WAVE / TMPC_PATH_BLOCK force one of them (TMPC_E_UNSUPPORTED if
 * it cannot take the problem).  Results agree to the refinement's accuracy either way
 * (tests/test_hip_parity.py).  tmpc_get_kernel_path reports the path a variant currently takes.
 * Environment (developer knob, read by tmpc_create): TMPC_BLOCK_PAIRS=0 makes the workgroup-per-QP
 * kernel keep one row of G per constraint row instead of one per pair of mirrored rows
 * (DESIGN.md section 4); the answers agree (tests/test_block_layout.py).
 */
#define TMPC_PATH_AUTO  0
#define TMPC_PATH_WAVE  1
#define TMPC_PATH_BLOCK 2
int tmpc_set_kernel_path(tmpc_handle *h, int path);
int tmpc_get_kernel_path(const tmpc_handle *h, int variant);

/*
 * Name of the kernel instantiation that solves a variant on the current path, as it appears (up to the
 * anonymous-namespace qualifier) in a rocprofv3 kernel trace: "tmpc::solve_kernel<NV,DP,DS,KC,CP,CS,WPB>" (one wavefront
 * per QP; shape = padded variables, dense paired / single slots, factored width and paired / single slots, waves per
 * workgroup) or "tmpc::solve_block_kernel<T>" (one workgroup per QP).  The string belongs to the library.
 */
const char *tmpc_kernel_name(const tmpc_handle *h, int variant);

/*
 * Diagnostics (tests/wavesim: the kernel sources compiled for the CPU under sanitizers): writes to `path` everything the
 * wave-per-QP kernel receives for `variant` -- two int32 words {tag "TMPC" = 0x43504d54, dump format = tmpc::DUMP_FORMAT of
 * csrc/tmpc_device.hpp, bumped with every change of the records below: a reader built against another format must reject the
 * file}, the compiled shape (NVP, DP, DS, KC, CP, CS; int32 x 6), the size of tmpc::DeviceQP (csrc/tmpc_device.hpp; uint64)
 * and the structure itself, then, for each of its arrays in field order (Gt, Hct, Psi, Hs, Hinv, F1s, F2s, g0p, Esp, vmask,
 * row_of, gp0, Ep, Dv, Tzs, Txf, Mth, A, B, cip), a uint64 byte count and the bytes.  Host-only handles (device < 0) only:
 * TMPC_E_UNSUPPORTED otherwise, or when no wave shape covers the variant.  Not part of the solve path.
 */
int tmpc_debug_dump_layout(const tmpc_handle *h, int variant, const char *path);
/* The same for the workgroup-per-QP kernel: the two format words, tiles and workspace rows (int32 x 2), the sizes of
 * tmpc::DeviceQP and tmpc::BlockQP (uint64 x 2), the two structures (BlockQP as of format 2: ncp, nz4, zx0, znx, mir, ng, ngp, the
 * array pointers, row_start[9]), then 20 records, each a uint64 byte count and the bytes: Hs, Hinv, F1s, F2s, gp0, Ep, Dv, Tzs,
 * Txf, Mth, A, B of the first structure; Grm ([ngp + NVP][NVP]: the rows of G, then the NVP rows of Hs), Gcm, GHrm, g0, Es,
 * ncols of the second; Gw ([ncp][NVP], G by constraint row) -- zero bytes when Gw is Grm (mir == 0); and ci ([ncp], format 3). */
int tmpc_debug_dump_block_layout(const tmpc_handle *h, int variant, const char *path);
/* The same for the batched LP kernel (tmpc_lp_batch): the polytope (H, h) in kernel units -- int32 d, nr, nrp, DP (padded
 * dimension), max_iter; double tol, relax_by, hm; then H transposed [DP][nrp], h [nrp] and the row scale [nrp].  No device
 * is touched.  TMPC_E_INVALID when the batch would be decided on the host (no normal at all, or a row 0 <= h_r < 0). */
int tmpc_debug_dump_lp_layout(int32_t d, int32_t nr, const double *H, const double *h, double relax_by, const char *path);

/*
 * Device-resident closed loop over a lossy network for B independent trajectories and T time steps: the body
 * of the reference's Monte-Carlo loop (Results/results_linear_system.py:209-259,291; with extended != 0
 * results_linear_system_with_extendedMPC.py:247-378) -- controller packet, packet losses in both directions,
 * consistent actuator with nominal model and ancillary feedback (SmartActuator.py:125-231), plant update,
 * estimator / robust estimator (Estimator.py:9-161) -- run as per-trajectory state machines, one wavefront per trajectory,
 * on the handle's stream: INSIDE the solve kernel, between two solves of the trajectory, where the controller has one
 * QP (one launch for the whole sweep; tmpc_mc_set_fused below), else in ONE launch per time step next to the solve's.
 * Only the statistics return to the host.
 *
 *   in   p_loss B        loss probability of the trajectory (both directions)
 *        ref    T        position reference; the solve gets ref_t = [ref[t], 0, ...]   (:240)
 *        th_u   B*T      uniforms: controller->plant packet of step t is lost iff t > 0 and th_u < p_loss
 *        ga_u   B*T      same for the plant->controller packet
 *        w      B*T*nx   disturbance realisations
 *        x0     B*nx or NULL (zeros)
 *        HZ,hZ  rZ x nx, rZ   the tube cross-section Z for the membership check (:258); rZ = 0 skips it
 *   out  (any may be NULL)
 *        err2        B   sum_t (x_t[0]-ref_t)^2 + |x_t[1:]|^2   (tracking error of :291 = sqrt(err2)/T)
 *        tube_viol   B   steps with x_t - x_nom_t outside Z
 *        not_optimal B   solves with status != 0 (a solve with status >= 2 sends no packet)
 *        x_final     B*nx
 *        consistent  B   max |x_hat - x_nom| over the steps with Theta_t = gamma_t = 1 (0 by Proposition 1; not for extended)
 *        iters_sum   B   interior-point iterations spent on the trajectory (the solve effort the reference's scripts report
 *                        as times, results_linear_system.py:305-315)
 * All pointers are HOST pointers; the call returns when the results are in place.
 */
int tmpc_mc_run(tmpc_handle *h, int64_t B, int32_t T, int extended, const double *p_loss, const double *ref,
                const double *th_u, const double *ga_u, const double *w, const double *x0, const double *HZ, const double *hZ,
                int32_t rZ, double *err2, int32_t *tube_viol, int32_t *not_optimal, double *x_final, double *consistent,
                int32_t *iters_sum);

/*
 * The state machines of tmpc_mc_run driven by GIVEN controller packets instead of solved ones -- the test entry that pins the
 * device-side Estimator / RobustEstimator (Estimator.py:43-161) and SmartActuator / ConsistentActuator (SmartActuator.py:57-231)
 * directly against trajectories recorded from the reference's classes (tests/golden/glue_golden.npz, glue_smart_golden.npz).
 * Step t of trajectory b runs exactly the kernel tmpc_mc_run launches after its solve, with the packet
 * U_pkt[b][t] = [u_0 .. u_{N-1} | terminal column] ((N+1) x nu, row per column of the reference's U_t) and, for the extended
 * controller, x_nom_0 = xn0_pkt[b][t]; theta / gamma are the ARRIVAL flags of the two links (1 = arrives; step 0 always
 * arrives, results_linear_system.py:211-214), w the disturbances, x0 the initial state (NULL: zeros).  No QP is solved.
 *   out  trace_f  B*T*(3 nx + nu)   per step: x_{t+1}, x_hat_{t+1}, the nominal state in the plant's packet, u_t
 *        trace_i  B*T*3             per step: s_t, Theta_t, and q_t as the controller's packet of step t carried it
 * Actuator kind, plant and gains are the handle's (tmpc_mc_set_actuator, tmpc_mc_set_plant, K / K_anc of tmpc_problem).
 * All pointers are HOST pointers.  Added without an ABI bump: a new export, nothing else changed.
 */
int tmpc_mc_replay(tmpc_handle *h, int64_t B, int32_t T, int extended, const double *U_pkt, const double *xn0_pkt,
                   const uint8_t *theta, const uint8_t *gamma, const double *w, const double *x0, double *trace_f, int32_t *trace_i);

/*
 * Warm start inside tmpc_mc_run (off by default).  Consecutive QPs of a trajectory share most of their active set: with
 * on != 0 every solve first hands the working set certified by the trajectory's previous solve (of the same problem
 * variant) to the active-set refinement.  The refinement accepts a point only if it is primal feasible on ALL rows with
 * non-negative multipliers, i.e. only the exact minimiser; otherwise the solve falls back to the cold interior-point
 * start.  Results are therefore the same with and without (tests/test_closed_loop.py); only the iteration counts drop.
 * One-wave-per-QP kernel only; the workgroup-per-QP kernel ignores the setting.
 */
int tmpc_mc_set_warm_start(tmpc_handle *h, int on);

/*
 * How tmpc_mc_run steps its trajectories.  TMPC_MC_FUSED_ON: ONE launch for the whole sweep -- a wavefront keeps its
 * trajectory for all T time steps and alternates between the QP solve and the trajectory's state machines inside the
 * kernel (the reference's loop body, Results/results_linear_system.py:209-259, with nothing between two of its
 * iterations).  TMPC_MC_FUSED_OFF: per time step one solve launch (per problem variant) and one
 * launch of the state machines.  The two give the same numbers bit for bit: a trajectory's arithmetic does not depend on
 * which wavefront runs it or when.  TMPC_MC_FUSED_AUTO (default): fused when the trajectories fill their rounds on the
 * card's resident wavefronts to at least 85 % (or fit in one round), else per step -- a fused work item is T solves
 * long, a per-step one a single solve.  Fusing needs ONE problem on the one-wave-per-QP kernel: the extended controller
 * (two problems, chosen per step by the arrival flag) takes ONE launch per problem and time step with the state machines
 * of the problem's trajectories inside (two launches per step instead of three: _ON, and _AUTO from one round of resident
 * wavefronts on -- a smaller batch is bound by the latency of its launches, which the state machines inside lengthen);
 * the workgroup-per-QP kernel and tmpc_mc_replay always take a solve launch per problem and a state-machine launch per
 * step.  tmpc_mc_last_fused: what the last tmpc_mc_run of the handle did -- 1: one launch for the sweep; 2: one launch
 * per problem and step, state machines inside; 0: solve launches + a state-machine launch per step.
 */
#define TMPC_MC_FUSED_OFF 0
#define TMPC_MC_FUSED_ON 1
#define TMPC_MC_FUSED_AUTO 2
int tmpc_mc_set_fused(tmpc_handle *h, int mode);
int tmpc_mc_last_fused(const tmpc_handle *h);

/*
 * Per-solve computation times -- what the reference's controllers keep in _computational_times and the scripts print as
 * max / quantiles / median (TubeTrackingMPC.py:205, 242-243; results_linear_system.py:305-315).  On the device an MPC solve
 * is one instance of a batched launch; with tmpc_set_solve_timing(on) every instance records the time from the moment its
 * wavefront / workgroup picks it up to the moment its outputs are written, in ticks of the GPU's constant 100 MHz counter
 * (s_memrealtime: 1 tick = 10 ns).  tmpc_get_solve_ticks copies out the B tick counts of the last tmpc_solve_batch /
 * tmpc_solve_batch_device call (instances a variant selector skipped: 0); after a tmpc_mc_run, tmpc_mc_get_solve_ticks
 * gives per trajectory the sum and the maximum over its T solves (either pointer may be NULL).  Off by default.
 */
int tmpc_set_solve_timing(tmpc_handle *h, int on);
int tmpc_get_solve_ticks(tmpc_handle *h, int64_t B, int64_t *ticks);
int tmpc_mc_get_solve_ticks(tmpc_handle *h, int64_t B, int64_t *ticks_sum, int64_t *ticks_max);

/*
 * Sample trajectory of tmpc_mc_run -- what the scripts keep for their plots (x_traj, x_nom_traj of one run per loss rate,
 * results_linear_system.py:298-301).  tmpc_mc_set_capture(index >= 0) makes the following runs record trajectory `index`
 * (-1: off); after a run tmpc_mc_get_capture copies out, for t = 0 .. T-1, the plant state x_t, the nominal state the tube
 * check of step t uses, and the applied input u_t (row-major T x nx, T x nx, T x nu; any pointer may be NULL).  Dead
 * trajectories of the R-MPC loop leave zeros from their last step on.
 */
int tmpc_mc_set_capture(tmpc_handle *h, int64_t index);
int tmpc_mc_get_capture(tmpc_handle *h, int32_t T, double *x_traj, double *x_nom_traj, double *u_traj);

/*
 * Realisations drawn on the device (throughput runs: the host arrays of a 10 x 1000 x 250 sweep are 120 MB and their
 * generation takes longer than the closed loop).  With tmpc_mc_set_device_rng(on = 1, seed, first_trajectory, w_bound)
 * the following tmpc_mc_run calls ignore th_u, ga_u and w (NULL allowed) and draw, for trajectory b of the call and step t,
 * from Philox4x64-10 with key (seed, first_trajectory + b) and counter (t, j, 0, 0): block j = 0 gives the theta and gamma
 * uniforms and w_0, w_1, block j >= 1 gives w_{4j-2} .. w_{4j+1}; a uniform is (x >> 11) * 2^-53, a disturbance component
 * w_bound[i] * (2 u - 1) (the reference draws rng_w.uniform(-w_bound, w_bound), results_linear_system.py:229-233).
 * A trajectory's stream depends on (seed, its global index, t) only -- not on the batch it is solved in nor on the rank.
 * LinearMPCOverNetworks.montecarlo.draw_realisations_philox is the numpy twin (tests: identical closed loops).
 * w_bound: nx half-widths, NULL = no disturbance.  on = 0 returns to host arrays.
 */
int tmpc_mc_set_device_rng(tmpc_handle *h, int on, uint64_t seed, int64_t first_trajectory, const double *w_bound);

/*
 * Plant simulated by tmpc_mc_run.  TMPC_PLANT_LINEAR (default): x+ = A x + B u + w (results_linear_system.py:248).
 * TMPC_PLANT_CARTPOLE: the nonlinear cart-pole the linear model was derived from (results_linear_system.py:26-47;
 * the reference integrates it with PyBullet at 500 Hz, results_nonlinear_system.py:30-37), zero-order hold of the
 * input over the sampling period, classical RK4 with `substeps` steps; w is added to the result (pass zeros).
 * par = {M, m, b, I, g, l, Th}.  Needs nx = 4, nu = 1.
 */
#define TMPC_PLANT_LINEAR   0
#define TMPC_PLANT_CARTPOLE 1
int tmpc_mc_set_plant(tmpc_handle *h, int kind, const double *par7, int substeps);
/*
 * With a nonlinear plant tmpc_mc_run also sums |x - ref|^2 over the T * substeps physics steps (the state at the start of
 * every physics step, i.e. x_traj[:, 0:-1] of results_nonlinear_system.py:361, whose tracking error is taken at 500 Hz);
 * copied out per trajectory by tmpc_mc_get_physics_error (NaN for an R-MPC trajectory that stopped).
 */
int tmpc_mc_get_physics_error(tmpc_handle *h, int64_t B, double *err2_phys);

/*
 * Plant-side actuator simulated by tmpc_mc_run.  TMPC_ACTUATOR_CONSISTENT (default): ConsistentActuator with nominal
 * model and ancillary feedback (SmartActuator.py:125-231), the remote tube MPC's actuator.  TMPC_ACTUATOR_SMART: the plain
 * SmartActuator (SmartActuator.py:11-123) the reference pairs with the non-robust TrackingMPC (results_linear_system.py:
 * 198-205, 262-287): no nominal model, terminal law on the measured state, the plant packet carries the measured state.
 * With it a trajectory whose solve is infeasible stops (the reference sets track_feasible = False, :268-270): its err2
 * becomes NaN, its x_final the last state reached, and not_optimal counts the one failed solve.
 */
#define TMPC_ACTUATOR_CONSISTENT 0
#define TMPC_ACTUATOR_SMART      1
int tmpc_mc_set_actuator(tmpc_handle *h, int kind);

/* Block until everything enqueued on the handle's stream has finished. */
int tmpc_synchronize(tmpc_handle *h);

/*
 * Device time of the solve kernel(s) of the most recent tmpc_solve_batch[_device]
 * call, from HIP events recorded on the handle's stream around the launch
 * (synchronises the stream).  This is what bench.py reports as the kernel's
 * launch duration.
 */
int tmpc_last_kernel_ms(tmpc_handle *h, float *ms);

/*
 * Sum of the per-call device times (same HIP events as above) of all solve calls since the
 * last reset, and their count (at most 4096 calls are tracked between resets).
 * Synchronises the stream.  bench.py divides the two for the average launch duration.
 */
int tmpc_kernel_ms_total(tmpc_handle *h, float *total_ms, int32_t *launches, int reset);

/*
 * Introspection for DESIGN.md / bench.py's roofline accounting: dimensions of the
 * condensed QP of `variant` as the kernels see it.
 *   nv   decision variables after condensing
 *   nc   inequality rows kept (rows that cannot bind for any x_k are dropped)
 *   npar rows that depend on x_k only (checked once per instance, not iterated on)
 */
int tmpc_get_dims(const tmpc_handle *h, int variant, int32_t *nv, int32_t *nc, int32_t *npar);
/*
 * How the nc rows are stored: nd general ("dense") rows of width nv and one block of ncc rows of rank kc kept in factored
 * form Hc * Psi (the terminal set acts on [x_N; theta] only, TubeTrackingMPC.py:149; the initial-state set of the
 * packet-received problem on x_0 only, :278); nc = nd + ncc, ncc = kc = 0 when nothing is factored.  bench.py prices the
 * factored block at its rank in `roofline_factored`.  Added without an ABI bump: a new export, nothing else changed.
 */
int tmpc_get_factoring(const tmpc_handle *h, int variant, int32_t *nd, int32_t *ncc, int32_t *kc);

/*
 * Copies the condensed, unscaled QP data of `variant` to caller buffers (any may be
 * NULL): the QP is  min 1/2 z'Hz + (F1 x_k + F2 ref)'z  s.t.  G z <= g0 + E x_k.
 *   H nv*nv, F1 nv*nx, F2 nv*nx, G nc*nv, g0 nc, E nc*nx.   Used by the tests.
 */
int tmpc_get_condensed(const tmpc_handle *h, int variant,
                       double *H, double *F1, double *F2, double *G, double *g0, double *E);

/*
 * Offline stage: a batch of support-function linear programs over ONE polytope,
 *
 *        val[b] = max  C[b,:] . x   s.t.  H x <= h   (row relax[b] of h raised by relax_by)
 *
 * Replaces the one-at-a-time scipy.optimize.linprog calls of the reference's set
 * computations (reference src/LinearMPCOverNetworks/utils_polytope.py:12-23 `support`,
 * :19 the linprog call; used by the Gilbert-Tan recursion :247-268, the Pontryagin
 * difference :25-38, and the redundancy removal of polytope.reduce,
 * TubeRegulatorMPC.py:74).  One wavefront per LP: interior-point iterations handed over
 * to primal active-set steps, so that the value is the vertex value (csrc/tmpc_lp.hip).
 *
 *   d       dimension, 1 <= d <= 32           nr  rows of H (row-major nr x d), nr >= 1
 *   B       number of objectives              C   B x d, row-major
 *   relax   B row indices or NULL; relax[b] = -1 leaves h alone.  Row relax[b] is raised
 *           by relax_by IN THE UNITS OF h AS PASSED (the redundancy test of row i is
 *           "maximise H[i,:] x with h[i] + 1", polytope.reduce)
 *   val     B        x   B x d maximiser or NULL
 *   status  B  TMPC_STATUS_* (OPTIMAL: x feasible to 1e-11 max(|h_r|, 1), with multipliers
 *           y >= -1e-10 max(y) on active rows and |c - H'y|_inf <= 1e-11 |c| -- a point of
 *           the optimal face, or the interior-point iterate at gap 1e-12 and the same dual
 *           residual; MAX_ITER: last iterate without that certificate, accurate to about
 *           1e-8; INFEASIBLE; UNBOUNDED: val = +inf)
 *   iters   B  interior-point iterations
 * All pointers are HOST pointers (this is a set-up step; the data is small).  Errors:
 * negative TMPC_E_* code, text through tmpc_last_error(NULL).
 */
int tmpc_lp_batch(int device, int32_t d, int32_t nr, const double *H, const double *h,
                  int64_t B, const double *C, const int32_t *relax, double relax_by,
                  double *val, double *x, int32_t *status, int32_t *iters);

#ifdef __cplusplus
}
#endif
#endif /* TMPC_H */
