// Device-side view of one condensed, scaled QP variant and the launch interface
// between tmpc_api.cpp (host) and tmpc_kernels.hip (device).
#pragma once
#ifdef TMPC_HOST_SIM
#include "hip_sim.hpp"
#else
#include <hip/hip_runtime.h>
#endif

#include <cstddef>
#include <cstdint>

#include "../../include/tmpc.h"

namespace tmpc {

// First two words of the files tmpc_debug_dump_layout / _block_layout write (include/tmpc.h): a tag and the version of the
// record layout below -- bumped whenever DeviceQP / BlockQP or the list of dumped arrays changes, so that a reader built
// against another layout (tests/wavesim) rejects the file instead of mis-parsing it.
constexpr int32_t DUMP_TAG = 0x43504d54;       // "TMPC"
constexpr int32_t DUMP_FORMAT = 3;             // 2: BlockQP with mir / ng / ngp, Grm = [ngp + NVP][NVP], 19th record Gw; 3: cip / ci (records 20 / 20)

// Working set handed from one solve to the next (closed loop): per instance WS_STRIDE ints = [m, row ids ...]; a row id is
// (row side) * 64 + lane in the wave kernel's slot layout.  m = 0: nothing to start from.
constexpr int WS_CAP = 36;       // largest working set the refinement handles (24 for the shapes with NV <= 24)
constexpr int WS_STRIDE = 40;

// All pointers are device pointers owned by the handle.  The wave kernel's instantiation for a variant is
// (NVP, DP, DS, KCP, CP, CS): padded variable count; 64-functional slots of dense paired / dense single rows; padded width
// of the factored block and its paired / single slots (KCP = CP = CS = 0: everything dense).  See tmpc_kernels.hip.
struct DeviceQP {
    int nx, nu, N;
    int nv, nc, npar, nth;
    int nd, ncc, kc;      // dense rows, factored rows (nc = nd + ncc), factor width
    int nks;              // k-steps of the MFMA pass over the dense functionals (4 functionals each)
    int off_theta, off_x0, off_aux;
    int max_iter, always_infeasible;
    double tol;
    const double *Gt;     // [(DP+DS)*64][LDG]   dense functionals (paired slots first), scaled, row-major, zero padded; LDG = 16 ceil((NVP+1)/16) + 1
    const double *Hct;    // [KCP][(CP+CS)*64]   factored functionals' left factor, transposed, zero padded
    const double *Psi;    // [KCP][NVP]          factored rows' right factor
    const double *Hs;     // [NVP][NVP]          scaled Hessian, identity on the padding
    const double *Hinv;   // [NVP][NVP]
    const double *F1s;    // [nv][nx]
    const double *F2s;    // [nv][nx]
    const double *g0p;    // [RS*64]       right-hand side offsets per row side, slot layout (padding: 1); RS = 2 DP + DS + 2 CP + CS
    const double *Esp;    // [nx][RS*64]   right-hand side dependence on x_k, slot layout, one plane per state (coalesced per lane)
    const double *cip;    // [RS*64]       1 / (g Hs^-1 g') per row side, slot layout (padding: 0): the multiplier of the QP with that
                          //               row alone is its violation times this (starting point of the interior-point phase)
    const uint32_t *vmask;   // [64]       bit i of entry l: row side i of lane l is a real row
    const int32_t *row_of;   // [RS*64]    row (order of Condensed::Gs) behind each row side, -1: padding
    const double *gp0;    // [npar]
    const double *Ep;     // [npar][nx]
    const double *Dv;     // [nv]
    const double *Tzs;    // [nvf][nv]  z_full = Tzs zs + Txf x_k (Dv folded in); NULL: z_full = Dv .* zs  (tmpc_condense.hpp: Tz, Tx)
    const double *Txf;    // [nvf][nx]
    int nvf;              // length of z_full = [u | theta | x_0 | aux]; the outputs are read from it at off_theta / off_x0
    const double *Mth;    // [nx+nu][nth]
    const double *A;      // [nx][nx]
    const double *B;      // [nx][nu]
    long long *dbg;       // diagnostic builds only (TMPC_STAMPS); nullptr otherwise
    float *save;          // [resident waves][2][RS][64] or nullptr: (s, lambda) at the hand-over to the refinement, so that a
                          // refinement that fails to certify its set continues the interior-point phase instead of repeating it
    long long *ticks;     // [B] or nullptr: time the instance spent in its wave / workgroup, in s_memrealtime ticks (10 ns)
};

// Block path (tmpc_block.hip): all rows dense, nv padded to 16 * tiles.
//
// Two layouts.  mir == 0: a row of G per constraint row, in the order of Condensed::Gs (initial-state rows first, the general
// rows by the column tiles they reach).  mir != 0 (every row has its mirror row, Condensed::mirror: box-type sets and sets
// derived from them, which is what the reference builds): a row of G per FUNCTIONAL g; the per-row arrays (g0, Es, the
// kernel's workspace) hold its upper side g'z <= h at index f and its lower side -g'z <= h at index f + mir.  The three
// G-sized passes of an iteration (G'DG on the matrix cores, the G'v products, the row products G v) then read half the rows:
// the functional's weight is d_f + d_{f+mir}, its G'v weight v_f - v_{f+mir}, and G v serves both sides with one product.
struct BlockQP {
    int ncp;              // length of the per-row arrays (multiple of 64): rows padded (mir == 0), or 2 * ngp
    int nz4, zx0, znx;    // rows [0, nz4) act on columns [zx0, zx0 + znx) only (initial-state rows; nz4 = 0: no such block)
    int mir;              // 0, or ngp: offset of a functional's lower side in the per-row arrays
    int ng, ngp;          // rows of G (nc, or nc / 2 functionals) and their padded count (multiple of 64)
    const double *Grm;    // [ngp + NVP][NVP]  scaled G, row-major, zero padded (MFMA operands, G'v passes), then the rows of Hs (gt_products)
    const double *Gcm;    // [NVP][ngp]  the same, column-major (thread-per-row products)
    const double *Gw;     // [ncp][NVP]  G by constraint row (row f + mir = -row f): the refinement gathers its working rows here; mir == 0: Grm
    const double *GHrm;   // [ncp][NVP]  G * Hs^-1 by constraint row, row-major (refinement: S = G_W Hs^-1 G_W')
    const double *g0;     // [ncp]       right-hand side offsets (padding rows: 1)
    const double *Es;     // [ncp][nx]   right-hand side dependence on x_k
    const double *ci;     // [ncp]       1 / (g Hs^-1 g') per constraint row (padding rows: 0), see DeviceQP::cip
    const int32_t *ncols; // [ngp]       columns a row of G reaches (its zeros beyond are skipped); rows >= nz4 are ordered by it
    int row_start[9];     // row_start[t]: first row of G that reaches the 16-column tile t (row_start[0] = nz4; ng if none)
};

// What solve_block_kernel reads of the model: one record in device memory per variant (written once by tmpc_create), handed
// to the kernel as ONE pointer.  By value the two structures were sixty kernel-argument pointers that stayed live through the
// whole kernel: they overflowed the scalar registers into vector lanes and from there into scratch (round 2: 89 spilled VGPRs,
// 480 B private segment).  Read through the constant address space, a field costs a scalar load where it is used.
struct BlockArgs {
    DeviceQP qp;          // (qp.ticks / qp.dbg of this copy are not used: they change per launch and travel as kernel arguments)
    BlockQP bq;
};

struct KernelShape {
    int nvp = 0, dp = 0, ds = 0, kcp = 0, cp = 0, cs = 0;
};

// Chooses the cheapest compiled shape that covers a variant with nd2 / nd1 dense paired / single functionals and nc2 / nc1
// factored ones of width kc; false if none does.
bool pick_config(int nv, int nd2, int nd1, int kc, int nc2, int nc1, KernelShape *shape);
size_t lds_bytes(const KernelShape &shape, int grows);      // grows = 4 * nks rows of dense functionals staged in LDS
const char *kernel_name(const KernelShape &shape);
bool parks_in_lds(const KernelShape &shape);               // the hand-over iterate stays in LDS: no DeviceQP::save slot needed

// Work counters of the persistent grids (wave kernel and block kernel): every launch draws its instances from a fresh, zeroed device word of
// a ring that is cleared in one piece when it has gone round -- no reset and no extra stream operation per launch.
// Launches that share a ring must be ordered on one stream.
struct WorkCounter {
    unsigned long long *ring = nullptr;
    int size = 0, pos = 0;
};

// ws_in / ws_out: optional working sets (WS_STRIDE ints per instance), see above
hipError_t launch_solve(const DeviceQP &qp, const KernelShape &shape, int variant_id, int64_t B,
                        const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                        double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, const int32_t *ws_in, int32_t *ws_out,
                        WorkCounter *wc, int n_cu, hipStream_t stream);

// Block path: tiles = NVP / 16 in {1, 2, 4, 8} (0: nv > 128, unsupported); workspace = blocks * rows * ncp doubles
int block_tiles(int nv);
int block_workspace_rows();
size_t block_lds_bytes(int tiles);
int block_occupancy(int tiles);
// dargs: the device copy of {qp, bq} (unused on the host execution model of tests/wavesim, which reads qp / bq directly)
hipError_t launch_block(const DeviceQP &qp, const BlockQP &bq, const BlockArgs *dargs, int tiles, double *ws, int ws_blocks, int variant_id, int64_t B,
                        const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                        double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, WorkCounter *wc, hipStream_t stream);

// Device-resident closed loop (tmpc_mc.hip)
struct McModel {
    int nx, nu, N, extended, rZ;
    int plant, substeps;                 // TMPC_PLANT_*, RK4 steps per sampling period
    int smart;                           // TMPC_ACTUATOR_SMART: no nominal model, infeasible trajectories stop
    double par[7];                       // cart-pole: M, m, b, I, g, l, Th
    const double *A, *B, *K, *K_anc;     // model and gains (row-major)
    const double *HZ, *hZ;               // tube cross-section Z for the membership check
};
struct McState {                         // all [trajectory]-major device arrays
    double *x, *x_hat, *x_nom;           // plant state, controller-side estimate, plant-side nominal state   [B][nx]
    double *Ubuf;                        // sequence buffered by the actuator                                  [B][N+1][nu]
    double *u_latest0, *x_nom0_latest;   // first input / x_nom_0 of the last sequence sent                    [B][nu], [B][nx]
    double *ref_k;                       // reference handed to the solve                                       [B][nx]
    double *err2, *consistent;           // statistics                                                          [B]
    double *err2_phys;                   // sum of |x - ref|^2 over the physics steps of a nonlinear plant (or nullptr)  [B]
    int32_t *q_est, *q_act, *s, *Theta, *last_lost, *tube_viol, *not_optimal, *iters_sum;
    uint8_t *gamma;                      // arrival of the previous plant packet = variant of the next solve   [B]
    uint8_t *dead;                       // trajectory stopped after an infeasible solve (smart actuator only)      [B]
    const double *p_loss, *th_u, *ga_u, *w;   // realisations: [B], [B][T], [B][T], [B][T][nx]
    // realisations drawn on the device instead (tmpc_mc_set_device_rng; th_u, ga_u, w are then not read): Philox4x64-10,
    // key = (seed, first trajectory + b), counter = (t, block, 0, 0); block 0 = [theta, gamma, w_0, w_1], block j = w_{4j-2 ..}
    int rng_on;
    unsigned long long rng_seed;
    long long rng_first;
    const double *w_bound;                    // [nx] half-widths of the disturbance box
    const long long *ticks;                   // per-solve durations of the step just solved, or nullptr                [B]
    long long *tick_sum, *tick_max;           // their sum and maximum along the trajectory (with ticks)                 [B]
    long long cap_index;                      // trajectory whose states are recorded (-1: none)
    double *cap;                              // [T][2 nx + nu]: x_t, the nominal state the tube check uses, u_t
    // packet injection (tmpc_mc_replay): the controller's packets come from the caller instead of from a solve, and every
    // step of every trajectory is recorded
    const double *rp_U;                       // [B][T][N+1][nu] packets U_t (terminal column included), or nullptr: solved packets
    const double *rp_xn0;                     // [B][T][nx]      x_nom_0 of the packets (extended controller)
    double *trace_f;                          // [B][T][3 nx + nu]: x_{t+1}, x_hat_{t+1}, nominal state of the plant's packet, u_t; or nullptr
    int32_t *trace_i;                         // [B][T][3]: s_t, Theta_t, q_t (the controller's packet)
};
hipError_t launch_mark_invalid_variants(const uint8_t *variant, int nvariants, int64_t B, int nx, int nu, int N, double *u_nom,
                                        double *x_nom0, double *xu_ss, double *x_nom, int32_t *status, int32_t *iters,
                                        hipStream_t stream);
// one launch before the first solve (reference of step 0), then one launch per time step after the solve launch(es)
hipError_t launch_mc_pre(const McModel &m, const McState &st, int64_t B, double ref_0, hipStream_t stream);
hipError_t launch_mc_step(const McModel &m, const McState &st, int t, int T, int64_t B, double ref_t, double ref_next, const double *u_nom,
                          const double *x_nom0, const double *xu_ss, const int32_t *status, const int32_t *iters, hipStream_t stream);

// Fused closed loop (tmpc_fused.hip: the wave kernel with the state machines inside, a trajectory per work item): available for
// every wave shape; one problem variant (the plain controllers: the extended one changes its problem from step to step)
struct McFused {
    McModel m;
    McState st;
    int T;
    const double *ref_seq;               // [T] reference of every time step (device)
};
hipError_t launch_solve_mc(const DeviceQP &qp, const KernelShape &shape, int64_t B, double *u_nom, double *x_nom0, double *xu_ss,
                           int32_t *status, int32_t *iters, int32_t *ws, const McFused *mc /* device */, WorkCounter *wc, int n_cu, hipStream_t stream);
// one time step of one problem of the extended controller, the state machines of its trajectories inside (tmpc_fused_step.hip):
// `variant` is the selector this step reads, `gamma_out` the one it writes for the next step (two buffers, swapped by the caller)
hipError_t launch_solve_mc_step(const DeviceQP &qp, const KernelShape &shape, int variant_id, int64_t B, const uint8_t *variant, double *u_nom,
                                double *x_nom0, double *xu_ss, int32_t *status, int32_t *iters, int32_t *ws, const McFused *mc /* device */, int t,
                                uint8_t *gamma_out, WorkCounter *wc, int n_cu, hipStream_t stream);
// resident waves of the persistent grid of a shape (work items in flight): the host's choice between one fused launch and T launches
int resident_waves(const KernelShape &shape, int n_cu);

// LP kernel (tmpc_lp.hip): rows scaled to unit norm, h scaled by hm so that max |h| = 1
struct LpDevice {
    int d, nr, nrp;       // dimension, rows, row stride (multiple of 64)
    int max_iter;
    double tol, relax_by, hm;
    const double *Ht;     // [DP][nrp]  H transposed, zero padded (DP = lp_padded_dim(d))
    const double *h;      // [nrp]      padding rows: 1
    const double *rscale; // [nrp]      1 / (|H_r| hm): caller units of h -> kernel units
    unsigned long long *next_item;   // work counter of the launch, zero at its start (LPs beyond every wave's first)
};
int lp_padded_dim(int d);
int lp_waves_per_block();
int lp_workspace_arrays();
hipError_t launch_lp(const LpDevice &p, int64_t B, int nblocks, const double *C, const int32_t *relax, double *ws, double *val,
                     double *xout, int32_t *status, int32_t *iters, hipStream_t stream);

}  // namespace tmpc
