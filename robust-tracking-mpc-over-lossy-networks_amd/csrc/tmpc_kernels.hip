// HIP kernels for gfx950 (MI355X): batched tube-tracking QP solve, one wavefront (64 lanes) per QP instance.
//
// What one wave does for its instance (x_k, ref) -- the device side of solve_optimization_problem (reference
// TubeTrackingMPC.py:170-194 and :307-349, whose arithmetic the reference delegates to cvxpy/Clarabel at :183):
//
//   1. q = F1s x_k + F2s ref,  h = g0s + Es x_k      (the open-loop prediction x_i = A^i x_k + sum A^j B u_j is folded into
//      F1s / Es by tmpc_condense.cpp)
//   2. z = -Hs^-1 q; if G z <= h the unconstrained minimiser is the answer
//   3. Mehrotra predictor-corrector interior-point iterations on  min 1/2 z'Hs z + q'z  s.t.  G z + s = h, s >= 0
//   4. active-set refinement on W = {lambda_i > s_i}: proximal Newton steps on the KKT system of the equality-constrained
//      QP, accepted only when primal feasible on ALL rows with non-negative multipliers (the returned point is the exact
//      minimiser; the interior-point phase only has to identify the active set)
//   5. outputs: u_nom, x_nom[0], (x_bar, u_bar) = Mth theta, optionally x_nom
//
// Layout of the rows.  The constraint sets of the reference are boxes and sets derived from boxes (TubeTrackingMPC.py:48-50
// hard-codes box row counts), so almost every row  g'z <= h+  has a mirror row  -g'z <= h-.  tmpc_api.cpp pairs them:
// a FUNCTIONAL g with one or two SIDES.  A lane owns the functional (slot k, lane): its row of G is loaded once per sweep
// and feeds both sides -- half the LDS traffic, half the dot products and half the G'DG multiply-adds of a row-per-lane
// layout, while (s, lambda, r_p) stay per side.  Four kinds of 64-functional slots, all known at compile time:
//   dense paired (DP), dense single (DS), factored paired (CP), factored single (CS)
// "factored": one block of rows of rank KC << NV is kept as Hc * Psi (the terminal set acts on [x_N; theta] only,
// TubeTrackingMPC.py:149; the initial-state set of the packet-received problem acts on x_0 only, :278), which cuts its
// share of every sweep by NV / KC and of G'DG by (NV / KC)^2.
// Padding functionals carry g = 0, h = 1, s = 1, lambda = 0 and need no masks except where noted.
//
// Registers: (s, lambda, r_p, 1/s) per side stay in VGPRs for the whole interior-point phase; everything wave-uniform
// (z, the directions, their Psi-coordinates) lives in LDS and is read with broadcast loads; the model (G', Hc', Psi, Hs,
// Hs^-1) is staged once per workgroup in LDS.  The normal matrix M = Hs + G'DG is accumulated per lane (a few columns of
// its lower triangle at a time) and summed across the wave through an LDS transposition; the NV x NV solve is
// row-distributed (lane i holds row i, v_readlane broadcasts the pivot row).  The refinement starts only after (s, lambda)
// have been reduced to the working set, so the two phases never hold registers at the same time.
//
// Numerics are float64 throughout: cond(Hs) ~ 3e5 after scaling and the weights span 1e-1 .. 5e6 (R vs 10 P).
#ifdef TMPC_HOST_SIM
#include "hip_sim.hpp"      // tests/wavesim: this very source compiled for the CPU under sanitizers (never in the product)
#else
#include <hip/hip_runtime.h>
#endif

#include <cmath>
#include <mutex>
#include <cstdint>
#include <string>
#include <type_traits>
#include <utility>

#include "tmpc_device.hpp"
#include "tmpc_wave.hpp"
#include "tmpc_mc_step.hpp"

namespace tmpc {

namespace {

using wv::WAVE;
using wv::dpp_mov_d;
using wv::fast_rcp;
using wv::readlane_d;
using wv::lanes_backsub_lane;
using wv::lanes_factor;
using wv::lanes_forward;
using wv::vmax;
using wv::vmax_abs;
using wv::vmin;
using wv::OpMax;
using wv::OpMin;
using wv::OpSum;

#ifdef TMPC_HOST_SIM
unsigned long sim_rendezvous_total = 0;
#endif
constexpr int RED_STRIDE = 68;      // 64 lanes + a pad after every 16: conflict-free transposed reads
constexpr int ZERO_ROWS = 4;        // zero rows behind the staged dense functionals: row `grows` stands for every functional beyond them,
                                    // and the look-ahead of the MFMA pass ends one k-step (four rows) past the last one it multiplies

#if defined(TMPC_HOST_SIM)
#define STAMP(p) do { } while (0)
#elif defined(TMPC_STAMPS)
#define STAMP(p) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long now_ = __builtin_amdgcn_s_memtime(); \
                      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tph[p] += now_ - tlast; tlast = now_; } while (0)
#else
#define STAMP(p) do { asm volatile("; MARK " #p); } while (0)
#endif

#ifndef TMPC_FENCE_ALL
#define TMPC_FENCE_ALL 0      // diagnostic builds: 1 keeps the barrier in the one-wave-per-SIMD shapes as well
#endif
// Compiler-only barrier between two slots of a sweep: without it the loads of ALL slots are hoisted to the top of the
// unrolled loop and the kernel spills to scratch.
#ifdef TMPC_HOST_SIM
__device__ __forceinline__ void row_fence() {}        // (scheduling only: no LDS hand-over depends on it)
__device__ __forceinline__ void wave_lds_fence() { sim::wave_fence(); }
#else
__device__ __forceinline__ void row_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// Orders one wave's LDS traffic for the compiler (the hardware runs the DS instructions of a wave in issue order).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }
#endif

// 1/x to ~2^-27 for x > 0: v_rcp_f64 (good to about 2^-14) + ONE Newton step.  Used where the reciprocal only ranks
// step-length ratios against a fraction-to-the-boundary margin of at least 1e-4; the slacks' reciprocals, which enter the
// Newton system, take fast_rcp (two steps, full precision).
__device__ __forceinline__ double rcp1(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double wave_sum(double v) { return wv::wave_reduce<OpSum>(v); }
__device__ __forceinline__ double wave_min(double v) { return wv::wave_reduce<OpMin>(v); }
__device__ __forceinline__ double wave_max(double v) { return wv::wave_reduce<OpMax>(v); }

// An opaque copy of a per-lane value.  Everything derived from the lane id -- row addresses, the masks of `lane == k` and
// `lane < n` compares, validity bits -- is loop invariant, and the optimiser hoists all of it out of the instance loop,
// where it then has to survive the whole solve: hundreds of SGPR pairs parked in VGPR lanes and 64-bit addresses in
// scratch.  Re-deriving them from a fresh copy at every phase boundary costs an instruction each and keeps them local.
__device__ __forceinline__ int fresh(int v) {
#ifdef TMPC_HOST_SIM
    asm volatile("" : "+r"(v));
#else
    asm volatile("" : "+v"(v));
#endif
    return v;
}

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// Sum each of acc[0..CNT) over the 64 lanes and leave the totals in out[0..CNT) (LDS).  Round: RR (<= 16) entries are
// written as rows of a [RR][RED_STRIDE] LDS tile (lane l at column l + l/16); lane l then adds the 16-lane quarter (l & 3)
// of entry (l >> 2) -- conflict free for RED_STRIDE = 68 -- and the four quarters, which sit in one quad, meet through two
// DPP quad permutes.
template <int CNT, int RR>
__device__ __forceinline__ void wave_reduce_to_lds(const double (&acc)[CNT], double *red, double *out, int lane) {
    const int e = lane >> 2, qd = lane & 3;
    const int er = (RR < 16 && e >= RR) ? 0 : e;
    const int wcol = lane + (lane >> 4);
#pragma unroll
    for (int c0 = 0; c0 < CNT; c0 += RR) {
#pragma unroll
        for (int k = 0; k < RR; ++k)
            if (c0 + k < CNT) red[k * RED_STRIDE + wcol] = acc[c0 + k];
        wave_lds_fence();
        // (four partial sums: a chain of four dependent additions instead of sixteen)
        double t4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 16; ++j) t4[j & 3] += red[er * RED_STRIDE + qd * 17 + j];
        double t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
        t += dpp_mov_d<0xB1>(t);
        t += dpp_mov_d<0x4E>(t);
        if (qd == 0 && e < RR && c0 + e < CNT) out[c0 + e] = t;
        wave_lds_fence();
    }
}

// packed lower triangle, column-major: (i,j), i >= j, at col_off(j) + i - j
template <int NV>
__host__ __device__ constexpr int col_off(int j) { return j * NV - j * (j - 1) / 2; }

// entries per round of the transposition tile: 12 where the LDS is short (two waves per SIMD; the NV = 28 shape), else 16
constexpr int tile_rows(int nv, int wpb) { return (wpb == 8 || nv > 24) ? 12 : 16; }

// compile-time description of one kernel instantiation
template <int NV_, int DP_, int DS_, int KC_, int CP_, int CS_, int RR_>
struct Shape {
    static constexpr int NV = NV_, DP = DP_, DS = DS_, KC = KC_, CP = CP_, CS = CS_, RR = RR_;
    static constexpr int KCA = KC_ > 0 ? KC_ : 1;                 // array extents must be positive
    static constexpr int FD = DP_ + DS_, FC = CP_ + CS_;          // functional slots per lane: dense, factored
    static constexpr int NDP = FD * WAVE, NCCP = FC * WAVE;
    static constexpr int RS = 2 * DP_ + DS_ + 2 * CP_ + CS_;      // row sides per lane
    static constexpr int NT = NV_ * (NV_ + 1) / 2, KT = KC_ * (KC_ + 1) / 2;
    static constexpr int WCAP = NV_ <= 24 ? 24 : 28;              // max rows in the refinement's working set (<= WS_CAP)
    // Row strides of the small LDS matrices that are read with the row on the lane (Hs, Hs^-1, the normal matrix, the expanded
    // working rows, T = Hs^-1 G_W'): ODD, so that the lanes' rows start in different banks (an even stride of 24 puts six rows
    // on a bank, 28 four).  The variable count of a shape is the problem's own where that pays (NV = 11, 22, 26 for the
    // cart-pole at N = 10 and N = 20 / 21): a round-3 shape of 12 / 24 / 28 variables did the work of the padding columns
    // in every NV-long loop and NV^2 / 2 of it in the eliminations, and the 28-wide one had no LDS left for odd strides.
    static constexpr int odd_up(int v) { return v | 1; }
    static constexpr int LDH = odd_up(NV_);                       // Hs, Hs^-1
    static constexpr int LDM = odd_up(NV_ + 1);                   // normal matrix / its factor: NV entries and 1 / d_i per row
    static constexpr int LDW = odd_up(NV_), LDT = odd_up(WCAP);   // G_W, T
    // dense functionals, row-major in LDS: [FD * 64][LDG]; 16-column blocks of the MFMA tiling cover the NV columns of G
    // plus one more row of the product (row NV of A carries t: see sweep_a_dense); the odd stride keeps both the
    // lane-per-row reads of the sweeps and the 4 x 16 operand reads of the MFMA loop conflict free
    static constexpr int NB = (NV_ + 1 + 15) / 16;
    static constexpr int LDG = 16 * NB + 1;
    static constexpr int NTL = NB * (NB + 1) / 2;                 // 16 x 16 tiles of the lower triangle
    // first row side and number of sides of dense functional slot kd / factored functional slot kc
    static constexpr int dbase(int kd) { return kd < DP_ ? 2 * kd : 2 * DP_ + (kd - DP_); }
    static constexpr int dsides(int kd) { return kd < DP_ ? 2 : 1; }
    static constexpr int cbase(int kc) { return 2 * DP_ + DS_ + (kc < CP_ ? 2 * kc : 2 * CP_ + (kc - CP_)); }
    static constexpr int csides(int kc) { return kc < CP_ ? 2 : 1; }
};

// runtime counterpart for the refinement: row side i -> (dense?, functional slot, sign)
template <class SH>
__device__ __forceinline__ void side_info(int i, bool &dense, int &fslot, double &sgn) {
    if (i < 2 * SH::DP) { dense = true; fslot = i >> 1; sgn = (i & 1) ? -1.0 : 1.0; return; }
    if (i < 2 * SH::DP + SH::DS) { dense = true; fslot = SH::DP + (i - 2 * SH::DP); sgn = 1.0; return; }
    const int j = i - (2 * SH::DP + SH::DS);
    dense = false;
    if (j < 2 * SH::CP) { fslot = j >> 1; sgn = (j & 1) ? -1.0 : 1.0; return; }
    fslot = SH::CP + (j - 2 * SH::CP);
    sgn = 1.0;
}

// per-wave LDS workspace (doubles), see solve_kernel
template <class SH>
struct WaveLds {
    static constexpr int RED = SH::RR * RED_STRIDE;                                     // transposition tile
    static constexpr int POL = SH::WCAP * SH::LDW + SH::NV * SH::LDT + 4 * SH::WCAP;              // G_W, T, y, dy, W(idx)
    static constexpr int MFAC = SH::NV * SH::LDM;                                       // factor of the normal matrix between the two solves (row i at i LDM, then 1 / d_i)
    static constexpr int BIG = RED + MFAC > POL ? RED + MFAC : POL;                     // tile + factor (interior point) and the refinement's workspace are never live together
    static constexpr int SUMS = 2 * SH::NV + 8;                                         // two NV-vectors of G' products
    static constexpr int CSUMS = SH::KT + 2 * SH::KC + 8;                               // factored-block totals
    static constexpr int PMAT = SH::KCA * SH::NV;                                       // W * Psi
    static constexpr int HROW = SH::RS * WAVE;                                          // right-hand side h, [side][lane]
    static constexpr int VEC = 9 * SH::NV + 32;                                         // q, z, cost gradient, x_k, ref, scratch, dz_aff, dz, Psi-coordinates
    static constexpr int TOTAL = BIG + SUMS + CSUMS + PMAT + HROW + VEC;
    // The iterate of the hand-over is parked where the refinement leaves the workspace idle: behind its own arrays in BIG and
    // in the totals of the sweeps, which follow BIG directly.  One 32-bit word per row side and lane holds lambda in single
    // precision (the dual residual of the continuation is that of the iterate to 6e-8), then mu.  The slacks are not
    // parked: h - G z is the slack up to the primal residual of the hand-over, so a row well above that takes it (r_p = 0
    // there) and a row on its bound takes mu / lambda (solve_kernel has the rule).  Shapes without that much idle LDS
    // park (s, lambda) in HBM (DeviceQP::save, single precision).
    static constexpr int PARK_WORDS = SH::RS * WAVE + 2;
    static constexpr int IDLE_WORDS = 2 * ((BIG - POL) + SUMS + CSUMS + PMAT);
#ifdef TMPC_PARK_HBM
    static constexpr bool PARK_LDS = false;      // diagnostic builds: every shape parks in HBM
#else
    static constexpr bool PARK_LDS = IDLE_WORDS >= PARK_WORDS;
#endif
    static_assert(RED >= 2 * SH::NDP, "(D, t) of the dense functionals share the transposition tile");
};

// out = Psi vec for vectors kept in LDS: lane a < KC forms entry a (broadcast reads of vec)
template <class SH>
__device__ __forceinline__ void coords_lds(const double *Psi, const double *vec, double *out, int lane) {
    if constexpr (SH::KC > 0) {
        const int a_ = lane < SH::KC ? lane : 0;
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < SH::NV; ++j) t += Psi[a_ * SH::NV + j] * vec[j];
        if (lane < SH::KC) out[lane] = t;
    }
}

// g . v for this lane's dense functional of slot KD / factored functional of slot KCS; v (and c = Psi v) in LDS
template <class SH, int KD>
__device__ __forceinline__ double dense_dot(const double *Gt, int grows, const double *v, int lane) {
    const double *g = Gt + min(lane + KD * WAVE, grows) * SH::LDG;      // functionals beyond the staged rows read the zero row
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int j = 0; j + 1 < SH::NV; j += 2) { t0 += g[j] * v[j]; t1 += g[j + 1] * v[j + 1]; }
    if (SH::NV & 1) t0 += g[SH::NV - 1] * v[SH::NV - 1];
    return t0 + t1;
}
template <class SH, int KCS>
__device__ __forceinline__ double fact_dot(const double *Hct, const double *c, int lane) {
    const int r = lane + KCS * WAVE;
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < SH::KC; ++a) t += Hct[a * SH::NCCP + r] * c[a];
    return t;
}

// ---- sweep A, dense functionals: 1/s and the weights of both sides, the gap, |r_p|_inf; then
//          [ G'DG      ]     [ (D.G)' ]
//          [ (G't)'    ]  =  [  t'    ]  G          with D = sum of the sides' lambda / s, t = sum of +-(lambda / s) r_p
// on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): the sum over the functionals happens inside the instruction -- no
// per-lane accumulators (NV (NV + 1) / 2 of them in a vector-ALU formulation) and no cross-lane reduction.  A operand
// (16 x 4 per k-step): lane l holds column k = l / 16 of row i = l % 16, i.e. D_f g_f[16 I + i] of functional f = 4 ks + k
// (row NV: t_f); B operand (4 x 16): g_f[16 J + j], j = l % 16 -- the same LDS value, read once.  C/D: lane l holds
// rows (l / 16) + 4 reg, column l % 16 of a tile.  The tiles go to LDS as full rows of M (mt, stride LDM) and row NV
// as the vector G't (gdr).
typedef double v4d __attribute__((ext_vector_type(4)));
template <class SH>
__device__ __forceinline__ void sweep_a_dense(const double *Gt, int nks, const double (&s)[SH::RS], const double (&lam)[SH::RS],
                                              const double *rpw, double (&rs)[SH::RS], double &gap_l, double &rpn_l,
                                              double *dtw, double *mt, double *gdr, int lane) {
    constexpr int NV = SH::NV, NB = SH::NB, LDG = SH::LDG, IT = NV / 16, CT = NV % 16;
    static_for<SH::FD>([&](auto kd_) {
        constexpr int kd = decltype(kd_)::value;
        double D = 0.0, t = 0.0;
#pragma unroll
        for (int sd = 0; sd < SH::dsides(kd); ++sd) {
            constexpr int base = SH::dbase(kd);
            const int i = base + sd;
            const double rsi = fast_rcp(s[i]);
            rs[i] = rsi;
            const double d = lam[i] * rsi;
            const double rpi = rpw[i * WAVE + lane];
            gap_l = fma(s[i], lam[i], gap_l);
            rpn_l = vmax_abs(rpn_l, rpi);
            D += d;
            t = sd ? fma(-d, rpi, t) : fma(d, rpi, t);
        }
        dtw[2 * (kd * WAVE + lane)] = D;
        dtw[2 * (kd * WAVE + lane) + 1] = t;
    });
    wave_lds_fence();
    const int c = lane & 15, kq = lane >> 4;
    const double tmask = (c == CT) ? 1.0 : 0.0;
    // Two k-steps in flight: the operands of step ks + 1 are on their way while step ks multiplies (an LDS round trip per
    // step otherwise), and with a single tile (NB = 1) even and odd steps accumulate into tiles of their own, so that
    // successive matrix instructions do not wait for each other's result.
    constexpr int NACC = SH::NTL == 1 ? 2 : 1;
    v4d acc[NACC][SH::NTL];
#pragma unroll
    for (int h2 = 0; h2 < NACC; ++h2)
#pragma unroll
        for (int q = 0; q < SH::NTL; ++q) acc[h2][q] = v4d{0.0, 0.0, 0.0, 0.0};
    // (k-step ks reads functional 4 ks + kq; the look-ahead fetches k-step nks at most: ZERO_ROWS)
    const double *dp = dtw + 2 * kq, *gp = Gt + kq * LDG + c;
    auto fetch = [&](double &Dk, double &tk, double (&g)[NB]) {
        Dk = dp[0];
        tk = dp[1];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) g[blk] = gp[16 * blk];
        dp += 8;
        gp += 4 * LDG;
    };
    auto multiply = [&](auto h_, double Dk, double tk, const double (&g)[NB]) {
        constexpr int h2 = decltype(h_)::value;
        double a[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) a[blk] = Dk * g[blk];
        a[IT] = fma(tmask, tk, a[IT]);              // (column NV of G is zero: row NV of the product is G't)
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J)
                acc[h2][I * (I + 1) / 2 + J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], g[J], acc[h2][I * (I + 1) / 2 + J], 0, 0, 0);
    };
    {
        double D0, t0, g0[NB], D1, t1, g1[NB];
        fetch(D0, t0, g0);
        int ks = 0;
        for (; ks + 1 < nks; ks += 2) {
            fetch(D1, t1, g1);
            multiply(std::integral_constant<int, 0>{}, D0, t0, g0);
            fetch(D0, t0, g0);                      // (past the end: zero rows, never multiplied)
            multiply(std::integral_constant<int, NACC - 1>{}, D1, t1, g1);
        }
        if (ks < nks) multiply(std::integral_constant<int, 0>{}, D0, t0, g0);
    }
    static_assert(WaveLds<SH>::RED >= 2 * (SH::NDP + 4), "look-ahead of the MFMA pass stays inside the transposition tile");
    double *mtk = mt + kq * SH::LDM + c;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * I + kq + 4 * reg, col = 16 * J + c;
                double v = acc[0][I * (I + 1) / 2 + J][reg];
                if constexpr (NACC == 2) v += acc[1][I * (I + 1) / 2 + J][reg];
                if (row < NV && col < NV) {
                    mtk[(16 * I + 4 * reg) * SH::LDM + 16 * J] = v;
                    if (I != J) mt[col * SH::LDM + row] = v;
                }
                if (I == IT && row == NV && col < NV) gdr[col] = v;
            }
    wave_lds_fence();
}

// The FACTORED functionals: KC-wide left factor, so W = Hc' D Hc has only KT entries; one pass.
template <class SH>
__device__ __forceinline__ void sweep_a_factored(const double *Hct, const double (&s)[SH::RS], const double (&lam)[SH::RS],
                                                 const double *rpw, double (&rs)[SH::RS], double &gap_l, double &rpn_l,
                                                 double *red, double *csums, int lane) {
    constexpr int KC = SH::KC, KT = SH::KT, NCCP = SH::NCCP;
    double acc[KT + KC + 1];   // initialised by the first slot's products; the last entry carries the wave's complementarity gap
    static_for<SH::FC>([&](auto kc_) {
        constexpr int kc = decltype(kc_)::value;
        const int r = lane + kc * WAVE;
        double hc[KC];
#pragma unroll
        for (int a = 0; a < KC; ++a) hc[a] = Hct[a * NCCP + r];
        double D = 0.0, t = 0.0;
#pragma unroll
        for (int sd = 0; sd < SH::csides(kc); ++sd) {
            constexpr int base = SH::cbase(kc);
            const int i = base + sd;
            const double rsi = fast_rcp(s[i]);
            rs[i] = rsi;
            const double d = lam[i] * rsi;
            const double rpi = rpw[i * WAVE + lane];
            gap_l = fma(s[i], lam[i], gap_l);
            rpn_l = vmax_abs(rpn_l, rpi);
            D += d;
            t = sd ? fma(-d, rpi, t) : fma(d, rpi, t);
        }
#pragma unroll
        for (int a = 0; a < KC; ++a) {
            const double dg = D * hc[a];
#pragma unroll
            for (int b2 = a; b2 < KC; ++b2) acc[col_off<KC>(a) + b2 - a] = kc == 0 ? dg * hc[b2] : fma(dg, hc[b2], acc[col_off<KC>(a) + b2 - a]);
            acc[KT + a] = kc == 0 ? hc[a] * t : fma(hc[a], t, acc[KT + a]);
        }
    });
    // (the gap of ALL row sides of the lane rides along: a free place of the second round instead of a wave reduction of its
    // own -- eight DPP moves, eight v_readlane and a dependent chain of a hundred and fifty cycles)
    acc[KT + KC] = gap_l;
    wave_reduce_to_lds<KT + KC + 1, SH::RR>(acc, red, csums, lane);
}

// FUSED = true: the FUSED closed loop (tmpc_fused.hip).  An item of the work counter is then a TRAJECTORY: the wave that draws it
// solves its QP at t = 0 .. T-1 and runs the trajectory's state machines (mcstep::mc_step_wave, tmpc_mc_step.hpp) between two
// solves -- no launch, no host and no other wave between the steps of a trajectory, and the launch's tail (the waves that got
// the slowest instances last) is paid once per sweep instead of once per time step.  The solve reads the estimate and the
// reference the state machines wrote (mc.st.x_hat, mc.st.ref_k: plain pointers, x_k / ref are not used) and the state machines
// read the solve's outputs through the kernel's own pointers; a fence stands on either side.
// The closed loop's record (model, state arrays, T, the reference sequence) lies in device memory and is read through the constant
// address space from an opaque copy of its address: a scalar load per field where it is used.  As a by-value kernel argument its
// hundred-odd scalar registers stayed live through the whole solve (81 ... 113 spilled vector registers in the two-waves-per-SIMD shapes).
struct McNone {};
// FUSED = 0: no closed loop in the kernel.  1: a work item is a trajectory, all T steps (closed_loop_kernel).  2: a work item is one
// QP of time step t, followed by the state machines of its trajectory for that step (closed_loop_step_kernel: the extended controller,
// whose two problems are two kernel shapes -- one launch per problem and step, none for the state machines).
struct McStepArg {
    const McFused *rec;
    int t;
    uint8_t *gamma_out;          // arrival flags written by this step = problem selector of the NEXT step (the selector read in this
};                               // step is the other of two buffers: a trajectory must not be taken by both launches of a step)
template <int FUSED> using McArg = std::conditional_t<FUSED == 1, const McFused *, std::conditional_t<FUSED == 2, McStepArg, McNone>>;
#if defined(TMPC_HOST_SIM)
typedef const McFused *McRecord;
__device__ __forceinline__ McRecord mc_record(const McFused *p) { return p; }
#else
typedef const __attribute__((address_space(4))) McFused *McRecord;
__device__ __forceinline__ McRecord mc_record(const McFused *p) {
    unsigned lo = static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)), hi = static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return (McRecord)((static_cast<uintptr_t>(hi) << 32) | lo);
}
#endif

// The body of both kernels below (inlined into each: one persistent workgroup per CU, a wave per work item)
template <int NV, int DP, int DS, int KC, int CP, int CS, int WPB, int FUSED>
__device__ __forceinline__ void solve_body(
    const DeviceQP &qp, const int variant_id, const int64_t B,
    const double *__restrict__ x_k, const double *__restrict__ ref, const uint8_t *__restrict__ variant,
    double *__restrict__ u_nom, double *__restrict__ x_nom0, double *__restrict__ xu_ss,
    double *__restrict__ x_nom, int32_t *__restrict__ status, int32_t *__restrict__ iters,
    const int32_t *ws_in, int32_t *ws_out, unsigned long long *__restrict__ next_item,      // (ws_in may alias ws_out: the closed loop updates the records in place)
    const McArg<FUSED> mc) {
    using SH = Shape<NV, DP, DS, KC, CP, CS, tile_rows(NV, WPB)>;
    using WL = WaveLds<SH>;
    constexpr int RS = SH::RS, FD = SH::FD, FC = SH::FC, NDP = SH::NDP, NCCP = SH::NCCP, KT = SH::KT, WCAP = SH::WCAP, LDG = SH::LDG;
#ifdef TMPC_HOST_SIM
    double *smem = sim::lds<double>();
#else
    extern __shared__ __attribute__((aligned(16))) double smem[];
#endif
    double *wbase = smem;                      // WPB per-wave workspaces
    double *Hct = wbase + WPB * WL::TOTAL;     // [KC][NCCP]
    double *Psi = Hct + KC * NCCP;             // [KC][NV]
    double *Hs = Psi + KC * NV;                // [NV][LDH]
    double *Hinv = Hs + NV * SH::LDH;          // [NV][LDH]
    double *Gt = Hinv + NV * SH::LDH;              // [grows + ZERO_ROWS][LDG]  dense functionals in use, row-major (odd stride), then zero rows
    const int grows = 4 * qp.nks;              // (last in the layout: its size is the only one that depends on the problem)

    const int tid = threadIdx.x;
    const int lane_k = tid & (WAVE - 1);
    int lane = lane_k;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the workspace addresses stay scalar
    const int nx = qp.nx, nu = qp.nu, N = qp.N, nc = qp.nc;

    // ---- stage the shared model once per workgroup (coalesced, L2-resident source)
    for (int i = tid; i < (grows + ZERO_ROWS) * LDG; i += blockDim.x) Gt[i] = (i < grows * LDG) ? qp.Gt[i] : 0.0;
    for (int i = tid; i < KC * NCCP; i += blockDim.x) Hct[i] = qp.Hct[i];
    for (int i = tid; i < KC * NV; i += blockDim.x) Psi[i] = qp.Psi[i];
    for (int i = tid; i < NV * NV; i += blockDim.x) { const int r_ = i / NV, c_ = i - r_ * NV; Hs[r_ * SH::LDH + c_] = qp.Hs[i]; Hinv[r_ * SH::LDH + c_] = qp.Hinv[i]; }
    __syncthreads();

    double *red = wbase + wave * WL::TOTAL;       // transposition tile / refinement workspace
    double *sums = red + WL::BIG;                 // dense totals: triangle (column-major packed), two vectors
    double *csums = sums + WL::SUMS;              // factored-block totals: W (packed), two kc-vectors
    double *Pm = csums + WL::CSUMS;               // [KC][NV] W * Psi
    double *hw = Pm + WL::PMAT;                   // h, [side][lane]
    double *vec = hw + WL::HROW;
    double *dtw = red;                // [NDP][2] (D, t) of the dense functionals during the MFMA pass of sweep A (the tile is idle then)
    double *Mf = red + WL::RED;       // [NV][LDM]  normal matrix, then its factor (behind the transposition tile, inside the refinement's idle workspace)
    unsigned *park = reinterpret_cast<unsigned *>(red + WL::POL);   // [RS][64] parked (s, lambda), see WaveLds::PARK_LDS
    double *qv = vec;                 // [NV] linear term
    double *zv = vec + NV;            // [NV] current z (wave-uniform copy)
    double *cgv = vec + 2 * NV;       // [NV] cost gradient
    double *xin = vec + 3 * NV;       // [2*nx] x_k | ref   (nx <= 16)
    double *tv = vec + 3 * NV + 32;   // [NV] scratch
    double *uv = vec + 4 * NV + 32;   // [NV] scratch
    double *dzav = vec + 5 * NV + 32; // [NV] affine direction
    double *dzv = vec + 6 * NV + 32;  // [NV] final direction
    double *czv = vec + 7 * NV + 32;  // [8] Psi z
    double *cdzav = czv + 8;          // [8] Psi dz_aff
    double *cdzv = czv + 16;          // [8] Psi dz        (KC <= 8 and 24 <= 2 NV whenever KC > 0)

    // which of this lane's row sides are real rows (bit i: side i); padding sides keep s = 1, lambda = 0
    const int vmask_k = static_cast<int>(qp.vmask[lane]);
    unsigned vmask = static_cast<unsigned>(vmask_k);
    auto valid = [&](int i) { return ((vmask >> i) & 1u) != 0u; };
#define TMPC_REFRESH() do { lane = fresh(lane_k); vmask = static_cast<unsigned>(fresh(vmask_k)); } while (0)

    // Work distribution: the waves of the persistent grid draw instances from one counter (zeroed by the host before the
    // launch).  The cost of an instance varies by 2 x cold (10 ... 19 interior-point iterations) and by 5 x in the
    // warm-started closed loop (refinement only, or refinement + cold solve), so a static split leaves most SIMDs idle
    // while the unluckiest finishes.  Every wave leaves the loop at its first draw >= B.
    // The first instance of a wave is fixed (its index in the grid): two thousand simultaneous draws on one word at kernel
    // start cost more than the balance gains; the later draws are spread over time by the very imbalance they repair.
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * WPB;
    bool first_item = true;
    for (;;) {
        int64_t b;
        if (first_item) {
            // (wave-major: wave w of workgroup k takes item w * gridDim.x + k, so that a batch smaller than the card's resident waves
            // spreads over the CUs -- one wave per SIMD, or per CU -- instead of filling a few CUs with two waves per SIMD)
            // -- and block-major for a full batch: the eight instances of a workgroup are then neighbours in the output arrays (wave-major
            // throughout cost 0.4 MB more HBM traffic per launch of 4096)
            b = B < n_waves ? static_cast<int64_t>(wave) * gridDim.x + blockIdx.x : static_cast<int64_t>(blockIdx.x) * WPB + wave;
            first_item = false;
        } else {
            // a plain look first: at the end of the launch every wave would otherwise add one failing draw to a burst of
            // two thousand on the same word (device-scope atomics on one address retire at some 15 per microsecond)
            unsigned long long drawn = ~0ull >> 2;
            if (lane_k == 0) {
                const unsigned long long seen = __hip_atomic_load(next_item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n_waves + static_cast<int64_t>(seen) < B) drawn = atomicAdd(next_item, 1ull);
            }
            b = n_waves + static_cast<int64_t>((static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(drawn >> 32)))) << 32) |
                                               static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(drawn & 0xffffffffull))));
        }
        if (b >= B) break;
        if (variant != nullptr && variant[b] != variant_id) continue;
        if (variant == nullptr && variant_id != 0) continue;

        int n_steps = 1;
        if constexpr (FUSED == 1) n_steps = mc_record(mc)->T;
        for (int t_mc = 0; t_mc < n_steps; ++t_mc) {        // (FUSED: the time steps of trajectory b; otherwise the one solve of instance b)
        // ------------------------------------------------------------ per-instance data
        TMPC_REFRESH();
        if constexpr (FUSED != 0) {
            McRecord mr;
            if constexpr (FUSED == 1) mr = mc_record(mc); else mr = mc_record(mc.rec);
            const double *xh = mr->st.x_hat, *rk = mr->st.ref_k;
            if (lane < nx) { xin[lane] = xh[b * nx + lane]; xin[nx + lane] = rk[b * nx + lane]; }
        } else {
            if (lane < nx) { xin[lane] = x_k[b * nx + lane]; xin[nx + lane] = ref[b * nx + lane]; }
        }
        wave_lds_fence();
        int st = TMPC_STATUS_MAX_ITER;
        int it_done = 0;
#ifdef TMPC_ITERS_TOTAL
        int n_reruns = 0, n_rounds = 0, amb_level = 0;
#endif
        const long long t_begin = qp.ticks ? static_cast<long long>(__builtin_amdgcn_s_memrealtime()) : 0;
#ifdef TMPC_STAMPS
        long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_amdgcn_s_memtime();
#endif
        bool infeasible_par = qp.always_infeasible != 0;
        for (int r = lane; r < qp.npar; r += WAVE) {
            double v = qp.gp0[r];
            for (int c = 0; c < nx; ++c) v += qp.Ep[r * nx + c] * xin[c];
            if (v < -1e-9 * (1.0 + fabs(qp.gp0[r]))) infeasible_par = true;
        }
        infeasible_par = __any(infeasible_par);

        if (lane < NV) {
            double v = 0.0;
            if (lane < qp.nv)
                for (int c = 0; c < nx; ++c) v += qp.F1s[lane * nx + c] * xin[c] + qp.F2s[lane * nx + c] * xin[nx + c];
            qv[lane] = v;
        }
        // h = g0 + E x_k into hw; returns this lane's max |h| over its real rows.  The interior-point phase keeps the carried
        // primal residual in the same LDS region (it never reads h), so h is formed again before anything that needs it.
        auto compute_h = [&]() {
            TMPC_REFRESH();
            // h = g0 + E x_k: one plane of E per state, the RS loads of a plane are issued together (nx round trips to L2
            // instead of RS nx: this runs at set-up and again at every hand-over, its LDS region holds r_p in between)
            // (four planes at a time, all of their loads in flight together with those of g0: a round trip to L2 costs some 2000
            // cycles here, and plane by plane -- round 3 -- the cart-pole's h took five of them: 11 k cycles per call, at the
            // set-up and again at every hand-over, 3 % of an instance)
            double hv[RS];
            const double *__restrict__ E0 = qp.Esp + lane;
            constexpr size_t PL = static_cast<size_t>(RS) * WAVE;
            for (int c0 = 0; c0 < nx; c0 += 4) {
                double ev[4][RS];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t cu = static_cast<size_t>(c0 + u < nx ? c0 + u : nx - 1);      // (planes beyond nx: a valid address, weight 0)
#pragma unroll
                    for (int i = 0; i < RS; ++i) ev[u][i] = E0[cu * PL + i * WAVE];
                }
                if (c0 == 0) {
#pragma unroll
                    for (int i = 0; i < RS; ++i) hv[i] = qp.g0p[i * WAVE + lane];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double xc = c0 + u < nx ? xin[c0 + u] : 0.0;
#pragma unroll
                    for (int i = 0; i < RS; ++i) hv[i] = fma(ev[u][i], xc, hv[i]);
                }
            }
            double hmax = 1.0;
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                if (valid(i)) hmax = fmax(hmax, fabs(hv[i]));
                hw[i * WAVE + lane] = hv[i];
            }
            wave_lds_fence();
            return hmax;
        };
        const double hn = wave_max(compute_h());
        double qn = 1.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) qn = fmax(qn, fabs(qv[j]));
        qn = readlane_d(qn, 0);              // (every lane holds the same value: keep it in scalar registers for the whole solve)

        // z = -Hinv q  -> zv, Psi z -> czv
        auto unconstrained_minimiser = [&]() {
            TMPC_REFRESH();
            if (lane < NV) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) v -= Hinv[lane * SH::LDH + j] * qv[j];
                zv[lane] = v;
            }
            wave_lds_fence();
            coords_lds<SH>(Psi, zv, czv, lane);
            wave_lds_fence();
        };
        // raw slacks h - G z of this lane's sides (z in zv / czv); returns the smallest over the valid sides of the lane
        auto raw_slacks = [&](double (&sl)[RS]) {
            TMPC_REFRESH();
            double smin = INFINITY;
            static_for<FD>([&](auto kd_) {
                constexpr int kd = decltype(kd_)::value;
                const double gz = dense_dot<SH, kd>(Gt, grows, zv, lane);
#pragma unroll
                for (int sd = 0; sd < SH::dsides(kd); ++sd) {
                    const int i = SH::dbase(kd) + sd;
                    sl[i] = hw[i * WAVE + lane] - (sd ? -gz : gz);
                    if (valid(i)) smin = vmin(smin, sl[i]);
                }
            });
            static_for<FC>([&](auto kc_) {
                constexpr int kc = decltype(kc_)::value;
                const double gz = fact_dot<SH, kc>(Hct, czv, lane);
#pragma unroll
                for (int sd = 0; sd < SH::csides(kc); ++sd) {
                    const int i = SH::cbase(kc) + sd;
                    sl[i] = hw[i * WAVE + lane] - (sd ? -gz : gz);
                    if (valid(i)) smin = vmin(smin, sl[i]);
                }
            });
            return smin;
        };

        // Control flow of one instance.  The interior-point phase and the active-set refinement never hold registers at
        // the same time: at the hand-over the working set {lambda_i > s_i} and its multipliers are all the refinement
        // takes from (s, lambda), which go to the wave's save slot in HBM (DeviceQP::save).  Should the refinement fail to
        // certify its set (rare: 3 of the 4096 bench instances), the interior-point phase CONTINUES from the saved iterate with
        // a 100 x tighter hand-over tolerance (without a save slot it is re-run from its start, which retraces the same iterates).
        // With a working set handed in by the caller (ws_in: the previous time step's, closed loop) the refinement is
        // tried on it BEFORE any interior-point iteration; it is exact or rejected, never approximate.
        double try_tol = qp.tol;
        const double ncd = static_cast<double>(nc);
        double rdn_last = 0.0;
        bool floor_tried = false;      // the refinement has had the iterate of a stalled gap
        bool inW[RS];
        double yall[RS];
        int warm_m = 0;
        {
            unconstrained_minimiser();
            double sl[RS];
            const double smin = wave_min(raw_slacks(sl));
            if (infeasible_par) st = TMPC_STATUS_INFEASIBLE;
            else if (smin >= 0.0) st = TMPC_STATUS_OPTIMAL;
            if (ws_in != nullptr) warm_m = ws_in[b * WS_STRIDE];
        }
        STAMP(0);
        bool try_warm = warm_m > 0 && warm_m <= WCAP;
        int resume_it = -1;        // >= 0: continue the interior-point phase at this iteration from the saved (s, lambda)
        float *const save = qp.save ? qp.save + (static_cast<size_t>(blockIdx.x) * WPB + wave) * (2 * RS * WAVE) : nullptr;
        bool saved = false;        // (s, lambda) of the last hand-over are in the save slot
        bool h_valid = true;
        int ws_m = 0;
        if (st == TMPC_STATUS_MAX_ITER)
        for (;;) {
            bool want_polish = false;
            if (try_warm) {
#pragma unroll
                for (int i = 0; i < RS; ++i) { inW[i] = false; yall[i] = 0.0; }
                for (int k = 0; k < warm_m; ++k) {
                    const int gid = ws_in[b * WS_STRIDE + 1 + k];
#pragma unroll
                    for (int i = 0; i < RS; ++i) inW[i] = inW[i] || (gid == i * WAVE + lane && valid(i));
                }
                want_polish = true;
            } else {
                // -------------------------------------------------------- interior point
                double s[RS], lam[RS], rs[RS];
                double *rpw = hw;       // carried primal residual, [side][lane], in the region of h
                int it0 = 0;
                double mu_hand = 0.0;       // mu of the iterate that is handed over
                if (resume_it >= 0) {
                    // the refinement did not certify its working set: on with the interior-point iteration from the iterate of
                    // the hand-over (z is still in LDS, (s, lambda) come back from the wave's save slot, r_p is formed anew)
                    if (!h_valid) { compute_h(); h_valid = true; }
                    coords_lds<SH>(Psi, zv, czv, lane);
                    wave_lds_fence();
                    (void)raw_slacks(s);
                    h_valid = false;
                    const double park_mu = WL::PARK_LDS ? *reinterpret_cast<const double *>(park + RS * WAVE) : 0.0;
                    const double park_thr = 1e3 * try_tol * hn;        // (try_tol is a hundredth of the hand-over's by now)
#pragma unroll
                    for (int i = 0; i < RS; ++i) {
                        const bool vl = valid(i);
                        const double raw = s[i];
                        double sv, lv;
                        if constexpr (WL::PARK_LDS) {
                            // h - G z is the slack up to the primal residual of the hand-over (<= tol hn): above ten times that
                            // it IS the slack; below, the row sits on its bound as far as h - G z can tell and takes the
                            // central path's mu / lambda, kept inside that band
                            lv = fmax(static_cast<double>(__uint_as_float(park[i * WAVE + lane])), 1e-30);
                            sv = raw > park_thr ? raw : fmin(fmax(raw, park_mu * fast_rcp(lv)), park_thr);
                        } else {
                            sv = static_cast<double>(save[i * WAVE + lane]);
                            lv = static_cast<double>(save[(RS + i) * WAVE + lane]);
                            // the rounding of s must not come back as a primal residual s - (h - G z): where the parked slack is
                            // the raw slack to single precision, the raw slack is the better copy
                            if (raw > 0.0 && fabs(sv - raw) <= 2.5e-7 * sv) sv = raw;
                        }
                        s[i] = vl ? sv : 1.0;
                        lam[i] = vl ? lv : 0.0;
                        rpw[i * WAVE + lane] = vl ? sv - raw : 0.0;
                        rs[i] = 1.0;
                    }
                    wave_lds_fence();
                    it0 = resume_it;
                    resume_it = -1;
                } else {
                    if (!h_valid) { compute_h(); h_valid = true; }
                    unconstrained_minimiser();
                    const double smin = wave_min(raw_slacks(s));
                    const double fl = 0.1 * fmax(-smin, 1.0);
                    h_valid = false;
                    // Starting multipliers.  The QP with row r ALONE has the multiplier viol_r / (g_r Hs^-1 g_r') at its
                    // minimiser: the largest of them over the violated rows is the scale the multipliers have to reach (1e1 ...
                    // 1e3 for the cart-pole, whose cost weights span seven decades), and the interior-point phase grows its
                    // multipliers by a decade or so per iteration.  lambda_0 = (that scale)^(1/4), between 1 and 1e3 --
                    // round 3 started from 1 whatever the problem: the iteration count falls by 3 - 7 % and its upper tail,
                    // which sets the launch time of a batch, by more (the oracle and the block kernel alike).
                    double l1 = 0.0;
#pragma unroll
                    for (int i = 0; i < RS; ++i) l1 = vmax(l1, valid(i) ? -s[i] * qp.cip[i * WAVE + lane] : 0.0);
                    l1 = wave_max(l1);
#ifdef TMPC_LAM0_ONE
                    const double lam0 = l1 < 0.0 ? 2.0 : 1.0;      // (diagnostic builds: the round-3 start)
#else
                    const double lam0 = fmin(fmax(sqrt(sqrt(l1)), 1.0), 1e3);
#endif
#pragma unroll
                    for (int i = 0; i < RS; ++i) {
                        const bool vl = valid(i);
                        const double raw = s[i];
                        s[i] = vl ? fmax(raw, fl) : 1.0;
                        lam[i] = vl ? lam0 : 0.0;
                        rpw[i * WAVE + lane] = vl ? s[i] - raw : 0.0;   // r_p = G z + s - h with h - G z = raw; carried from here on:
                        rs[i] = 1.0;                                    // the Newton step gives r_p <- (1 - alpha) r_p exactly (ds = -r_p - G dz)
                    }
                    wave_lds_fence();
                }
                for (int it = it0; it < qp.max_iter; ++it) {
                    it_done = it;
                    STAMP(9);
                    TMPC_REFRESH();
                    // ---- sweep A: 1/s, weights, gap, |r_p|, G'DG (dense functionals by column blocks, factored ones as W), G'(d.r_p)
                    double gap_l = 0.0, rpn_l = 0.0;
                    if constexpr (FD > 0) sweep_a_dense<SH>(Gt, qp.nks, s, lam, rpw, rs, gap_l, rpn_l, dtw, Mf, sums, lane);
                    if constexpr (KC > 0) {
                        TMPC_REFRESH();
                        sweep_a_factored<SH>(Hct, s, lam, rpw, rs, gap_l, rpn_l, red, csums, lane);
                        // fold the factored block into the dense totals: P = W Psi now, Psi' P when the rows of M are formed
                        for (int idx = lane; idx < KC * NV; idx += WAVE) {
                            const int a = idx / NV, j = idx - a * NV;
                            double v = 0.0;
#pragma unroll
                            for (int b2 = 0; b2 < KC; ++b2) {
                                const int lo = a < b2 ? a : b2, hi2 = a < b2 ? b2 : a;
                                v += csums[lo * KC - lo * (lo - 1) / 2 + hi2 - lo] * Psi[b2 * NV + j];
                            }
                            Pm[idx] = v;
                        }
                        double v1 = 0.0;
                        if (lane < NV) {
                            v1 = (FD > 0) ? sums[lane] : 0.0;
#pragma unroll
                            for (int a = 0; a < KC; ++a) v1 += Psi[a * NV + lane] * csums[KT + a];
                        }
                        wave_lds_fence();
                        if (lane < NV) sums[lane] = v1;
                        wave_lds_fence();
                    }
                    // |r_p| and max lambda are only ever compared with thresholds: a ballot each instead of a reduction
                    double lmax_l = 0.0;
#pragma unroll
                    for (int i = 0; i < RS; ++i) lmax_l = vmax(lmax_l, lam[i]);
                    const bool rp_small = !__any(rpn_l > try_tol * hn);
                    const bool lam_big = __any(lmax_l > 1e10);
                    double gap;
                    if constexpr (KC > 0) gap = readlane_d(csums[KT + KC], 0);       // (summed with the factored block's totals: sweep_a_factored)
                    else gap = wave_sum(gap_l);
                    const double mu = gap / ncd;
                    STAMP(1);
                    TMPC_REFRESH();
                    // cost gradient cg = Hs z + q (lane i computes entry i)
                    if (lane < NV) {
                        double v = 0.0;
#pragma unroll
                        for (int j = 0; j < NV; ++j) v += Hs[lane * SH::LDH + j] * zv[j];
                        cgv[lane] = v + qv[lane];
                    }
                    wave_lds_fence();
                    double obj = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        const double cgj = cgv[j], qj = qv[j];
                        obj += zv[j] * (0.5 * (cgj - qj) + qj);
                    }
                    if (!(mu == mu)) { st = TMPC_STATUS_NUMERICAL; break; }
                    const double objs = fmax(fabs(obj), 1.0);
                    STAMP(2);
                    // The dual residual needs G'lam, which nothing else uses: it is formed (one more pass over the functionals)
                    // only when the primal residual and the gap already pass, or when the multipliers blow up (Farkas test).
                    const bool near = rp_small && (gap <= try_tol * objs);
                    if (near || lam_big) {
                        if constexpr (FD > 0) {
                            double accl[NV];
#pragma unroll
                            for (int i = 0; i < NV; ++i) accl[i] = 0.0;
                            static_for<FD>([&](auto kd_) {
                                constexpr int kd = decltype(kd_)::value;
                                const int r = min(lane + kd * WAVE, grows);
                                double dl2 = lam[SH::dbase(kd)];
                                if constexpr (SH::dsides(kd) == 2) dl2 -= lam[SH::dbase(kd) + 1];
#pragma unroll
                                for (int j = 0; j < NV; ++j) accl[j] = fma(Gt[r * LDG + j], dl2, accl[j]);
                                if constexpr (WPB == 8 || TMPC_FENCE_ALL) row_fence();
                            });
                            wave_reduce_to_lds<NV, SH::RR>(accl, red, sums + NV, lane);
                        }
                        if constexpr (KC > 0) {
                            double accl[KC];
#pragma unroll
                            for (int a = 0; a < KC; ++a) accl[a] = 0.0;
                            static_for<FC>([&](auto kc_) {
                                constexpr int kc = decltype(kc_)::value;
                                const int r = lane + kc * WAVE;
                                double dl2 = lam[SH::cbase(kc)];
                                if constexpr (SH::csides(kc) == 2) dl2 -= lam[SH::cbase(kc) + 1];
#pragma unroll
                                for (int a = 0; a < KC; ++a) accl[a] = fma(Hct[a * NCCP + r], dl2, accl[a]);
                            });
                            wave_reduce_to_lds<KC, SH::RR>(accl, red, csums + KT + KC, lane);
                            double gl = 0.0;
                            if (lane < NV) {
                                gl = (FD > 0) ? sums[NV + lane] : 0.0;
#pragma unroll
                                for (int a = 0; a < KC; ++a) gl += Psi[a * NV + lane] * csums[KT + KC + a];
                            }
                            wave_lds_fence();
                            if (lane < NV) sums[NV + lane] = gl;
                            wave_lds_fence();
                        }
                        double rdn = 0.0, gn = 0.0;
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            const double glj = sums[NV + j];
                            rdn = fmax(rdn, fabs(cgv[j] + glj));
                            gn = fmax(gn, fabs(glj));
                        }
                        if (!(rdn == rdn)) { st = TMPC_STATUS_NUMERICAL; break; }
#ifdef TMPC_DEBUG_PRINT
                        if (lane == 0 && b < 2) printf("b %lld it %d rdn %.3e qn %.3e\n", (long long)b, it, rdn, qn);
#endif
                        if (near && rdn <= 1e3 * try_tol * qn) { want_polish = true; rdn_last = readlane_d(rdn, 0); mu_hand = readlane_d(mu, 0); break; }
                        if (lam_big) {
                            // Farkas-type certificate: lam blows up, G'lam -> 0, h'lam < 0
                            const double lmax = wave_max(lmax_l);
                            double hl = 0.0;       // (h is formed on the fly: its LDS region holds r_p here)
#pragma unroll
                            for (int i = 0; i < RS; ++i) {
                                const int sl = i * WAVE + lane;
                                double hv = qp.g0p[sl];
                                for (int c = 0; c < nx; ++c) hv += qp.Esp[static_cast<size_t>(c) * (RS * WAVE) + sl] * xin[c];
                                hl += hv * lam[i];
                            }
                            hl = wave_sum(hl);
                            if (hl < 0.0 && gn <= 1e-6 * lmax) { st = TMPC_STATUS_INFEASIBLE; break; }
                        }
                    }
#ifdef TMPC_DEBUG_PRINT
                    if (lane == 0 && b < 2) printf("b %lld it %d gap %.3e objs %.3e hn %.3e near %d tol %.1e\n", (long long)b, it, gap, objs, hn, (int)near, try_tol);
#endif
                    if (gap <= 1e-15 * objs) {
                        // nothing left to gain from further iterations (the dual residual stalls on the ill-conditioned systems
                        // of such a small mu and keeps the hand-over test above from passing): the refinement gets this iterate
                        // as it is, once; uncertified it stays MAX_ITER
                        st = TMPC_STATUS_MAX_ITER;
                        if (!floor_tried) { floor_tried = true; want_polish = true; rdn_last = INFINITY; mu_hand = readlane_d(mu, 0); }
                        break;
                    }
                    // ---- M = Hs + G'DG by rows (lane i holds row i), elimination with the predictor rhs carried along
                    TMPC_REFRESH();
                    double rhs_i = 0.0;
                    const int li = lane < NV ? lane : 0;
                    {
                        double mrow[NV], mdinv = 1.0;
                        double shift = 0.0;
                        bool spd = false;
                        for (int attempt = 0; attempt < 2 && !spd; ++attempt) {
                            // M = Hs + G'DG: the dense part sits in the tile (rows of stride LDM, written by the MFMA pass), the
                            // factored part is Psi' (W Psi); the NV^2 entries are spread over the 64 lanes, then lane i reads row i
                            double *mt = Mf + li * SH::LDM;        // (not the transposition tile: the dual-residual pass may have used it)
                            if (attempt == 0) {
#pragma unroll
                                for (int q = 0; q * WAVE < NV * NV; ++q) {
                                    const int e = lane + q * WAVE;
                                    if (e < NV * NV) {
                                        const int i = e / NV, j = e - i * NV;
                                        double v = Hs[i * SH::LDH + j];
                                        if constexpr (FD > 0) v += Mf[i * SH::LDM + j];
                                        if constexpr (KC > 0) {
#pragma unroll
                                            for (int a = 0; a < KC; ++a) v += Psi[a * NV + i] * Pm[a * NV + j];
                                        }
                                        Mf[i * SH::LDM + j] = v;
                                    }
                                }
                            } else if (lane < NV) {
                                mt[li] += shift;
                            }
                            wave_lds_fence();
                            rhs_i = (lane < NV) ? -cgv[li] - sums[li] : 0.0;
                            if constexpr (NV > wv::DPP_ROW && wv::DPP_ROW > 0) {
                                // two matrix rows per lane, DPP forms throughout (tmpc_wave.hpp: rows32_*); the right-hand side goes
                                // through LDS, the factor stays in place of the matrix
                                if (lane < NV) tv[lane] = rhs_i;
                                wave_lds_fence();
                                spd = wv::rows32_factor_solve<NV, SH::LDM>(Mf, tv, dzav, lane);
                                (void)mrow; (void)mdinv;
                            } else {
#pragma unroll
                            for (int j = 0; j < NV; ++j) mrow[j] = mt[j];      // (lanes >= NV eliminate a copy of row 0 that nothing reads)
                            wave_lds_fence();
                            double bb = rhs_i;
                            mdinv = 1.0;
                            spd = lanes_factor<NV>(mrow, bb, mdinv, lane);
                            if (spd) {
                                const double xl = lanes_backsub_lane<NV>(mrow, bb, mdinv, lane);
                                if (lane < NV) {
                                    dzav[lane] = xl;
                                    // the factor waits in LDS for the corrector's solve: 2 NV registers less through sweep B
#pragma unroll
                                    for (int j = 0; j < NV; ++j) Mf[li * SH::LDM + j] = mrow[j];
                                    Mf[li * SH::LDM + NV] = mdinv;
                                }
                            }
                            }
                            if (spd) {
                            } else {
                                // non-positive pivot from cancellation: retry once with a 1e-13 * trace(M) shift (the rows of M are
                                // still in the tile)
                                shift = 1e-13 * wave_sum(lane < NV ? mt[li] : 0.0);
                            }
                        }
                        if (!spd) { st = TMPC_STATUS_NUMERICAL; break; }
                    }
                    wave_lds_fence();
                    if constexpr (KC > 0) { coords_lds<SH>(Psi, dzav, cdzav, lane); wave_lds_fence(); }
                    STAMP(3);
                    TMPC_REFRESH();
                    // ---- sweep B: affine step statistics and the corrector's G' products; w = ds_aff * dl_aff kept per side
                    double rho_aff = 0.0, sb1 = 0.0, sb2 = 0.0;
                    double wv_[RS];
                    double v2 = 0.0, v3 = 0.0;          // lane i < NV: entries i of G'(dsa.dla/s) and G'(1/s)
                    {
                        // one side: returns its contributions (c1, c2) to the two G' products (sign included).
                        // -ds/s = -q and -dl/lam = 1 + q with q = dsa / s (dla = -lam (1 + q)): no reciprocal of lambda.
                        // Padding sides (s = 1, lam = 0, r_p = 0, g = 0) give q = 0: ratio 1, never binding.
                        auto side_stats = [&](int i, bool neg, double gdz, double &c1, double &c2) {
                            const double rpi = rpw[i * WAVE + lane];
                            const double dsa = neg ? gdz - rpi : -rpi - gdz;
                            const double q = dsa * rs[i];
                            const double u = 1.0 + q;
                            const double dla = -lam[i] * u;
                            rho_aff = vmax(rho_aff, vmax(-q, u));
                            const double w = dsa * dla;
                            sb1 = fma(s[i], dla, fma(lam[i], dsa, sb1));
                            sb2 += w;
                            wv_[i] = w;
                            c1 = neg ? fma(-w, rs[i], c1) : fma(w, rs[i], c1);      // (padding functionals have g = 0: their 1/s = 1
                            c2 = neg ? c2 - rs[i] : c2 + rs[i];                      //  never reaches G'(1/s); paired slots hold complete pairs only)
                        };
                        if constexpr (FD > 0) {
                            double accb[2 * NV];
                            static_for<FD>([&](auto kd_) {
                                constexpr int kd = decltype(kd_)::value;
                                const int r = min(lane + kd * WAVE, grows);
                                double g[NV];
                                double gd0 = 0.0, gd1 = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) g[j] = Gt[r * LDG + j];
#pragma unroll
                                for (int j = 0; j + 1 < NV; j += 2) { gd0 = fma(g[j], dzav[j], gd0); gd1 = fma(g[j + 1], dzav[j + 1], gd1); }
                                if (NV & 1) gd0 = fma(g[NV - 1], dzav[NV - 1], gd0);
                                double c1 = 0.0, c2 = 0.0;
#pragma unroll
                                for (int sd = 0; sd < SH::dsides(kd); ++sd) side_stats(SH::dbase(kd) + sd, sd == 1, gd0 + gd1, c1, c2);
#pragma unroll
                                for (int j = 0; j < NV; ++j) {
                                    accb[j] = kd == 0 ? g[j] * c1 : fma(g[j], c1, accb[j]);
                                    accb[NV + j] = kd == 0 ? g[j] * c2 : fma(g[j], c2, accb[NV + j]);
                                }
                                if constexpr (WPB == 8 || TMPC_FENCE_ALL) row_fence();
                            });
                            wave_reduce_to_lds<2 * NV, SH::RR>(accb, red, sums, lane);   // overwrites G'(d.rp), G'lam (consumed)
                        }
                        if constexpr (KC > 0) {
                            double accc[2 * KC + 2];       // (+ the two sums of the affine step's statistics: see sweep_a_factored)
                            static_for<FC>([&](auto kc_) {
                                constexpr int kc = decltype(kc_)::value;
                                const int r = lane + kc * WAVE;
                                double hc[KC];
                                double gdz = 0.0;
#pragma unroll
                                for (int a = 0; a < KC; ++a) { hc[a] = Hct[a * NCCP + r]; gdz = fma(hc[a], cdzav[a], gdz); }
                                double c1 = 0.0, c2 = 0.0;
#pragma unroll
                                for (int sd = 0; sd < SH::csides(kc); ++sd) side_stats(SH::cbase(kc) + sd, sd == 1, gdz, c1, c2);
#pragma unroll
                                for (int a = 0; a < KC; ++a) {
                                    accc[a] = kc == 0 ? hc[a] * c1 : fma(hc[a], c1, accc[a]);
                                    accc[KC + a] = kc == 0 ? hc[a] * c2 : fma(hc[a], c2, accc[KC + a]);
                                }
                            });
                            accc[2 * KC] = sb1;
                            accc[2 * KC + 1] = sb2;
                            wave_reduce_to_lds<2 * KC + 2, SH::RR>(accc, red, csums + KT, lane);
                            sb1 = readlane_d(csums[KT + 2 * KC], 0);
                            sb2 = readlane_d(csums[KT + 2 * KC + 1], 0);
                        } else {
                            sb1 = wave_sum(sb1);
                            sb2 = wave_sum(sb2);
                        }
                        rho_aff = wave_max(rho_aff);
                        if (lane < NV) {
                            if constexpr (FD > 0) { v2 = sums[lane]; v3 = sums[NV + lane]; }
                            if constexpr (KC > 0) {
#pragma unroll
                                for (int a = 0; a < KC; ++a) { v2 += Psi[a * NV + lane] * csums[KT + a]; v3 += Psi[a * NV + lane] * csums[KT + KC + a]; }
                            }
                        }
                    }
                    STAMP(4);
                    TMPC_REFRESH();
                    const double aaff = rho_aff > 1.0 ? 1.0 / rho_aff : 1.0;
                    const double mu_aff = (gap + aaff * sb1 + aaff * aaff * sb2) / ncd;
                    double sigma = mu_aff / mu;
                    sigma = fmin(sigma * sigma * sigma, 1.0);
                    const double smu = sigma * mu;
                    if constexpr (NV > wv::DPP_ROW && wv::DPP_ROW > 0) {
                        if (lane < NV) tv[lane] = rhs_i + v2 - smu * v3;
                        wave_lds_fence();
                        wv::rows32_resolve<NV, SH::LDM>(Mf, tv, dzv, lane);
                    } else {
                        double mrow[NV];
#pragma unroll
                        for (int j = 0; j < NV; ++j) mrow[j] = Mf[li * SH::LDM + j];
                        const double mdinv = Mf[li * SH::LDM + NV];
                        double bb = (lane < NV) ? rhs_i + v2 - smu * v3 : 0.0;
                        lanes_forward<NV>(mrow, bb, lane);
                        const double xl = lanes_backsub_lane<NV>(mrow, bb, mdinv, lane);
                        if (lane < NV) dzv[lane] = xl;
                    }
                    wave_lds_fence();
                    if constexpr (KC > 0) { coords_lds<SH>(Psi, dzv, cdzv, lane); wave_lds_fence(); }
                    STAMP(5);
                    TMPC_REFRESH();
                    // ---- sweep D: final direction (ds -> the register of 1/s, dl -> the register of w), step length, update
                    double om = (1.0 - aaff) * (1.0 - aaff);
                    om = fmin(fmax(om, 1e-4), 1e-2);
                    const double tau = 1.0 - om;
                    double rho = 0.0;
                    {
                        // the reciprocal of lambda only ranks the step-length ratios (rcp1: one Newton step)
                        auto side_step = [&](int i, bool neg, double gdz) {
                            const double rpi = rpw[i * WAVE + lane];
                            const double dsk = neg ? gdz - rpi : -rpi - gdz;
                            const double t1 = fma(lam[i], dsk, wv_[i] - smu);
                            const double dlk = valid(i) ? fma(-t1, rs[i], -lam[i]) : 0.0;
                            const double rl = valid(i) ? rcp1(lam[i]) : 0.0;
                            rho = vmax(rho, vmax(-dsk * rs[i], -dlk * rl));
                            rs[i] = dsk;
                            wv_[i] = dlk;
                        };
                        static_for<FD>([&](auto kd_) {
                            constexpr int kd = decltype(kd_)::value;
                            const double gdz = dense_dot<SH, kd>(Gt, grows, dzv, lane);
#pragma unroll
                            for (int sd = 0; sd < SH::dsides(kd); ++sd) side_step(SH::dbase(kd) + sd, sd == 1, gdz);
                            if constexpr (WPB == 8 || TMPC_FENCE_ALL) row_fence();
                        });
                        static_for<FC>([&](auto kc_) {
                            constexpr int kc = decltype(kc_)::value;
                            const double gdz = fact_dot<SH, kc>(Hct, cdzv, lane);
#pragma unroll
                            for (int sd = 0; sd < SH::csides(kc); ++sd) side_step(SH::cbase(kc) + sd, sd == 1, gdz);
                        });
                    }
                    rho = wave_max(rho);
                    const double alpha = rho > tau ? tau / rho : 1.0;
                    const double oma = 1.0 - alpha;
#pragma unroll
                    for (int i = 0; i < RS; ++i) {
                        s[i] = fma(alpha, rs[i], s[i]);
                        lam[i] = fma(alpha, wv_[i], lam[i]);
                        rpw[i * WAVE + lane] *= oma;
                    }
                    if (lane < NV) zv[lane] += alpha * dzv[lane];
                    wave_lds_fence();
                    it_done = it + 1;
                    STAMP(6);
                }
                // hand-over: the working set and its multipliers are all the refinement takes from (s, lambda)
                TMPC_REFRESH();
#ifdef TMPC_DEBUG_PRINT
                if (lane == 0 && b < 2) printf("b %lld ipm exit: it_done %d st %d want_polish %d tol %.1e it0 %d\n", (long long)b, it_done, st, (int)want_polish, try_tol, it0);
#endif
#pragma unroll
                for (int i = 0; i < RS; ++i) { inW[i] = valid(i) && (lam[i] > s[i]); yall[i] = lam[i]; }
#ifdef TMPC_ITERS_TOTAL
                if (want_polish && amb_level == 0) {
                    // diagnostic: how clearly does (s, lambda) separate the working set?  level k: some row has lambda / s within 10^k of 1
                    amb_level = 9;
#pragma unroll
                    for (int i = 0; i < RS; ++i) {
                        if (valid(i)) {
                            const double lr = fabs(log10(lam[i] / s[i]));
                            const int k = lr < 1.0 ? 1 : (lr < 2.0 ? 2 : (lr < 3.0 ? 3 : (lr < 4.0 ? 4 : (lr < 6.0 ? 6 : 9))));
                            amb_level = k < amb_level ? k : amb_level;
                        }
                    }
                    amb_level = -static_cast<int>(wave_max(-static_cast<double>(amb_level)));
                }
#endif
                if constexpr (WL::PARK_LDS) {
                    // parked in LDS: every hand-over, a store per row side (WaveLds::PARK_LDS has the form of the continuation)
                    if (want_polish) {
#pragma unroll
                        for (int i = 0; i < RS; ++i) park[i * WAVE + lane] = __float_as_uint(static_cast<float>(lam[i]));
                        if (lane == 0) *reinterpret_cast<double *>(park + RS * WAVE) = mu_hand;
                        saved = true;
                        wave_lds_fence();
                    }
                } else if (save != nullptr && want_polish) {
                    // (s, lambda) are parked only when they do not separate the working set clearly -- some row with lambda / s
                    // within two decades of 1: every re-run but 3 of 30 720 closed-loop instances at N = 10 (13 at N = 20) comes
                    // from that fifth of the instances -- and in single precision: the continuation starts from an interior point
                    // 6e-8 away from the iterate, which an interior-point iteration does not notice
                    bool amb = false;
#pragma unroll
                    for (int i = 0; i < RS; ++i) amb = amb || (valid(i) && lam[i] < 100.0 * s[i] && s[i] < 100.0 * lam[i]);
                    saved = __ballot(amb) != 0ull;
                    if (saved) {
#pragma unroll
                        for (int i = 0; i < RS; ++i) {
                            save[i * WAVE + lane] = static_cast<float>(s[i]);
                            save[(RS + i) * WAVE + lane] = static_cast<float>(lam[i]);
                        }
                    }
                }
            }
            if (!want_polish) break;
            // ------------------------------------------------ active-set refinement
            bool ok = false;
            if (!h_valid) { compute_h(); h_valid = true; }
            {
                // workspace carved from the (now idle) transposition tile
                double *GW = red;                         // [WCAP][LDW]  rows of the working set, expanded (sign included)
                double *T = GW + WCAP * SH::LDW;          // [NV][LDT]
                double *yv = T + NV * SH::LDT;            // [WCAP]
                double *dyv = yv + WCAP;                  // [WCAP]
                int *Widx = reinterpret_cast<int *>(dyv + WCAP);   // [WCAP] row ids: side * 64 + lane
                double *zpv = dzav;                       // the refinement's iterate
                if (lane < NV) zpv[lane] = zv[lane];
                wave_lds_fence();
                // (continuing the interior-point phase costs a third of a round per iteration: the first attempt gives up early)
                const int max_rounds = (try_tol == qp.tol && saved && !try_warm) ? 4 : 10;
                int loose_retries = 0;        // rounds that only repeat the Newton steps on an unchanged working set (see below)
                for (int round = 0; round < max_rounds + loose_retries && !ok; ++round) {
#ifdef TMPC_ITERS_TOTAL
                    ++n_rounds;
#endif
                    // compact the working set: W[0..m)
                    TMPC_REFRESH();
                    int m = 0;
                    {
                        // More candidate rows than the refinement can hold (degenerate vertices of the 854-row initial-state set):
                        // the weakest multipliers go first, a decade at a time.  Rows the verification has just added (y = 0) stay;
                        // whatever is wrongly dropped comes back through the verification of ALL rows below.
                        int cnt = 0;
#pragma unroll
                        for (int i = 0; i < RS; ++i) cnt += __popcll(__ballot(inW[i]));
                        if (cnt > WCAP) {
                            double ym = 0.0;
#pragma unroll
                            for (int i = 0; i < RS; ++i) ym = fmax(ym, inW[i] ? yall[i] : 0.0);
                            double thr = 1e-14 * wave_max(ym);
                            for (int tries = 0; tries < 14 && cnt > WCAP; ++tries) {
                                thr *= 10.0;
                                cnt = 0;
#pragma unroll
                                for (int i = 0; i < RS; ++i) {
                                    inW[i] = inW[i] && (yall[i] == 0.0 || yall[i] > thr);
                                    cnt += __popcll(__ballot(inW[i]));
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < RS; ++i) {
                        const unsigned long long bal = __ballot(inW[i]);
                        const int pos = m + __popcll(bal & ((1ull << lane) - 1ull));
                        if (inW[i] && pos < WCAP) { Widx[pos] = i * WAVE + lane; yv[pos] = yall[i]; }
                        m += __popcll(bal);
                    }
                    wave_lds_fence();
                    STAMP(10);
                    if (m > WCAP) break;
                    if (m == 0) {
                        if (lane < NV) {
                            double v = 0.0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) v -= Hinv[lane * SH::LDH + j] * qv[j];
                            zpv[lane] = v;
                        }
                        wave_lds_fence();
                    } else {
                        // expand the working rows: dense rows are copied, factored rows are Hc_r * Psi; the side's sign goes in
                        for (int idx = lane; idx < m * NV; idx += WAVE) {
                            const int k = idx / NV, j = idx - k * NV;
                            const int gid = Widx[k];
                            bool dense;
                            int fs;
                            double sgn;
                            side_info<SH>(gid >> 6, dense, fs, sgn);
                            const int r = fs * WAVE + (gid & 63);
                            double v = 0.0;
                            if (dense) {
                                if constexpr (FD > 0) v = Gt[r * LDG + j];
                            } else {
                                if constexpr (KC > 0) {
#pragma unroll
                                    for (int a = 0; a < KC; ++a) v += Hct[a * NCCP + r] * Psi[a * NV + j];
                                }
                            }
                            GW[k * SH::LDW + j] = sgn * v;
                        }
                        wave_lds_fence();
                        // T = Hinv G_W'  (entry (i,k): i = idx / m, k = idx % m)
                        for (int idx = lane; idx < NV * m; idx += WAVE) {
                            const int i = idx / m, k = idx - i * m;
                            double v = 0.0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) v += Hinv[i * SH::LDH + j] * GW[k * SH::LDW + j];
                            T[i * SH::LDT + k] = v;
                        }
                        wave_lds_fence();
                        STAMP(11);
                        // S = G_W T, its LDL' and the Newton steps, compiled for working sets of at most 12 rows (the common case: the
                        // fully unrolled elimination costs MC^2 / 2 broadcasts whatever m is) and for the full capacity
                        auto solve_working_set = [&]<int MC>(std::integral_constant<int, MC>) -> bool {
                        TMPC_REFRESH();
                        // S = G_W T (+ delta I) by rows in registers: lane a holds row a (identity rows beyond m); LDL' by
                        // readlane elimination like the normal matrix
                        double srow[MC], sdinv = 1.0;
                        {
                            double gw[NV];
                            const int la = lane < m ? lane : 0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) gw[j] = GW[la * SH::LDW + j];
                            double sdiag = 0.0;
#pragma unroll
                            for (int c2 = 0; c2 < MC; ++c2) {
                                double v = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += gw[j] * T[j * SH::LDT + c2];
                                const bool in = lane < m && c2 < m;
                                srow[c2] = in ? v : ((c2 == lane) ? 1.0 : 0.0);
                                if (c2 == lane && in) sdiag = v;
                            }
                            const double dmax = wave_max(sdiag);
#pragma unroll
                            for (int c2 = 0; c2 < MC; ++c2) if (c2 == lane && lane < m) srow[c2] += 1e-11 * dmax;
                        }
                        {
                            double bdummy = 0.0;
                            if (!lanes_factor<MC>(srow, bdummy, sdinv, lane)) return false;
                        }
                        STAMP(12);
                        // proximal Newton steps on the KKT system of the working set (at most twelve -- nearly parallel working rows need them --; they stop once a step
                        // no longer moves the iterate)
                        double dz_prev = 0.0;
                        for (int stp = 0; stp < 12; ++stp) {
                            TMPC_REFRESH();
                            // r1 = Hs zp + q + G_W' y   (lane i -> entry i)
                            if (lane < NV) {
                                double v = qv[lane];
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += Hs[lane * SH::LDH + j] * zpv[j];
                                for (int k = 0; k < m; ++k) v += GW[k * SH::LDW + lane] * yv[k];
                                tv[lane] = v;
                            }
                            wave_lds_fence();
                            // t1 = Hinv r1
                            if (lane < NV) {
                                double v = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += Hinv[lane * SH::LDH + j] * tv[j];
                                uv[lane] = v;
                            }
                            wave_lds_fence();
                            // dy rhs: (G_W zp - h_W) - G_W t1
                            double bb = 0.0;
                            if (lane < m) {
                                double gz = 0.0, gt = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) { const double g = GW[lane * SH::LDW + j]; gz += g * zpv[j]; gt += g * uv[j]; }
                                bb = gz - hw[Widx[lane]] - gt;
                            }
                            lanes_forward<MC>(srow, bb, lane);
                            const double dyl = lanes_backsub_lane<MC>(srow, bb, sdinv, lane);
                            if (lane < m) { dyv[lane] = dyl; yv[lane] += dyl; }
                            wave_lds_fence();
                            // zp -= t1 + T dy  (lane j -> entry j)
                            double dzl = 0.0, zl = 0.0;
                            if (lane < NV) {
                                double v = uv[lane];
                                for (int k = 0; k < m; ++k) v += T[lane * SH::LDT + k] * dyv[k];
                                zl = zpv[lane] - v;
                                zpv[lane] = zl;
                                dzl = fabs(v);
                            }
                            wave_lds_fence();
                            const double dzn = wave_max(dzl), zn = fmax(wave_max(fabs(zl)), 1.0);
                            // the steps contract linearly (ratio ~ delta / lambda_min(S), 1e-11 for well separated rows): stop
                            // when this step no longer moves the iterate, or when what is left after it -- dz rho / (1 - rho)
                            // with the observed ratio rho -- cannot
                            if (stp >= 1) {
                                const double rho = dzn / fmax(dz_prev, 1e-300);
                                if (dzn <= 1e-14 * zn || (loose_retries == 0 && rho < 0.5 && dzn * rho <= 0.5e-15 * zn)) break;
                            }
                            dz_prev = dzn;
                        }
                            return true;
                        };
                        bool fact_ok;
                        if (m <= 12) fact_ok = solve_working_set(std::integral_constant<int, 12>{});
                        else if (WCAP == 24 || m <= 24) fact_ok = solve_working_set(std::integral_constant<int, 24>{});
                        else fact_ok = solve_working_set(std::integral_constant<int, WCAP>{});
                        if (!fact_ok) break;
                    }
                    STAMP(13);
                    // ---- verify: primal feasibility on all rows, sign of y on W
                    TMPC_REFRESH();
                    double ymax = 1.0;
                    for (int k = 0; k < m; ++k) ymax = fmax(ymax, fabs(yv[k]));
                    int nviol = 0, nneg = 0, nloose = 0;
                    double yloose = INFINITY;          // smallest multiplier among this lane's loose working rows
                    {
                        if constexpr (KC > 0) { coords_lds<SH>(Psi, zpv, cdzav, lane); wave_lds_fence(); }
                        int mm = 0;
                        auto check_side = [&](int i, double gz) {
                            const double hk = hw[i * WAVE + lane];
                            const double rr = gz - hk;
                            const unsigned long long bal = __ballot(inW[i]);
                            const int pos = mm + __popcll(bal & ((1ull << lane) - 1ull));
                            mm += __popcll(bal);
                            const double hi = fmax(fabs(hk), 1.0);
                            const bool viol = valid(i) && !inW[i] && rr > 1e-12 * hi;
                            // a working-set row that is not on its bound: the Newton steps have not converged
                            const bool loose = inW[i] && fabs(rr) > 1e-11 * hi;
                            bool neg = false;
                            if (inW[i]) { yall[i] = yv[pos < WCAP ? pos : 0]; neg = yall[i] < -1e-10 * ymax; }
                            nviol += __popcll(__ballot(viol));
                            nneg += __popcll(__ballot(neg));
                            nloose += __popcll(__ballot(loose));
                            if (loose && !neg) yloose = fmin(yloose, yall[i]);
                            if (neg) { inW[i] = false; yall[i] = 0.0; }
                            if (viol) { inW[i] = true; yall[i] = 0.0; }
                        };
                        static_for<FD>([&](auto kd_) {
                            constexpr int kd = decltype(kd_)::value;
                            const double gz = dense_dot<SH, kd>(Gt, grows, zpv, lane);
#pragma unroll
                            for (int sd = 0; sd < SH::dsides(kd); ++sd) check_side(SH::dbase(kd) + sd, sd ? -gz : gz);
                        });
                        static_for<FC>([&](auto kc_) {
                            constexpr int kc = decltype(kc_)::value;
                            const double gz = fact_dot<SH, kc>(Hct, cdzav, lane);
#pragma unroll
                            for (int sd = 0; sd < SH::csides(kc); ++sd) check_side(SH::cbase(kc) + sd, sd ? -gz : gz);
                        });
                    }
                    wave_lds_fence();
                    STAMP(14);
#ifdef TMPC_DEBUG_PRINT
                    if (lane == 0 && b < 2) {
                        printf("b %lld warm %d tol %.1e it %d round %d m %d nviol %d nneg %d nloose %d ymax %.3e W:", (long long)b, (int)try_warm, try_tol, it_done, round, m, nviol, nneg, nloose, ymax);
                        for (int k = 0; k < m && k < WCAP; ++k) printf(" %d(%.2e)", Widx[k], yv[k]);
                        printf("\n");
                    }
#endif
                    // Rows of W off their bound with nothing left to correct: the working set holds rows that are nearly dependent
                    // AND not all active at the minimiser (neighbouring facets of the 854-row initial-state set), so its equalities
                    // are inconsistent at the 1e-9 level.  The loose row with the weakest multiplier leaves; if it belongs to the
                    // active set after all, the check of all rows brings it back.  (With wrong rows still in W looseness is
                    // expected: correct W first.)
                    // A handed-in working set is worth two or three corrections, not more: when it is far from the new active set the
                    // corrections add dependent rows, the multipliers explode and ten rounds cost more than the cold solve they
                    // were meant to save (seen right after reference steps).
                    if (try_warm && !(nviol == 0 && nneg == 0) && (round >= 2 || nviol + nneg > 6 || nloose != 0)) break;
                    if (nloose != 0 && nviol == 0 && nneg == 0) {
                        if (try_warm) break;
                        // Nearly parallel working rows that are BOTH active (neighbouring facets of the 854-row initial-state
                        // set): S is nearly singular along their difference and the proximal steps contract that component
                        // slowly although z no longer moves.  The set is right, so it gets up to two more rounds of steps
                        // (without the contraction-based stop) before a row has to leave -- dropping one only brings it back as
                        // a violated row, round after round (seen on 11 of 65536 packet-received instances at N = 20).
                        if (loose_retries < 2) { ++loose_retries; continue; }
                        const double ymin = wave_min(yloose);
#pragma unroll
                        for (int i = 0; i < RS; ++i)
                            if (inW[i] && yall[i] == ymin) { inW[i] = false; yall[i] = 0.0; }
                        continue;
                    }
                    if (nviol == 0 && nneg == 0) {
                        ok = true;
                        if (lane < NV) zv[lane] = zpv[lane];
                        // the certified working set goes back to the caller (next time step's warm start)
                        ws_m = m;
                        if (ws_out != nullptr && lane < m) ws_out[b * WS_STRIDE + 1 + lane] = Widx[lane];
                        wave_lds_fence();
                    }
                }
            }
            STAMP(7);
            if (ok) { st = TMPC_STATUS_OPTIMAL; break; }
            if (try_warm) { try_warm = false; continue; }          // the handed-in set did not certify: cold interior-point start
            if (try_tol <= 1e-12) { st = (rdn_last <= 1e-9 * qn) ? TMPC_STATUS_OPTIMAL : TMPC_STATUS_MAX_ITER; break; }
            try_tol *= 1e-2;
            if (saved) resume_it = it_done;        // (else the phase is re-run from its start: exact, slower, rare)
            saved = false;
#ifdef TMPC_ITERS_TOTAL
            ++n_reruns;
#endif
        }
        // (iteration cap or a stalled gap: the last iterate goes out under TMPC_STATUS_MAX_ITER -- the reference uses what an
        // inaccurate solve leaves in its variables, TubeTrackingMPC.py:185-192; INFEASIBLE is only ever declared with the
        // Farkas-type certificate of the interior-point phase or by the rows that depend on x_k alone)
#ifdef TMPC_DEBUG_PRINT
        if (lane == 0 && (st == TMPC_STATUS_MAX_ITER || st == TMPC_STATUS_NUMERICAL) && nx == 4)
            printf("NONOPT b %lld st %d it %d tol %.1e x %a %a %a %a r %a\n", (long long)b, st, it_done, try_tol, xin[0], xin[1], xin[2], xin[3], xin[4]);
#endif
        if (ws_out != nullptr && lane == 0) ws_out[b * WS_STRIDE] = ws_m;

        // ---------------------------------------------------------------- outputs
        TMPC_REFRESH();
        const bool good = st < TMPC_STATUS_INFEASIBLE;
        const double nanv = __longlong_as_double(0x7ff8000000000000ll);
        // z_full = [u | theta | x_0 ..] in LDS so that any lane can read any entry: Dv .* z, or -- when a further equality was
        // eliminated at set-up (terminal equality of the tracking MPC) -- Tzs z + Txf x_k
        wave_lds_fence();
        const double *zo = zv;
        if (qp.Tzs != nullptr) {
            double *zf = hw;                       // (the region of h is free here: nvf <= NV + nx entries)
            for (int i = lane; i < qp.nvf; i += WAVE) {
                double v = 0.0;
                for (int c = 0; c < nx; ++c) v += qp.Txf[i * nx + c] * xin[c];
                for (int j = 0; j < qp.nv; ++j) v += qp.Tzs[i * qp.nv + j] * zv[j];
                zf[i] = v;
            }
            zo = zf;
        } else if (lane < NV) {
            zv[lane] = (lane < qp.nv) ? qp.Dv[lane] * zv[lane] : 0.0;
        }
        wave_lds_fence();
        for (int i = lane; i < N * nu; i += WAVE) u_nom[b * N * nu + i] = good ? zo[i] : nanv;
        if (lane < nx + nu) {
            double v = 0.0;
            for (int j = 0; j < qp.nth; ++j) v += qp.Mth[lane * qp.nth + j] * zo[qp.off_theta + j];
            if (xu_ss) xu_ss[b * (nx + nu) + lane] = good ? v : nanv;
        }
        // x_nom: x_0 then the recursion x_{i+1} = A x_i + B u_i (reference :138)
        if (lane < nx) {
            const double x0 = (qp.off_x0 >= 0) ? zo[qp.off_x0 + lane] : xin[lane];
            if (x_nom0) x_nom0[b * nx + lane] = good ? x0 : nanv;
            tv[lane] = x0;
        }
        if (x_nom) {
            wave_lds_fence();
            if (lane < nx) x_nom[b * (N + 1) * nx + lane] = good ? tv[lane] : nanv;
            for (int i = 0; i < N; ++i) {
                double v = 0.0;
                if (lane < nx) {
                    for (int j = 0; j < nx; ++j) v += qp.A[lane * nx + j] * tv[j];
                    for (int j = 0; j < nu; ++j) v += qp.B[lane * nu + j] * zo[i * nu + j];
                }
                wave_lds_fence();
                if (lane < nx) { tv[lane] = v; x_nom[b * (N + 1) * nx + (i + 1) * nx + lane] = good ? v : nanv; }
                wave_lds_fence();
            }
        }
#ifdef TMPC_ITERS_TOTAL
        if (lane == 0) { status[b] = st; iters[b] = it_done + 100 * n_reruns + 10000 * n_rounds + 1000000 * amb_level; }   // diagnostic build: re-runs and refinement rounds folded in
#else
        if (lane == 0) { status[b] = st; iters[b] = it_done; }
#endif
        if (qp.ticks && lane == 0) qp.ticks[b] = static_cast<long long>(__builtin_amdgcn_s_memrealtime()) - t_begin;
#ifdef TMPC_STAMPS
        STAMP(8);
        if (b == 0 && lane == 0 && qp.dbg) { for (int p_ = 0; p_ < 16; ++p_) qp.dbg[p_] = tph[p_]; }
#endif
        wave_lds_fence();
        if constexpr (FUSED != 0) {
            // the trajectory's state machines for step t_mc: packet, losses, actuator, statistics, plant, estimator, next reference.
            // Their 128 doubles of hand-round space are the head of the (idle) transposition tile.
            TMPC_REFRESH();
            // (workgroup scope: writer and reader are lanes of the SAME wave, the CU's own L1 serves both -- the wait for the stores, no L2
            // write-back and no invalidate as at device scope)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // the solve's outputs, written lane by lane, are read across lanes
            McRecord mr;
            int t_now = t_mc, t_all = n_steps;
            uint8_t *gamma_out = nullptr;
            if constexpr (FUSED == 1) mr = mc_record(mc);
            else { mr = mc_record(mc.rec); t_now = mc.t; t_all = mr->T; gamma_out = mc.gamma_out; }
            const int t_next = t_now + 1 < t_all ? t_now + 1 : t_now;
            const double *rseq = mr->ref_seq;
            const bool alive = mcstep::mc_step_wave(mr->m, mr->st, t_now, t_all, b, rseq[t_now], rseq[t_next], u_nom, x_nom0, xu_ss, status,
                                                    iters, *reinterpret_cast<double (*)[mcstep::V_COUNT][mcstep::MAXN]>(red), lane, gamma_out);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // x_hat, ref_k of the next solve
            wave_lds_fence();
            if (!alive) break;                                          // (R-MPC: the trajectory ended on an infeasible solve)
        }
        }
    }
}

// tmpc_solve_batch and the per-step closed loop: a work item is one QP
template <int NV, int DP, int DS, int KC, int CP, int CS, int WPB>
__global__ __launch_bounds__(WAVE *WPB, 1) void solve_kernel(
    const DeviceQP qp, const int variant_id, const int64_t B,
    const double *__restrict__ x_k, const double *__restrict__ ref, const uint8_t *__restrict__ variant,
    double *__restrict__ u_nom, double *__restrict__ x_nom0, double *__restrict__ xu_ss,
    double *__restrict__ x_nom, int32_t *__restrict__ status, int32_t *__restrict__ iters,
    const int32_t *ws_in, int32_t *ws_out, unsigned long long *__restrict__ next_item) {
    solve_body<NV, DP, DS, KC, CP, CS, WPB, 0>(qp, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, ws_in, ws_out,
                                                   next_item, McNone{});
}
// the fused closed loop (tmpc_mc_run with one problem): a work item is one TRAJECTORY, all of its T steps
template <int NV, int DP, int DS, int KC, int CP, int CS, int WPB>
__global__ __launch_bounds__(WAVE *WPB, 1) void closed_loop_kernel(
    const DeviceQP qp, const int64_t B, double *__restrict__ u_nom, double *__restrict__ x_nom0, double *__restrict__ xu_ss,
    int32_t *__restrict__ status, int32_t *__restrict__ iters, int32_t *ws, unsigned long long *__restrict__ next_item,
    const McFused *mc) {
    solve_body<NV, DP, DS, KC, CP, CS, WPB, 1>(qp, 0, B, nullptr, nullptr, nullptr, u_nom, x_nom0, xu_ss, nullptr, status, iters, ws, ws,
                                               next_item, mc);
}
// the extended controller's closed loop: the QPs of problem `variant_id` at time step mc.t, each followed by its trajectory's state machines
template <int NV, int DP, int DS, int KC, int CP, int CS, int WPB>
__global__ __launch_bounds__(WAVE *WPB, 1) void closed_loop_step_kernel(
    const DeviceQP qp, const int variant_id, const int64_t B, const uint8_t *__restrict__ variant,
    double *__restrict__ u_nom, double *__restrict__ x_nom0, double *__restrict__ xu_ss,
    int32_t *__restrict__ status, int32_t *__restrict__ iters, int32_t *ws, unsigned long long *__restrict__ next_item,
    const McStepArg mc) {
    solve_body<NV, DP, DS, KC, CP, CS, WPB, 2>(qp, variant_id, B, nullptr, nullptr, variant, u_nom, x_nom0, xu_ss, nullptr, status, iters, ws, ws,
                                               next_item, mc);
}

// LDS of a workgroup: per-wave workspaces, the shared model, and `grows` (+ 1 zero) rows of the dense functionals
template <class SH>
constexpr size_t kernel_lds_bytes(int wpb, int grows) {
    return sizeof(double) * (static_cast<size_t>(SH::LDG) * (grows + ZERO_ROWS) + SH::KC * SH::NCCP + SH::KC * SH::NV + 2 * SH::NV * SH::LDH +
                             static_cast<size_t>(wpb) * WaveLds<SH>::TOTAL);
}
// rows of dense functionals the shapes are sized for: one slot in full, 96 of the 128 of two slots (cartpole N = 20: 92 / 93)
template <int DP, int DS>
constexpr int design_rows() { return (DP + DS) * WAVE <= 64 ? (DP + DS) * WAVE : ((DP + DS) * WAVE * 3) / 4; }
// Waves per workgroup = waves per CU (one persistent workgroup per CU).  Eight (two per SIMD: at most 256 registers each)
// for the small shapes, whose live set fits; four (one per SIMD, the accumulation-register file as spill space) otherwise,
// fewer if the LDS does not hold four workspaces.
template <int NV, int DP, int DS, int KC, int CP, int CS>
constexpr int waves_per_block() {
    constexpr int gr = design_rows<DP, DS>();
    // (twelve variables with four slots of single rows sit AT the 256-register cap of two waves per SIMD -- any change of the kernel tipped that
    // shape into scratch -- and take one wave per SIMD like the sixteen-variable shapes)
    if (NV <= 12 && !(NV > 8 && DS >= 4) && kernel_lds_bytes<Shape<NV, DP, DS, KC, CP, CS, tile_rows(NV, 8)>>(8, gr) <= 160 * 1024) return 8;
    using SH = Shape<NV, DP, DS, KC, CP, CS, tile_rows(NV, 4)>;
    return kernel_lds_bytes<SH>(4, gr) <= 160 * 1024 ? 4 : (kernel_lds_bytes<SH>(3, gr) <= 160 * 1024 ? 3 : 2);
}

#ifdef TMPC_HOST_SIM
// tests/wavesim: one workgroup of WPB waves on the host execution model; the persistent grid is that one workgroup
template <int NV, int DP, int DS, int KC, int CP, int CS, int WPB, int FUSED = 0>
hipError_t launch_wpb(const DeviceQP &qp, int variant_id, int64_t B, const double *x_k, const double *ref,
                      const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                      int32_t *status, int32_t *iters, const int32_t *ws_in, int32_t *ws_out, WorkCounter *wc, int n_cu,
                      hipStream_t stream, McArg<FUSED> mc = McArg<FUSED>{}) {
    using SH = Shape<NV, DP, DS, KC, CP, CS, tile_rows(NV, WPB)>;
    const size_t lds = kernel_lds_bytes<SH>(WPB, 4 * qp.nks);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    (void)n_cu; (void)stream; (void)wc;
    unsigned long long next_item = 0;
    sim::Dim3 bi, gd;
    bi.x = bi.y = bi.z = 0;                    // workgroup 0 of a grid of one
    sim_rendezvous_total += sim::run_block(WAVE * WPB, lds, bi, gd, [&]() {
        if constexpr (FUSED == 1)
            closed_loop_kernel<NV, DP, DS, KC, CP, CS, WPB>(qp, B, u_nom, x_nom0, xu_ss, status, iters, ws_out, &next_item, mc);
        else if constexpr (FUSED == 2)
            closed_loop_step_kernel<NV, DP, DS, KC, CP, CS, WPB>(qp, variant_id, B, variant, u_nom, x_nom0, xu_ss, status, iters, ws_out, &next_item, mc);
        else
            solve_kernel<NV, DP, DS, KC, CP, CS, WPB>(qp, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters,
                                                      ws_in, ws_out, &next_item);
    });
    return hipSuccess;
}
#else
template <int NV, int DP, int DS, int KC, int CP, int CS, int WPB, int FUSED = 0>
hipError_t launch_wpb(const DeviceQP &qp, int variant_id, int64_t B, const double *x_k, const double *ref,
                      const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                      int32_t *status, int32_t *iters, const int32_t *ws_in, int32_t *ws_out, WorkCounter *wc, int n_cu,
                      hipStream_t stream, McArg<FUSED> mc = McArg<FUSED>{}) {
    using SH = Shape<NV, DP, DS, KC, CP, CS, tile_rows(NV, WPB)>;
    const size_t lds = kernel_lds_bytes<SH>(WPB, 4 * qp.nks);
    if (lds > 160 * 1024) return hipErrorInvalidValue;       // (tmpc_api.cpp checks lds_bytes() before it accepts the wave path)
    static_assert(SH::RS <= 32, "validity mask is one 32-bit word per lane");
    // > 64 KiB of dynamic LDS needs the opt-in per device and instantiation.  The size depends on the problem (rows of dense
    // functionals staged), so the attribute is raised whenever a handle asks for more than any before it on this device
    // (handles may be driven from different host threads: the check and the call are one critical section).
    {
        static std::mutex attr_mutex;
        static size_t attr_lds[64] = {};
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        std::lock_guard<std::mutex> guard(attr_mutex);
        if (dev_id < 0 || dev_id >= 64 || attr_lds[dev_id] < lds) {
            const void *fn = nullptr;
            if constexpr (FUSED == 1) fn = reinterpret_cast<const void *>(&closed_loop_kernel<NV, DP, DS, KC, CP, CS, WPB>);
            else if constexpr (FUSED == 2) fn = reinterpret_cast<const void *>(&closed_loop_step_kernel<NV, DP, DS, KC, CP, CS, WPB>);
            else fn = reinterpret_cast<const void *>(&solve_kernel<NV, DP, DS, KC, CP, CS, WPB>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
            if (dev_id >= 0 && dev_id < 64) attr_lds[dev_id] = lds;
        }
    }
    int64_t blocks = B;                                      // (capped at one workgroup per CU below; the first items are dealt wave-major)
    // one persistent workgroup per CU (the LDS footprint admits no second one): the model is staged once and every wave
    // fetches its next instance as soon as it is done with the current one (grid-stride over the batch)
    const int64_t cap = static_cast<int64_t>(n_cu);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    // every launch gets a fresh (zero) word of the counter ring; the ring is cleared in one piece when it has gone round
    if (wc->pos >= wc->size) {
        hipError_t e0 = hipMemsetAsync(wc->ring, 0, sizeof(unsigned long long) * wc->size, stream);
        if (e0 != hipSuccess) return e0;
        wc->pos = 0;
    }
    if constexpr (FUSED == 1)
        hipLaunchKernelGGL((closed_loop_kernel<NV, DP, DS, KC, CP, CS, WPB>), dim3(static_cast<unsigned>(blocks)), dim3(WAVE * WPB), lds, stream,
                           qp, B, u_nom, x_nom0, xu_ss, status, iters, ws_out, wc->ring + wc->pos, mc);
    else if constexpr (FUSED == 2)
        hipLaunchKernelGGL((closed_loop_step_kernel<NV, DP, DS, KC, CP, CS, WPB>), dim3(static_cast<unsigned>(blocks)), dim3(WAVE * WPB), lds, stream,
                           qp, variant_id, B, variant, u_nom, x_nom0, xu_ss, status, iters, ws_out, wc->ring + wc->pos, mc);
    else
        hipLaunchKernelGGL((solve_kernel<NV, DP, DS, KC, CP, CS, WPB>), dim3(static_cast<unsigned>(blocks)), dim3(WAVE * WPB), lds, stream,
                           qp, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, ws_in, ws_out,
                           wc->ring + wc->pos);
    ++wc->pos;
    return hipGetLastError();
}

#endif

}  // namespace

#ifdef TMPC_HOST_SIM
unsigned long sim_rendezvous_count() { return sim_rendezvous_total; }
#endif

// Compiled shapes (NVP, DP, DS, KC, CP, CS): padded variables; 64-functional slots of dense paired / dense single rows; width
// of the factored block and its paired / single slots.  Dense-single-only shapes cover the small and the irregular problems
// (config 1); the paired + factored shapes cover the cartpole: base problem at N <= 11 (bench) and N <= 23 (the reference's
// N = 20; terminal block of 420 rows = 210 functionals of width 5), packet-received problem at N <= 11 and N <= 23
// (initial-state block Z (-) W of 854 rows = 427 functionals of width 4).
#if defined(TMPC_ONE_SHAPE)
// developer builds: one shape, e.g. -DTMPC_ONE_SHAPE="12,0,4,0,0,0" (register notes of a single instantiation in seconds)
#define TMPC_APPLY(X, ...) X(__VA_ARGS__)
#define TMPC_SHAPES(X) TMPC_APPLY(X, TMPC_ONE_SHAPE)
#elif defined(TMPC_ONLY_BENCH)
#define TMPC_SHAPES(X) X(11, 1, 0, 5, 4, 0)
#elif defined(TMPC_SIM_SHAPES_N20)
// tests/wavesim, developer builds: the two shapes of the cart-pole at the reference's horizon N = 20
#define TMPC_SHAPES(X) X(22, 2, 0, 5, 4, 0) X(26, 2, 0, 4, 7, 0)
#elif defined(TMPC_SIM_SHAPES_EXT)
// tests/wavesim: the two problems of the extended controller at N = 10 (the closed loop with a launch per problem and step)
#define TMPC_SHAPES(X) X(11, 1, 0, 5, 4, 0) X(15, 1, 0, 4, 7, 0)
#elif defined(TMPC_SIM_SHAPES)
// tests/wavesim: the bench shape (paired + factored functionals) and the shape of BASELINE config 1 (all rows dense and single)
#define TMPC_SHAPES(X) X(11, 1, 0, 5, 4, 0) X(8, 0, 2, 0, 0, 0)
#else
#define TMPC_SHAPES(X) \
    X(8, 0, 2, 0, 0, 0) X(8, 0, 4, 0, 0, 0) X(12, 0, 2, 0, 0, 0) X(12, 0, 4, 0, 0, 0) X(16, 0, 2, 0, 0, 0) X(16, 0, 4, 0, 0, 0) \
    X(11, 1, 0, 5, 4, 0) X(12, 1, 0, 5, 4, 0) X(22, 2, 0, 5, 4, 0) X(24, 2, 0, 5, 4, 0) X(15, 1, 0, 4, 7, 0) X(16, 1, 0, 4, 7, 0) X(26, 2, 0, 4, 7, 0)
#endif

#if defined(TMPC_HOST_SIM) || (defined(TMPC_FUSED_TU) && TMPC_FUSED_TU == 2)
// tmpc_fused_step.hip: the FUSED = 2 instantiations and nothing else
hipError_t launch_solve_mc_step(const DeviceQP &qp, const KernelShape &s, int variant_id, int64_t B, const uint8_t *variant, double *u_nom,
                                double *x_nom0, double *xu_ss, int32_t *status, int32_t *iters, int32_t *ws, const McFused *mc, int t,
                                uint8_t *gamma_out, WorkCounter *wc, int n_cu, hipStream_t stream) {
#if defined(TMPC_HOST_SIM) && !defined(TMPC_SIM_SHAPES_EXT)
    (void)qp; (void)s; (void)variant_id; (void)B; (void)variant; (void)u_nom; (void)x_nom0; (void)xu_ss; (void)status; (void)iters; (void)ws; (void)mc; (void)t;
    (void)gamma_out; (void)wc; (void)n_cu; (void)stream;
    return hipErrorInvalidValue;       // (tests/wavesim: only the binary of the extended controller instantiates closed_loop_step_kernel)
#else
    const McStepArg arg{mc, t, gamma_out};
#define TMPC_CASE(A, B_, C, D, E, F)                                                                                      \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F)                                    \
        return launch_wpb<A, B_, C, D, E, F, waves_per_block<A, B_, C, D, E, F>(), 2>(qp, variant_id, B, nullptr, nullptr, variant, u_nom, x_nom0, \
                                                                                      xu_ss, nullptr, status, iters, ws, ws, wc, n_cu, stream, arg);
    TMPC_SHAPES(TMPC_CASE)
#undef TMPC_CASE
    return hipErrorInvalidValue;
#endif
}
#endif
#if defined(TMPC_HOST_SIM) || (defined(TMPC_FUSED_TU) && TMPC_FUSED_TU == 1)
// tmpc_fused.hip: this translation unit holds the FUSED = 1 instantiations and nothing else (compiled next to the main one)
hipError_t launch_solve_mc(const DeviceQP &qp, const KernelShape &s, int64_t B, double *u_nom, double *x_nom0, double *xu_ss,
                           int32_t *status, int32_t *iters, int32_t *ws, const McFused *mc, WorkCounter *wc, int n_cu, hipStream_t stream) {
#if defined(TMPC_HOST_SIM) && defined(TMPC_SIM_SHAPES_EXT)
    return hipErrorInvalidValue;       // (tests/wavesim: the binary of the extended controller instantiates closed_loop_step_kernel only)
#else
#define TMPC_CASE(A, B_, C, D, E, F)                                                                                      \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F)                                    \
        return launch_wpb<A, B_, C, D, E, F, waves_per_block<A, B_, C, D, E, F>(), 1>(qp, 0, B, nullptr, nullptr, nullptr, u_nom, x_nom0, \
                                                                                         xu_ss, nullptr, status, iters, ws, ws, wc, n_cu, stream, mc);
#if defined(TMPC_HOST_SIM) && defined(TMPC_SIM_SHAPES)
    TMPC_CASE(11, 1, 0, 5, 4, 0)       // (tests/wavesim: the fused loop is run on the bench shape; one instantiation less per binary)
#else
    TMPC_SHAPES(TMPC_CASE)
#endif
#undef TMPC_CASE
    return hipErrorInvalidValue;
#endif
}
#endif
#if defined(TMPC_HOST_SIM) || !defined(TMPC_FUSED_TU)
int resident_waves(const KernelShape &s, int n_cu) {
#define TMPC_RW(A, B_, C, D, E, F) \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F) return n_cu * waves_per_block<A, B_, C, D, E, F>();
    TMPC_SHAPES(TMPC_RW)
#undef TMPC_RW
    return 0;
}

size_t lds_bytes(const KernelShape &s, int grows) {
#define TMPC_LDS(A, B_, C, D, E, F) \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F) { \
        constexpr int wpb = waves_per_block<A, B_, C, D, E, F>(); \
        return kernel_lds_bytes<Shape<A, B_, C, D, E, F, tile_rows(A, wpb)>>(wpb, grows); }
    TMPC_SHAPES(TMPC_LDS)
#undef TMPC_LDS
    return 0;
}

bool parks_in_lds(const KernelShape &s) {
#define TMPC_PARK(A, B_, C, D, E, F) \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F) \
        return WaveLds<Shape<A, B_, C, D, E, F, tile_rows(A, waves_per_block<A, B_, C, D, E, F>())>>::PARK_LDS;
    TMPC_SHAPES(TMPC_PARK)
#undef TMPC_PARK
    return false;
}

const char *kernel_name(const KernelShape &s) {
#define TMPC_NAME(A, B_, C, D, E, F) \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F) { \
        static const std::string n = std::string("tmpc::solve_kernel<") + #A "," #B_ "," #C "," #D "," #E "," #F "," + \
                                     std::to_string(waves_per_block<A, B_, C, D, E, F>()) + ">"; \
        return n.c_str(); }
    TMPC_SHAPES(TMPC_NAME)
#undef TMPC_NAME
    return "";
}

bool pick_config(int nv, int nd2, int nd1, int kc, int nc2, int nc1, KernelShape *shape) {
    static const int table[][6] = {
#define TMPC_ROW(A, B_, C, D, E, F) {A, B_, C, D, E, F},
        TMPC_SHAPES(TMPC_ROW)
#undef TMPC_ROW
    };
    long best = -1;
    for (const auto &t : table) {
        if (nv > t[0] || nd2 > t[1] * WAVE || nd1 > t[2] * WAVE || nc2 > t[4] * WAVE || nc1 > t[5] * WAVE) continue;
        if ((kc > 0) != (t[3] > 0) || kc > t[3]) continue;
        const long cost = static_cast<long>(t[0]) * t[0] * (t[1] + t[2]) + static_cast<long>(t[3]) * t[3] * (t[4] + t[5]) +
                          8L * (2 * t[1] + t[2] + 2 * t[4] + t[5]) + t[0];
        if (best < 0 || cost < best) {
            best = cost;
            shape->nvp = t[0]; shape->dp = t[1]; shape->ds = t[2]; shape->kcp = t[3]; shape->cp = t[4]; shape->cs = t[5];
        }
    }
    return best >= 0;
}

hipError_t launch_solve(const DeviceQP &qp, const KernelShape &s, int variant_id, int64_t B,
                        const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                        double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, const int32_t *ws_in, int32_t *ws_out,
                        WorkCounter *wc, int n_cu, hipStream_t stream) {
#if defined(TMPC_HOST_SIM) && defined(TMPC_SIM_SHAPES_EXT)
    return hipErrorInvalidValue;
#else
#define TMPC_CASE(A, B_, C, D, E, F)                                                                                      \
    if (s.nvp == A && s.dp == B_ && s.ds == C && s.kcp == D && s.cp == E && s.cs == F)                                    \
        return launch_wpb<A, B_, C, D, E, F, waves_per_block<A, B_, C, D, E, F>()>(qp, variant_id, B, x_k, ref, variant, u_nom, x_nom0, \
                                                                                   xu_ss, x_nom, status, iters, ws_in, ws_out, wc, n_cu, stream);
    TMPC_SHAPES(TMPC_CASE)
#undef TMPC_CASE
    return hipErrorInvalidValue;
#endif
}

#endif      // main translation unit

}  // namespace tmpc
