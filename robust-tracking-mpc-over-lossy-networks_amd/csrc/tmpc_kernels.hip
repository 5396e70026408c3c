// HIP kernels for gfx950 (MI355X): batched tube-tracking QP solve, one wavefront
// (64 lanes) per QP instance.
//
// What one wave does for its instance (x_k, ref) -- the device side of
// solve_optimization_problem (reference TubeTrackingMPC.py:170-194, whose arithmetic
// the reference delegates to cvxpy/Clarabel at :183):
//
//   1. q = F1s x_k + F2s ref,  h = g0s + Es x_k        (the open-loop prediction
//      x_i = A^i x_k + sum A^j B u_j is folded into F1s/Es by tmpc_condense.cpp)
//   2. z = -Hs^-1 q; if G z <= h the unconstrained minimiser is the answer
//   3. Mehrotra predictor-corrector interior-point iterations on
//         min 1/2 z'Hs z + q'z  s.t.  Gs z + s = h, s >= 0
//      rows of Gs are spread over the lanes (row r lives on lane r % 64, slot r / 64); (s, lambda) and the two
//      carried row quantities (r_p, ds_aff * dl_aff) stay in registers, h in LDS; Gs is staged once per workgroup
//      in LDS (column-major, so a lane-per-row read is conflict free); the rows of the terminal block are kept
//      factored, Gs_T = Hc * Psi with Hc only nx+nth wide (tmpc_condense.hpp), which cuts their share of every
//      sweep by nv/kc and of G'DG by (nv/kc)^2.  The wave-uniform vectors (z, the two directions, their
//      Psi-coordinates) live in LDS only and are read with broadcast loads.  The normal matrix M = Hs + Gs' D Gs
//      is accumulated per lane in registers (in one pass, or a few columns of its lower triangle at a time in the
//      register-lean build) and summed across the wave through an LDS transposition.  The nv x nv solve is
//      row-distributed: lane i holds row i of M, Gaussian elimination broadcasts the pivot row with v_readlane,
//      the multipliers stay in place for the corrector's second right-hand side.
//   4. active-set refinement on W = {lambda_i > s_i}: proximal Newton steps on the
//      KKT system of the equality-constrained QP (range-space form, S = G_W Hs^-1 G_W', factored and solved by the
//      same readlane elimination on register rows), accepted only when primal feasible on all rows with
//      non-negative multipliers
//   5. outputs: u_nom, x_nom[0], (x_bar, u_bar) = Mth theta, optionally x_nom
//
// Numerics are float64 throughout: cond(Hs) ~ 3e5 after scaling and the weights
// span 1e-1 .. 5e6 (R vs 10 P), float32 cannot resolve the minimiser.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>
#include <utility>

#include "tmpc_device.hpp"

namespace tmpc {

namespace {

constexpr int WAVE = 64;
constexpr int WCAP = 24;          // max rows in the refinement's working set
constexpr int RED_ROWS_MAX = 16;  // entries per transposition round (12 in the two-waves-per-SIMD build: smaller tile)
constexpr int RED_STRIDE = 68;    // 64 lanes + a pad after every 16: conflict-free transposed reads

// Diagnostic build only (-DTMPC_STAMPS): per-phase cycle counts of the first wave, written to
// qp.dbg.  Never compiled into the shipped library; stamps fence the LDS queue and distort timing.
#ifdef TMPC_STAMPS
#define STAMP(p) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long now_ = __builtin_amdgcn_s_memtime(); \
                      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tph[p] += now_ - tlast; tlast = now_; } while (0)
#else
#define STAMP(p) do { } while (0)
#endif

// Compiler-only barrier between two row iterations of a sweep: without it the loads of ALL rows
// are hoisted to the top of the unrolled loop (24 VGPRs per row) and the kernel spills to scratch.
__device__ __forceinline__ void row_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void wave_lds_fence() {
    // Orders this wave's LDS traffic.  The hardware executes the DS instructions of one wave in
    // issue order, so a read issued after a write (by any lane of the wave) observes it; what has to
    // be stopped is the COMPILER moving LDS accesses across this point.  No s_waitcnt: the waits for
    // returned data are inserted where the data is used.  (Waves of a block run different QPs, so a
    // block-wide barrier is neither available nor needed here.)
    asm volatile("" ::: "memory");
}

// 1/x to full double precision for normal, finite x: v_rcp_f64 + two Newton steps, without
// the scale/fixup sequence of an IEEE division (s, lambda are positive and well inside range)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// ---- cross-lane reductions without LDS: DPP inside a row of 16 lanes, readlane across rows
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
struct OpSum { __device__ __forceinline__ static double f(double a, double b) { return a + b; } };
struct OpMin { __device__ __forceinline__ static double f(double a, double b) { return fmin(a, b); } };
struct OpMax { __device__ __forceinline__ static double f(double a, double b) { return fmax(a, b); } };
template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
    v = Op::f(v, dpp_mov_d<0xB1>(v));    // quad_perm [1,0,3,2]
    v = Op::f(v, dpp_mov_d<0x4E>(v));    // quad_perm [2,3,0,1]
    v = Op::f(v, dpp_mov_d<0x141>(v));   // row_half_mirror
    v = Op::f(v, dpp_mov_d<0x140>(v));   // row_mirror: every lane of a 16-lane row holds the row's value
    const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    return Op::f(Op::f(r0, r1), Op::f(r2, r3));
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<OpMin>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<OpMax>(v); }

// Sum each of acc[0..CNT) over the 64 lanes and leave the totals in out[0..CNT) (LDS).
// Round: 16 entries are written as rows of a [16][RED_STRIDE] LDS tile (lane l at column
// l + l/16, i.e. one pad after every 16 lanes); lane l then adds the 16-lane quarter (l & 3) of
// entry (l >> 2) -- conflict free for RED_STRIDE = 68 -- and the four quarters, which sit in one
// quad, meet through two DPP quad permutes (no LDS round trip).
template <int CNT, int RR = RED_ROWS_MAX>
__device__ __forceinline__ void wave_reduce_to_lds(const double (&acc)[CNT], double *red, double *out, int lane) {
    const int e = lane >> 2, qd = lane & 3;
    const int er = (RR < 16 && e >= RR) ? 0 : e;          // RR < 16: the lanes of entries RR..15 idle along
    const int wcol = lane + (lane >> 4);
#pragma unroll
    for (int c0 = 0; c0 < CNT; c0 += RR) {
#pragma unroll
        for (int k = 0; k < RR; ++k)
            if (c0 + k < CNT) red[k * RED_STRIDE + wcol] = acc[c0 + k];
        wave_lds_fence();
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[er * RED_STRIDE + qd * 17 + j];
        t += dpp_mov_d<0xB1>(t);
        t += dpp_mov_d<0x4E>(t);
        if (qd == 0 && e < RR && c0 + e < CNT) out[c0 + e] = t;
        wave_lds_fence();
    }
}

// ---- packed lower triangle, column-major: (i,j), i >= j, at col_off(j) + i - j
template <int NV>
__host__ __device__ constexpr int col_off(int j) { return j * NV - j * (j - 1) / 2; }

// Column blocks of the lower triangle accumulated per sweep.  NV <= 12: one block (<= 90 accumulators fit the 512
// registers of a lone wave and save re-reading the rows; measured 3 % on the bench shape); larger NV: <= ~40 each.
template <int NV> struct Blocks;
template <> struct Blocks<8>  { static constexpr int n = 1; static constexpr int b[3] = {0, 8, 8}; };
template <> struct Blocks<12> { static constexpr int n = 1; static constexpr int b[4] = {0, 12, 12, 12}; };
template <> struct Blocks<16> { static constexpr int n = 5; static constexpr int b[6] = {0, 2, 4, 7, 11, 16}; };
template <> struct Blocks<24> { static constexpr int n = 10; static constexpr int b[11] = {0, 1, 3, 5, 7, 9, 11, 14, 17, 20, 24}; };
// ---- nv x nv solve, rows distributed over lanes.
// Lane i (< NV) holds row i of the symmetric positive definite M in registers.  Gaussian
// elimination without pivoting (= LDL'): at step k the pivot row is broadcast with v_readlane
// (wave-uniform SGPR operands), every lane below eliminates its entry and keeps the multiplier
// in its place.  24 VGPRs for NV = 12 instead of the 156 a per-lane copy of the factor needs;
// no LDS traffic, no waits.  `b` is carried along as an extra column.
template <int NV>
__device__ __forceinline__ bool rows_factor(double (&row)[NV], double &b, double &dinv, int lane) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double pkk = readlane_d(row[k], k);
        ok = ok && (pkk > 0.0);
        const double pinv = fast_rcp(pkk);
        const double f = (lane > k) ? row[k] * pinv : 0.0;
#pragma unroll
        for (int j = k + 1; j < NV; ++j) row[j] = fma(-f, readlane_d(row[j], k), row[j]);
        b = fma(-f, readlane_d(b, k), b);
        if (lane > k) row[k] = f;
        if (lane == k) dinv = pinv;
    }
    return ok;
}
// forward elimination of a further right-hand side with the stored multipliers
template <int NV>
__device__ __forceinline__ void rows_forward(const double (&row)[NV], double &b, int lane) {
#pragma unroll
    for (int k = 0; k < NV - 1; ++k) {
        const double f = (lane > k) ? row[k] : 0.0;
        b = fma(-f, readlane_d(b, k), b);
    }
}
// back substitution that leaves x_i on lane i (instead of wave-uniform copies): the caller stores it to LDS,
// from where the row sweeps read it with broadcast loads -- no 64-lane register copies of dz
template <int NV>
__device__ __forceinline__ double rows_backsub_lane(const double (&row)[NV], double b, double dinv, int lane) {
    double xl = 0.0;
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) {
        const double bi = b * dinv;
        const double xi = readlane_d(bi, i);
        xl = (lane == i) ? bi : xl;
        b = fma(-row[i], xi, b);
    }
    return xl;
}

// compile-time description of one kernel instantiation
// DIET: the build for two waves per SIMD (eight waves per workgroup): at most 256 registers and ~14 KB of LDS per wave,
// paid for with a smaller transposition tile and a rolled normal-matrix assembly
template <int NV_, int RD_, int KC_, int RC_, bool DIET_ = false>
struct Shape {
    static constexpr int NV = NV_, RD = RD_, KC = KC_, RC = RC_;
    static constexpr bool DIET = DIET_;
    static constexpr int RR = DIET_ ? 12 : RED_ROWS_MAX;
    static constexpr int KCA = KC_ > 0 ? KC_ : 1;            // array extents must be positive
    static constexpr int RT = RD_ + RC_;                     // 64-row slots per lane
    static constexpr int NDP = RD_ * WAVE, NCCP = RC_ * WAVE;
    static constexpr int NT = NV_ * (NV_ + 1) / 2, KT = KC_ * (KC_ + 1) / 2;
};

// per-wave LDS workspace (doubles), see solve_kernel
template <class SH>
struct WaveLds {
    static constexpr int RED = SH::RR * RED_STRIDE;                                     // transposition tile
    static constexpr int POL = 2 * SH::NV * WCAP + 4 * WCAP;                             // G_W, T, y, dy, W(idx)
    static constexpr int BIG = RED > POL ? RED : POL;                                   // never live together
    static constexpr int SUMS = SH::NT + 2 * SH::NV + 8;                                // dense totals of a sweep
    static constexpr int CSUMS = SH::KT + 2 * SH::KC + 8;                               // factored-block totals
    static constexpr int PMAT = SH::KCA * SH::NV;                                       // W * Psi
    static constexpr int HROW = SH::RT * WAVE;                                          // right-hand side h, [slot][lane]
    static constexpr int VEC = 9 * SH::NV + 32;                                         // q, z, cost gradient, x_k, ref, scratch, dz_aff, dz, Psi-coordinates
    static constexpr int TOTAL = BIG + SUMS + CSUMS + PMAT + HROW + VEC;
};

// c = Psi v (KC values) for a wave-uniform v: lane a < KC forms entry a from its row of Psi, the entries are
// then broadcast with v_readlane, so c lives in scalar registers (one dot product per wave, not one per lane)
template <class SH>
__device__ __forceinline__ void factor_coords(const double *Psi, const double (&v)[SH::NV], double (&c)[SH::KCA], int lane) {
    if constexpr (SH::KC > 0) {
        const int a_ = lane < SH::KC ? lane : 0;
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < SH::NV; ++j) t += Psi[a_ * SH::NV + j] * v[j];
#pragma unroll
        for (int a = 0; a < SH::KC; ++a) c[a] = readlane_d(t, a);
    }
}

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

// G_row . v for the row that lives in (slot k, this lane): dense rows read their row of Gt,
// factored rows read their kc-wide left factor and use c = Psi v.
template <class SH, int K>
__device__ __forceinline__ double row_dot(const double *Gt, const double *Hct, const double (&v)[SH::NV],
                                          const double (&c)[SH::KCA], int lane) {
    double t = 0.0;
    if constexpr (K < SH::RD) {
        const int r = lane + K * WAVE;
#pragma unroll
        for (int j = 0; j < SH::NV; ++j) t += Gt[j * SH::NDP + r] * v[j];
    } else {
        const int rc = lane + (K - SH::RD) * WAVE;
#pragma unroll
        for (int a = 0; a < SH::KC; ++a) t += Hct[a * SH::NCCP + rc] * c[a];
    }
    return t;
}

template <class SH>
__device__ __forceinline__ bool slot_valid(int k, int lane, int nd, int ncc) {
    return k < SH::RD ? (lane + k * WAVE < nd) : (lane + (k - SH::RD) * WAVE < ncc);
}

// One sweep over the DENSE rows of this lane for columns [J0, J1) of the lower triangle of G'DG.
// The FIRST sweep loads whole rows anyway, so it also forms r_p and accumulates G'(d.r_p), G'lam,
// the gap and |r_p|_inf.  Nothing per-row is kept besides (s, lam): r_p and 1/s are recomputed by
// the later sweeps, which is cheaper than carrying them through the register file.
template <class SH, int J0, int J1, bool FIRST, bool CARRY>
__device__ __forceinline__ void sweep_a_dense(const double *Gt, const double *hw, const double *zv,
                                              const double (&s)[SH::RT], const double (&lam)[SH::RT], double (&rp)[SH::RT],
                                              double (&dd)[SH::RD > 0 ? SH::RD : 1],
                                              double &gap_l, double &rpn_l, double *red, double *sums, int lane, int nd) {
    constexpr int NV = SH::NV, NDP = SH::NDP, NT = SH::NT;
    constexpr int TRI = col_off<NV>(J1) - col_off<NV>(J0);
    constexpr int CNT = TRI + (FIRST ? NV : 0);
    constexpr int I0 = FIRST ? 0 : J0;        // first column this sweep has to load
    double acc[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] = 0.0;
#pragma unroll
    for (int k = 0; k < SH::RD; ++k) {
        const int r = lane + k * WAVE;
        double g[NV];
#pragma unroll
        for (int j = I0; j < NV; ++j) g[j] = Gt[j * NDP + r];
        // padding rows carry s = 1, lam = 0, g = 0, h = 1: d = 0 and r_p = 0 without any masking
        if (FIRST) dd[k] = lam[k] * fast_rcp(s[k]);
        const double d = dd[k];
        if (FIRST) {
            double rpk;
            if constexpr (CARRY) {
                rpk = rp[k];                     // carried over from the previous iteration: r_p <- (1 - alpha) r_p
            } else {
                double gz0 = 0.0, gz1 = 0.0;
#pragma unroll
                for (int j = 0; j + 1 < NV; j += 2) { gz0 += g[j] * zv[j]; gz1 += g[j + 1] * zv[j + 1]; }
                if (NV & 1) gz0 += g[NV - 1] * zv[NV - 1];
                rpk = (gz0 + gz1) + s[k] - hw[k * WAVE + lane];
                rp[k] = rpk;
            }
            gap_l += s[k] * lam[k];
            rpn_l = fmax(rpn_l, fabs(rpk));
            const double t = d * rpk;
#pragma unroll
            for (int i = 0; i < NV; ++i) acc[TRI + i] += g[i] * t;
        }
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const double dg = d * g[j];
#pragma unroll
            for (int i = j; i < NV; ++i) acc[col_off<NV>(j) - col_off<NV>(J0) + i - j] += dg * g[i];
        }
        row_fence();
    }
    if (FIRST) {
        // triangle block -> its packed position; G'(d.rp) -> behind the triangle
        double tri[TRI > 0 ? TRI : 1], vecs[NV];
#pragma unroll
        for (int i = 0; i < TRI; ++i) tri[i] = acc[i];
#pragma unroll
        for (int i = 0; i < NV; ++i) vecs[i] = acc[TRI + i];
        if constexpr (TRI > 0) wave_reduce_to_lds<TRI, SH::RR>(tri, red, sums + col_off<NV>(J0), lane);
        wave_reduce_to_lds<NV, SH::RR>(vecs, red, sums + NT, lane);
    } else {
        wave_reduce_to_lds<CNT, SH::RR>(acc, red, sums + col_off<NV>(J0), lane);
    }
}

template <class SH, int BI, bool CARRY>
__device__ __forceinline__ void sweep_a_dense_all(const double *Gt, const double *hw, const double *zv,
                                                  const double (&s)[SH::RT], const double (&lam)[SH::RT], double (&rp)[SH::RT],
                                                  double (&dd)[SH::RD > 0 ? SH::RD : 1],
                                                  double &gap_l, double &rpn_l, double *red, double *sums, int lane, int nd) {
    using BL = Blocks<SH::NV>;      // the same blocks in both builds: the accumulators are not what overflows the lean build's registers
    if constexpr (BI < BL::n) {
        sweep_a_dense<SH, BL::b[BI], BL::b[BI + 1], BI == 0, CARRY>(Gt, hw, zv, s, lam, rp, dd, gap_l, rpn_l, red, sums, lane, nd);
        sweep_a_dense_all<SH, BI + 1, CARRY>(Gt, hw, zv, s, lam, rp, dd, gap_l, rpn_l, red, sums, lane, nd);
    }
}

// The FACTORED rows: kc-wide left factor, so W = Hc' D Hc has only KT entries; one pass.
template <class SH, bool CARRY>
__device__ __forceinline__ void sweep_a_factored(const double *Hct, const double *hw, const double *czv,
                                                 const double (&s)[SH::RT], const double (&lam)[SH::RT], double (&rp)[SH::RT],
                                                 double &gap_l, double &rpn_l, double *red, double *csums, int lane, int ncc) {
    constexpr int KC = SH::KC, KT = SH::KT, NCCP = SH::NCCP;
    double acc[KT + KC];
#pragma unroll
    for (int i = 0; i < KT + KC; ++i) acc[i] = 0.0;
#pragma unroll
    for (int k = SH::RD; k < SH::RT; ++k) {
        const int rc = lane + (k - SH::RD) * WAVE;
        double hc[KC];
        double rpk;
        if constexpr (CARRY) {
#pragma unroll
            for (int a = 0; a < KC; ++a) hc[a] = Hct[a * NCCP + rc];
            rpk = rp[k];
        } else {
            double gz = 0.0;
#pragma unroll
            for (int a = 0; a < KC; ++a) { hc[a] = Hct[a * NCCP + rc]; gz += hc[a] * czv[a]; }
            rpk = gz + s[k] - hw[k * WAVE + lane];
            rp[k] = rpk;
        }
        const double rsk = fast_rcp(s[k]);          // padding rows: lam = 0, so d = 0
        gap_l += s[k] * lam[k];
        rpn_l = fmax(rpn_l, fabs(rpk));
        const double d = lam[k] * rsk;
        const double t = d * rpk;
#pragma unroll
        for (int a = 0; a < KC; ++a) {
            const double dg = d * hc[a];
#pragma unroll
            for (int b2 = a; b2 < KC; ++b2) acc[col_off<KC>(a) + b2 - a] += dg * hc[b2];
            acc[KT + a] += hc[a] * t;
        }
    }
    wave_reduce_to_lds<KT + KC, SH::RR>(acc, red, csums, lane);
}

template <int NV, int RD, int KC, int RC, bool WARM, int WPB>
__global__ __launch_bounds__(WAVE *WPB, 1) void solve_kernel(
    const DeviceQP qp, const WarmStart warm, const int variant_id, const int64_t B,
    const double *__restrict__ x_k, const double *__restrict__ ref, const uint8_t *__restrict__ variant,
    double *__restrict__ u_nom, double *__restrict__ x_nom0, double *__restrict__ xu_ss,
    double *__restrict__ x_nom, int32_t *__restrict__ status, int32_t *__restrict__ iters) {
    using SH = Shape<NV, RD, KC, RC, WPB == 8>;
    using WL = WaveLds<SH>;
    constexpr int RT = SH::RT, NDP = SH::NDP, NCCP = SH::NCCP, NT = SH::NT, KT = SH::KT, KCA = SH::KCA;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *Gt = smem;                         // [NV][NDP]
    double *Hct = Gt + NV * NDP;               // [KC][NCCP]
    double *Psi = Hct + KC * NCCP;             // [KC][NV]
    double *Hs = Psi + KC * NV;                // [NV][NV]
    double *Hinv = Hs + NV * NV;               // [NV][NV]
    double *wbase = Hinv + NV * NV;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid >> 6;
    const int nx = qp.nx, nu = qp.nu, N = qp.N, nd = qp.nd, ncc = qp.ncc, nc = qp.nc;

    // ---- stage the shared model once per workgroup (coalesced, L2-resident source)
    for (int i = tid; i < NV * NDP; i += blockDim.x) Gt[i] = qp.Gt[i];
    for (int i = tid; i < KC * NCCP; i += blockDim.x) Hct[i] = qp.Hct[i];
    for (int i = tid; i < KC * NV; i += blockDim.x) Psi[i] = qp.Psi[i];
    for (int i = tid; i < NV * NV; i += blockDim.x) { Hs[i] = qp.Hs[i]; Hinv[i] = qp.Hinv[i]; }
    __syncthreads();

    double *red = wbase + wave * WL::TOTAL;       // transposition tile / refinement workspace
    double *sums = red + WL::BIG;                 // dense totals: triangle (column-major packed), two vectors
    double *csums = sums + WL::SUMS;              // factored-block totals: W (packed), two kc-vectors
    double *Pm = csums + WL::CSUMS;               // [KC][NV] W * Psi
    double *hw = Pm + WL::PMAT;                   // h, [slot][lane]
    double *vec = hw + WL::HROW;
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

    const int64_t wave_global = static_cast<int64_t>(blockIdx.x) * WPB + wave;
    const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * WPB;

    for (int64_t b = wave_global; b < B; b += wave_stride) {
        if (variant != nullptr && variant[b] != variant_id) continue;
        if (variant == nullptr && variant_id != 0) continue;

        // ------------------------------------------------------------ per-instance data
        if (lane < nx) { xin[lane] = x_k[b * nx + lane]; xin[nx + lane] = ref[b * nx + lane]; }
        wave_lds_fence();
        int st = TMPC_STATUS_MAX_ITER;
        int it_done = 0;
#ifdef TMPC_STAMPS
        long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
        double hn = 1.0;
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            const int sl = k * WAVE + lane;
            double v = qp.g0p[sl];
            for (int c = 0; c < nx; ++c) v += qp.Esp[sl * nx + c] * xin[c];
            if (slot_valid<SH>(k, lane, nd, ncc)) hn = fmax(hn, fabs(v));
            hw[sl] = v;
        }
        hn = wave_max(hn);
        wave_lds_fence();
        // z = -Hinv q
        if (lane < NV) {
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) v -= Hinv[lane * NV + j] * qv[j];
            zv[lane] = v;
        }
        wave_lds_fence();
        double qn = 1.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) qn = fmax(qn, fabs(qv[j]));

        double s[RT], lam[RT];
        // the primal residual is carried from iteration to iteration instead of being formed from
        // G z in sweep A -- the Newton step gives r_p(z + a dz, s + a ds) = (1 - a) r_p exactly (ds = -r_p - G dz)
        constexpr bool CARRY = !WARM && NV <= 16;      // (the 24-variable shape keeps its long-tested code path)
        double rpc[CARRY ? RT : 1];
        double smin = INFINITY;
        {
            double z[NV], cz[KCA];
#pragma unroll
            for (int j = 0; j < NV; ++j) z[j] = zv[j];
            factor_coords<SH>(Psi, z, cz, lane);
            // unrolled by the template recursion below
            auto init_slot = [&](auto kc_) {
                constexpr int k = decltype(kc_)::value;
                const double gz = row_dot<SH, k>(Gt, Hct, z, cz, lane);
                s[k] = hw[k * WAVE + lane] - gz;
                lam[k] = 0.0;
                if (slot_valid<SH>(k, lane, nd, ncc)) smin = fmin(smin, s[k]);
            };
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (init_slot(std::integral_constant<int, Ks>{}), ...); }
            (std::make_integer_sequence<int, RT>{});
        }
        smin = wave_min(smin);
        STAMP(0);

        // warm mode: the streaming kernel has already iterated this instance (tmpc_stream.hip)
        const int st1 = WARM ? warm.stat[b] : -1;
        if (infeasible_par) {
            st = TMPC_STATUS_INFEASIBLE;
        } else if (st1 == TMPC_STATUS_INFEASIBLE || st1 == TMPC_STATUS_NUMERICAL) {
            st = st1;
            it_done = warm.it[b];
        } else if (st1 == 4 || (st1 < 0 && smin >= 0.0)) {
            st = TMPC_STATUS_OPTIMAL;
        } else {
            // -------------------------------------------------------- interior point
            {
                const double viol = -smin;
                const double fl = 0.1 * fmax(viol, 1.0);
#pragma unroll
                for (int k = 0; k < RT; ++k) {
                    const bool valid = slot_valid<SH>(k, lane, nd, ncc);
                    const double raw = s[k];
                    s[k] = valid ? fmax(raw, fl) : 1.0;
                    lam[k] = valid ? 1.0 : 0.0;
                    if constexpr (CARRY) rpc[k] = valid ? s[k] - raw : 0.0;     // r_p = G z + s - h with h - G z = raw
                }
            }
            double try_tol = qp.tol;
            const double ncd = static_cast<double>(nc);
            int it = 0;
            double rdn_last = 0.0;
            bool skip_ipm = false;
            if (WARM && (st1 == 0 || st1 == 1)) {
                // pick up (z, s, lambda) where the streaming kernel left them
                if (lane < NV) zv[lane] = warm.z[b * NV + lane];
#pragma unroll
                for (int k = 0; k < RT; ++k) {
                    const int gid = k < RD ? lane + k * WAVE : nd + lane + (k - RD) * WAVE;
                    const bool valid = slot_valid<SH>(k, lane, nd, ncc);
                    s[k] = valid ? warm.s[b * warm.ncp + gid] : 1.0;
                    lam[k] = valid ? warm.lam[b * warm.ncp + gid] : 0.0;
                }
                wave_lds_fence();
                it = warm.it[b];
                it_done = it;
                skip_ipm = true;
                rdn_last = INFINITY;
            }
            // interior point until the active set can be read off, then the refinement; the pair is
            // repeated (with a tighter hand-over tolerance) only if the refinement cannot certify its set
            for (;;) {
            bool want_polish = false;
            if (WARM && skip_ipm) {
                want_polish = (st1 == 0);       // hand-over point reached: refine; iteration cap: fall through
                skip_ipm = false;
            } else
            for (; it < qp.max_iter; ++it) {
                it_done = it;
                // ---- sweeps A: residuals, G'DG (dense rows by column blocks, factored rows as W), G'(d.rp), G'lam.
                // r_p is kept per row for sweeps B and D; the iterate z and the directions live in LDS only.
                double gap_l = 0.0, rpn_l = 0.0;
                double rp[RT];
                if constexpr (CARRY) {
#pragma unroll
                    for (int k = 0; k < RT; ++k) rp[k] = rpc[k];
                }
                {
                    double dd[RD > 0 ? RD : 1];
                    if constexpr (RD > 0) sweep_a_dense_all<SH, 0, CARRY>(Gt, hw, zv, s, lam, rp, dd, gap_l, rpn_l, red, sums, lane, nd);
                }
                if constexpr (KC > 0) {
                    if constexpr (!CARRY) {
                        coords_lds<SH>(Psi, zv, czv, lane);
                        wave_lds_fence();
                    }
                    sweep_a_factored<SH, CARRY>(Hct, hw, czv, s, lam, rp, gap_l, rpn_l, red, csums, lane, ncc);
                    // fold the factored block into the dense totals: P = W Psi now, Psi' P when the rows are loaded
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
                        v1 = (RD > 0) ? sums[NT + lane] : 0.0;
#pragma unroll
                        for (int a = 0; a < KC; ++a) v1 += Psi[a * NV + lane] * csums[KT + a];
                    }
                    wave_lds_fence();
                    if (lane < NV) sums[NT + lane] = v1;
                    wave_lds_fence();
                }
                double lmax = 0.0;
#pragma unroll
                for (int k = 0; k < RT; ++k) lmax = fmax(lmax, lam[k]);
                const double rpn = wave_max(rpn_l);
                lmax = wave_max(lmax);
                const double gap = wave_sum(gap_l);
                const double mu = gap / ncd;
                STAMP(1);
                // cost gradient cg = Hs z + q (lane i computes entry i)
                if (lane < NV) {
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) v += Hs[lane * NV + j] * zv[j];
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
                // The dual residual needs G'lam, which nothing else uses: it is formed (one more pass over the rows) only when
                // the primal residual and the gap already pass, or when the multipliers blow up (Farkas test).
                const bool near = (rpn <= try_tol * hn) && (gap <= try_tol * objs);
                if (near || lmax > 1e10) {
                    if constexpr (RD > 0) {
                        double accl[NV];
#pragma unroll
                        for (int i = 0; i < NV; ++i) accl[i] = 0.0;
#pragma unroll
                        for (int k = 0; k < RD; ++k) {
                            const int r = lane + k * WAVE;
#pragma unroll
                            for (int j = 0; j < NV; ++j) accl[j] += Gt[j * NDP + r] * lam[k];
                            row_fence();
                        }
                        wave_reduce_to_lds<NV, SH::RR>(accl, red, sums + NT + NV, lane);
                    }
                    if constexpr (KC > 0) {
                        double accl[KC];
#pragma unroll
                        for (int a = 0; a < KC; ++a) accl[a] = 0.0;
#pragma unroll
                        for (int k = RD; k < RT; ++k) {
                            const int rc = lane + (k - RD) * WAVE;
#pragma unroll
                            for (int a = 0; a < KC; ++a) accl[a] += Hct[a * NCCP + rc] * lam[k];
                        }
                        wave_reduce_to_lds<KC, SH::RR>(accl, red, csums + KT + KC, lane);
                        double gl = 0.0;
                        if (lane < NV) {
                            gl = (RD > 0) ? sums[NT + NV + lane] : 0.0;
#pragma unroll
                            for (int a = 0; a < KC; ++a) gl += Psi[a * NV + lane] * csums[KT + KC + a];
                        }
                        wave_lds_fence();
                        if (lane < NV) sums[NT + NV + lane] = gl;
                        wave_lds_fence();
                    }
                    double rdn = 0.0, gn = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        const double glj = sums[NT + NV + j];
                        rdn = fmax(rdn, fabs(cgv[j] + glj));
                        gn = fmax(gn, fabs(glj));
                    }
                    if (!(rdn == rdn)) { st = TMPC_STATUS_NUMERICAL; break; }
                    if (near && rdn <= 1e3 * try_tol * qn) { want_polish = true; rdn_last = rdn; break; }
                    if (lmax > 1e10) {
                        // Farkas-type certificate: lam blows up, G'lam -> 0, h'lam < 0
                        double hl = 0.0;
#pragma unroll
                        for (int k = 0; k < RT; ++k) hl += hw[k * WAVE + lane] * lam[k];
                        hl = wave_sum(hl);
                        if (hl < 0.0 && gn <= 1e-6 * lmax) { st = TMPC_STATUS_INFEASIBLE; break; }
                    }
                }
                if (gap <= 1e-15 * objs) { st = TMPC_STATUS_MAX_ITER; break; }
                // ---- M = Hs + G'DG by rows (lane i holds row i), elimination with the predictor rhs carried along
                double mrow[NV], mdinv = 1.0, rhs_i = 0.0;
                {
                    const int li = lane < NV ? lane : 0;
                    double shift = 0.0;
                    bool spd = false;
                    for (int attempt = 0; attempt < 2 && !spd; ++attempt) {
                        if constexpr (SH::DIET) {
                            // register-lean build: row li of M is formed in a ROLLED loop into LDS (the transposition tile is idle
                            // here) and read back.  Unrolled, the 14 loads per entry are all hoisted to the top of the block and
                            // push the live set far past 256 registers (139 instead of 228 spilled registers with this alone).
                            double *mt = red + li * (NV + 1);
#pragma unroll 1
                            for (int j = 0; j < NV; ++j) {
                                const int lo = li < j ? li : j, hi2 = li < j ? j : li;
                                double v = Hs[li * NV + j] + (li == j ? shift : 0.0);
                                if constexpr (RD > 0) v += sums[lo * NV - lo * (lo - 1) / 2 + hi2 - lo];
                                if constexpr (KC > 0) {
#pragma unroll
                                    for (int a = 0; a < KC; ++a) v += Psi[a * NV + li] * Pm[a * NV + j];
                                }
                                if (lane < NV) mt[j] = v;
                            }
                            wave_lds_fence();
#pragma unroll
                            for (int j = 0; j < NV; ++j) mrow[j] = (lane < NV) ? mt[j] : 0.0;
                            wave_lds_fence();
                        } else {
#pragma unroll
                            for (int j = 0; j < NV; ++j) {
                                const int lo = li < j ? li : j, hi2 = li < j ? j : li;
                                double v = Hs[li * NV + j] + (li == j ? shift : 0.0);
                                if constexpr (RD > 0) v += sums[lo * NV - lo * (lo - 1) / 2 + hi2 - lo];
                                if constexpr (KC > 0) {
    #pragma unroll
                                    for (int a = 0; a < KC; ++a) v += Psi[a * NV + li] * Pm[a * NV + j];
                                }
                                mrow[j] = (lane < NV) ? v : 0.0;
                            }
                        }
                        rhs_i = (lane < NV) ? -cgv[li] - sums[NT + li] : 0.0;
                        double bb = rhs_i;
                        mdinv = 1.0;
                        spd = rows_factor<NV>(mrow, bb, mdinv, lane);
                        if (spd) {
                            const double xl = rows_backsub_lane<NV>(mrow, bb, mdinv, lane);
                            if (lane < NV) dzav[lane] = xl;
                        } else {
                            // non-positive pivot from cancellation: retry once with a 1e-13 * trace shift
                            double trc = 0.0;
#pragma unroll
                            for (int i = 0; i < NV; ++i) trc += Hs[i * NV + i];
                            shift = 1e-13 * (trc + lmax);
                        }
                    }
                    if (!spd) { st = TMPC_STATUS_NUMERICAL; break; }
                }
                wave_lds_fence();
                if constexpr (KC > 0) { coords_lds<SH>(Psi, dzav, cdzav, lane); wave_lds_fence(); }
                STAMP(3);
                // ---- sweep B: affine step statistics and the corrector's G' products; w = ds_aff * dl_aff kept per row
                double rho_aff = 0.0, sb1 = 0.0, sb2 = 0.0;
                double wv[RT];
                double v2 = 0.0, v3 = 0.0;          // lane i < NV: entries i of G'(dsa.dla/s) and G'(1/s)
                {
                    // returns (dsa*dla/s, 1/s) of the row; padding rows (s = 1, lam = 0, g = 0, r_p = 0) give zeros by themselves
                    auto row_stats = [&](int k, bool valid, double gdz, double &c1, double &rsk) {
                        const double rs0 = fast_rcp(s[k]);
                        const double dsa = -rp[k] - gdz;
                        const double dla = -lam[k] - lam[k] * rs0 * dsa;
                        const double rl = valid ? fast_rcp(lam[k]) : 0.0;
                        rho_aff = fmax(rho_aff, fmax(-dsa * rs0, -dla * rl));
                        const double w = dsa * dla;
                        sb1 += s[k] * dla + lam[k] * dsa;
                        sb2 += w;
                        wv[k] = w;
                        c1 = w * rs0;
                        rsk = valid ? rs0 : 0.0;
                    };
                    if constexpr (RD > 0) {
                        double accb[2 * NV];
#pragma unroll
                        for (int i = 0; i < 2 * NV; ++i) accb[i] = 0.0;
#pragma unroll
                        for (int k = 0; k < RD; ++k) {
                            const int r = lane + k * WAVE;
                            double g[NV];
                            double gd0 = 0.0, gd1 = 0.0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) g[j] = Gt[j * NDP + r];
#pragma unroll
                            for (int j = 0; j + 1 < NV; j += 2) { gd0 += g[j] * dzav[j]; gd1 += g[j + 1] * dzav[j + 1]; }
                            if (NV & 1) gd0 += g[NV - 1] * dzav[NV - 1];
                            double c1, rsk;
                            row_stats(k, r < nd, gd0 + gd1, c1, rsk);
#pragma unroll
                            for (int j = 0; j < NV; ++j) { accb[j] += g[j] * c1; accb[NV + j] += g[j] * rsk; }
                            row_fence();
                        }
                        wave_reduce_to_lds<2 * NV, SH::RR>(accb, red, sums + NT, lane);   // overwrites G'(d.rp), G'lam (consumed)
                    }
                    if constexpr (KC > 0) {
                        double accc[2 * KC];
#pragma unroll
                        for (int i = 0; i < 2 * KC; ++i) accc[i] = 0.0;
#pragma unroll
                        for (int k = RD; k < RT; ++k) {
                            const int rc = lane + (k - RD) * WAVE;
                            double hc[KC];
                            double gdz = 0.0;
#pragma unroll
                            for (int a = 0; a < KC; ++a) { hc[a] = Hct[a * NCCP + rc]; gdz += hc[a] * cdzav[a]; }
                            double c1, rsk;
                            row_stats(k, rc < ncc, gdz, c1, rsk);
#pragma unroll
                            for (int a = 0; a < KC; ++a) { accc[a] += hc[a] * c1; accc[KC + a] += hc[a] * rsk; }
                            row_fence();
                        }
                        wave_reduce_to_lds<2 * KC, SH::RR>(accc, red, csums + KT, lane);
                    }
                    rho_aff = wave_max(rho_aff);
                    sb1 = wave_sum(sb1);
                    sb2 = wave_sum(sb2);
                    if (lane < NV) {
                        if constexpr (RD > 0) { v2 = sums[NT + lane]; v3 = sums[NT + NV + lane]; }
                        if constexpr (KC > 0) {
#pragma unroll
                            for (int a = 0; a < KC; ++a) { v2 += Psi[a * NV + lane] * csums[KT + a]; v3 += Psi[a * NV + lane] * csums[KT + KC + a]; }
                        }
                    }
                }
                STAMP(4);
                const double aaff = rho_aff > 1.0 ? 1.0 / rho_aff : 1.0;
                const double mu_aff = (gap + aaff * sb1 + aaff * aaff * sb2) / ncd;
                double sigma = mu_aff / mu;
                sigma = fmin(sigma * sigma * sigma, 1.0);
                const double smu = sigma * mu;
                {
                    double bb = (lane < NV) ? rhs_i + v2 - smu * v3 : 0.0;
                    rows_forward<NV>(mrow, bb, lane);
                    const double xl = rows_backsub_lane<NV>(mrow, bb, mdinv, lane);
                    if (lane < NV) dzv[lane] = xl;
                }
                wave_lds_fence();
                if constexpr (KC > 0) { coords_lds<SH>(Psi, dzv, cdzv, lane); wave_lds_fence(); }
                STAMP(5);
                // ---- sweep D: final direction, step length, update
                double dsv[RT], dlv[RT];
                double om = (1.0 - aaff) * (1.0 - aaff);
                om = fmin(fmax(om, 1e-4), 1e-2);
                const double tau = 1.0 - om;
                double rho = 0.0;
                {
                    auto step_row = [&](auto kc_) {
                        constexpr int k = decltype(kc_)::value;
                        const bool valid = slot_valid<SH>(k, lane, nd, ncc);
                        double gdz = 0.0;
                        if constexpr (k < RD) {
                            const int r = lane + k * WAVE;
                            double g0 = 0.0, g1 = 0.0;
#pragma unroll
                            for (int j = 0; j + 1 < NV; j += 2) { g0 += Gt[j * NDP + r] * dzv[j]; g1 += Gt[(j + 1) * NDP + r] * dzv[j + 1]; }
                            if (NV & 1) g0 += Gt[(NV - 1) * NDP + r] * dzv[NV - 1];
                            gdz = g0 + g1;
                        } else {
                            const int rc = lane + (k - RD) * WAVE;
#pragma unroll
                            for (int a = 0; a < KC; ++a) gdz += Hct[a * NCCP + rc] * cdzv[a];
                        }
                        const double rs0 = fast_rcp(s[k]);
                        const double dsk = -rp[k] - gdz;
                        const double rc2 = s[k] * lam[k] + wv[k] - smu;
                        const double dlk = valid ? (-(rc2 + lam[k] * dsk) * rs0) : 0.0;
                        const double rl = valid ? fast_rcp(lam[k]) : 0.0;
                        rho = fmax(rho, fmax(-dsk * rs0, -dlk * rl));
                        dsv[k] = dsk;
                        dlv[k] = dlk;
                        row_fence();
                    };
                    [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (step_row(std::integral_constant<int, Ks>{}), ...); }
                    (std::make_integer_sequence<int, RT>{});
                }
                rho = wave_max(rho);
                const double alpha = rho > tau ? tau / rho : 1.0;
#pragma unroll
                for (int k = 0; k < RT; ++k) {
                    s[k] += alpha * dsv[k];
                    lam[k] += alpha * dlv[k];
                    if constexpr (CARRY) rpc[k] = (1.0 - alpha) * rp[k];
                }
                if (lane < NV) zv[lane] += alpha * dzv[lane];
                wave_lds_fence();
                it_done = it + 1;
                STAMP(6);
            }
            if (!want_polish) break;
            // ------------------------------------------------ active-set refinement
            bool ok = false;
            {
                // workspace carved from the (now idle) transposition tile
                double *GW = red;                         // [WCAP][NV]  rows of the working set, expanded
                double *T = GW + WCAP * NV;               // [NV][WCAP]
                double *yv = T + NV * WCAP;               // [WCAP]
                double *dyv = yv + WCAP;                  // [WCAP]
                int *Widx = reinterpret_cast<int *>(dyv + WCAP);   // [WCAP] global row ids
                bool inW[RT];
                double yall[RT];
#pragma unroll
                for (int k = 0; k < RT; ++k) { inW[k] = slot_valid<SH>(k, lane, nd, ncc) && (lam[k] > s[k]); yall[k] = lam[k]; }
                double zp[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) zp[j] = zv[j];
                for (int round = 0; round < 6 && !ok; ++round) {
                    // compact the working set: W[0..m)
                    int m = 0;
#pragma unroll
                    for (int k = 0; k < RT; ++k) {
                        const unsigned long long bal = __ballot(inW[k]);
                        const int pos = m + __popcll(bal & ((1ull << lane) - 1ull));
                        const int gid = k < RD ? lane + k * WAVE : nd + lane + (k - RD) * WAVE;
                        if (inW[k] && pos < WCAP) { Widx[pos] = gid; yv[pos] = yall[k]; }
                        m += __popcll(bal);
                    }
                    wave_lds_fence();
                    if (m > WCAP) break;
                    if (m == 0) {
                        if (lane < NV) {
                            double v = 0.0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) v -= Hinv[lane * NV + j] * qv[j];
                            tv[lane] = v;
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int j = 0; j < NV; ++j) zp[j] = tv[j];
                    } else {
                        // expand the working rows: dense rows are copied, factored rows are Hc_r * Psi
                        for (int idx = lane; idx < m * NV; idx += WAVE) {
                            const int k = idx / NV, j = idx - k * NV;
                            const int r = Widx[k];
                            double v = 0.0;
                            if (r < nd) {
                                v = Gt[j * NDP + r];
                            } else {
                                if constexpr (KC > 0) {
#pragma unroll
                                    for (int a = 0; a < KC; ++a) v += Hct[a * NCCP + (r - nd)] * Psi[a * NV + j];
                                }
                            }
                            GW[idx] = v;
                        }
                        wave_lds_fence();
                        // T = Hinv G_W'  (entry (i,k): i = idx / m, k = idx % m)
                        for (int idx = lane; idx < NV * m; idx += WAVE) {
                            const int i = idx / m, k = idx - i * m;
                            double v = 0.0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) v += Hinv[i * NV + j] * GW[k * NV + j];
                            T[i * WCAP + k] = v;
                        }
                        wave_lds_fence();
                        // S = G_W T (+ delta I) by rows in registers: lane a holds row a (identity rows beyond m); LDL' by
                        // readlane elimination like the normal matrix -- no LDS round trips on the critical paths
                        double srow[WCAP], sdinv = 1.0;
                        {
                            double gw[NV];
                            const int la = lane < m ? lane : 0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) gw[j] = GW[la * NV + j];
                            double sdiag = 0.0;
#pragma unroll
                            for (int c2 = 0; c2 < WCAP; ++c2) {
                                double v = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += gw[j] * T[j * WCAP + c2];
                                const bool in = lane < m && c2 < m;
                                srow[c2] = in ? v : ((c2 == lane) ? 1.0 : 0.0);
                                if (c2 == lane && in) sdiag = v;
                            }
                            const double dmax = wave_max(sdiag);
#pragma unroll
                            for (int c2 = 0; c2 < WCAP; ++c2) if (c2 == lane && lane < m) srow[c2] += 1e-11 * dmax;
                        }
                        {
                            double bdummy = 0.0;
                            if (!rows_factor<WCAP>(srow, bdummy, sdinv, lane)) break;
                        }
                        // the refinement iterate lives in LDS (zpv), one entry per lane updates it
                        double *zpv = dzav;
                        if (lane < NV) {
                            double mine = 0.0;
#pragma unroll
                            for (int j = 0; j < NV; ++j) mine = (lane == j) ? zp[j] : mine;
                            zpv[lane] = mine;
                        }
                        wave_lds_fence();
                        // proximal Newton steps on the KKT system of the working set (at most four; they stop once a step
                        // no longer moves the iterate)
                        for (int stp = 0; stp < 4; ++stp) {
                            // r1 = Hs zp + q + G_W' y   (lane i -> entry i)
                            if (lane < NV) {
                                double v = qv[lane];
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += Hs[lane * NV + j] * zpv[j];
                                for (int k = 0; k < m; ++k) v += GW[k * NV + lane] * yv[k];
                                tv[lane] = v;
                            }
                            wave_lds_fence();
                            // t1 = Hinv r1
                            if (lane < NV) {
                                double v = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += Hinv[lane * NV + j] * tv[j];
                                uv[lane] = v;
                            }
                            wave_lds_fence();
                            // dy rhs: (G_W zp - h_W) - G_W t1
                            double bb = 0.0;
                            if (lane < m) {
                                const int r = Widx[lane];
                                double gz = 0.0, gt = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) { const double g = GW[lane * NV + j]; gz += g * zpv[j]; gt += g * uv[j]; }
                                const int sl = r < nd ? r : NDP + (r - nd);
                                bb = gz - hw[sl] - gt;
                            }
                            rows_forward<WCAP>(srow, bb, lane);
                            const double dyl = rows_backsub_lane<WCAP>(srow, bb, sdinv, lane);
                            if (lane < m) { dyv[lane] = dyl; yv[lane] += dyl; }
                            wave_lds_fence();
                            // zp -= t1 + T dy  (lane j -> entry j)
                            double dzl = 0.0, zl = 0.0;
                            if (lane < NV) {
                                double v = uv[lane];
                                for (int k = 0; k < m; ++k) v += T[lane * WCAP + k] * dyv[k];
                                zl = zpv[lane] - v;
                                zpv[lane] = zl;
                                dzl = fabs(v);
                            }
                            wave_lds_fence();
                            const double dzn = wave_max(dzl), zn = wave_max(fabs(zl));
                            if (stp >= 1 && dzn <= 1e-14 * fmax(zn, 1.0)) break;
                        }
#pragma unroll
                        for (int j = 0; j < NV; ++j) zp[j] = zpv[j];
                    }
                    // ---- verify: primal feasibility on all rows, sign of y on W
                    double ymax = 1.0;
                    for (int k = 0; k < m; ++k) ymax = fmax(ymax, fabs(yv[k]));
                    int nviol = 0, nneg = 0, nloose = 0;
                    double rr[RT];
                    {
                        double czp[KCA];
                        if constexpr (KC > 0) factor_coords<SH>(Psi, zp, czp, lane);
                        int mm = 0;
                        auto check_slot = [&](auto kc_) {
                            constexpr int k = decltype(kc_)::value;
                            const double gz = row_dot<SH, k>(Gt, Hct, zp, czp, lane);
                            const double hk = hw[k * WAVE + lane];
                            rr[k] = gz - hk;
                            const unsigned long long bal = __ballot(inW[k]);
                            const int pos = mm + __popcll(bal & ((1ull << lane) - 1ull));
                            mm += __popcll(bal);
                            const bool valid = slot_valid<SH>(k, lane, nd, ncc);
                            const double hi = fmax(fabs(hk), 1.0);
                            bool viol = valid && !inW[k] && rr[k] > 1e-12 * hi;
                            // a working-set row that is not on its bound: the Newton steps have not converged
                            const bool loose = inW[k] && fabs(rr[k]) > 1e-11 * hi;
                            bool neg = false;
                            if (inW[k]) { yall[k] = yv[pos < WCAP ? pos : 0]; neg = yall[k] < -1e-10 * ymax; }
                            nviol += __popcll(__ballot(viol));
                            nneg += __popcll(__ballot(neg));
                            nloose += __popcll(__ballot(loose));
                            if (neg) { inW[k] = false; yall[k] = 0.0; }
                            if (viol) { inW[k] = true; yall[k] = 0.0; }
                        };
                        [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (check_slot(std::integral_constant<int, Ks>{}), ...); }
                        (std::make_integer_sequence<int, RT>{});
                    }
                    wave_lds_fence();
                    // rows of W off their bound with nothing left to correct: not converged, give up.  (With wrong
                    // rows still in W the system is inconsistent and looseness is expected: correct W first.)
                    if (nloose != 0 && nviol == 0 && nneg == 0) break;
                    if (nviol == 0 && nneg == 0) {
                        ok = true;
#pragma unroll
                        for (int j = 0; j < NV; ++j) if (lane == j) zv[j] = zp[j];
#pragma unroll
                        for (int k = 0; k < RT; ++k) {
                            lam[k] = inW[k] ? fmax(yall[k], 0.0) : 0.0;
                            s[k] = rr[k] < 0.0 ? -rr[k] : 0.0;
                        }
                    }
                }
            }
            STAMP(7);
            if (ok) { st = TMPC_STATUS_OPTIMAL; break; }
            if (try_tol <= 1e-12) { st = (rdn_last <= 1e-9 * qn) ? TMPC_STATUS_OPTIMAL : TMPC_STATUS_MAX_ITER; break; }
            try_tol *= 1e-2;
            }
            if (st == TMPC_STATUS_MAX_ITER) {
                // iteration cap: if the iterate still violates the constraints, call it infeasible
                double viol = 0.0;
                double z[NV], cz[KCA];
#pragma unroll
                for (int j = 0; j < NV; ++j) z[j] = zv[j];
                if constexpr (KC > 0) factor_coords<SH>(Psi, z, cz, lane);
                auto viol_slot = [&](auto kc_) {
                    constexpr int k = decltype(kc_)::value;
                    const double gz = row_dot<SH, k>(Gt, Hct, z, cz, lane);
                    if (slot_valid<SH>(k, lane, nd, ncc)) viol = fmax(viol, gz - hw[k * WAVE + lane]);
                };
                [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (viol_slot(std::integral_constant<int, Ks>{}), ...); }
                (std::make_integer_sequence<int, RT>{});
                viol = wave_max(viol);
                if (viol > 1e-6 * hn) st = TMPC_STATUS_INFEASIBLE;
            }
        }

        // ---------------------------------------------------------------- outputs
        const bool good = st < TMPC_STATUS_INFEASIBLE;
        const double nanv = __longlong_as_double(0x7ff8000000000000ll);
        // zu = Dv .* z -> zv (LDS) so that any lane can read any entry
        wave_lds_fence();
        if (lane < NV) zv[lane] = (lane < qp.nv) ? qp.Dv[lane] * zv[lane] : 0.0;
        wave_lds_fence();
        for (int i = lane; i < N * nu; i += WAVE) u_nom[b * N * nu + i] = good ? zv[i] : nanv;
        if (lane < nx + nu) {
            double v = 0.0;
            for (int j = 0; j < qp.nth; ++j) v += qp.Mth[lane * qp.nth + j] * zv[qp.off_theta + j];
            if (xu_ss) xu_ss[b * (nx + nu) + lane] = good ? v : nanv;
        }
        // x_nom: x_0 then the recursion x_{i+1} = A x_i + B u_i (reference :138)
        if (lane < nx) {
            const double x0 = (qp.off_x0 >= 0) ? zv[qp.off_x0 + lane] : xin[lane];
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
                    for (int j = 0; j < nu; ++j) v += qp.B[lane * nu + j] * zv[i * nu + j];
                }
                wave_lds_fence();
                if (lane < nx) { tv[lane] = v; x_nom[b * (N + 1) * nx + (i + 1) * nx + lane] = good ? v : nanv; }
                wave_lds_fence();
            }
        }
        if (lane == 0) { status[b] = st; iters[b] = it_done; }
#ifdef TMPC_STAMPS
        STAMP(8);
        if (b == 0 && lane == 0 && qp.dbg) { for (int p_ = 0; p_ < 12; ++p_) qp.dbg[p_] = tph[p_]; }
#endif
        wave_lds_fence();
    }
}

template <class SH>
constexpr size_t kernel_lds_bytes(int wpb) {
    return sizeof(double) * (static_cast<size_t>(SH::NV) * SH::NDP + SH::KC * SH::NCCP + SH::KC * SH::NV + 2 * SH::NV * SH::NV +
                             static_cast<size_t>(wpb) * WaveLds<SH>::TOTAL);
}
// four waves per workgroup (one per SIMD) unless the per-wave workspace does not leave room for that
template <int NV, int RD, int KC, int RC>
constexpr int waves_per_block() {
    using SH = Shape<NV, RD, KC, RC, false>;
    // NV = 24 stays at three waves: with four, solve_kernel<24,3,6,7> faulted (memory aperture violation) on the MI355X --
    // not understood yet, the three-wave build is the one the parity tests have always run
    return NV >= 24 ? (kernel_lds_bytes<SH>(3) <= 160 * 1024 ? 3 : 2)
                    : (kernel_lds_bytes<SH>(4) <= 160 * 1024 ? 4 : (kernel_lds_bytes<SH>(3) <= 160 * 1024 ? 3 : 2));
}
// eight waves per workgroup (two per SIMD, register-lean build) where that fits the LDS; worth it once every SIMD
// has more than one instance to work on (measured on the bench shape: 1.08 vs 1.20 ms at B = 4096, but 0.44 vs
// 0.29 ms for a single instance -- the 256-register cap costs spills)
template <int NV, int RD, int KC, int RC>
constexpr bool fits_two_per_simd() {
    return NV <= 12 && kernel_lds_bytes<Shape<NV, RD, KC, RC, true>>(8) <= 160 * 1024;
}

template <int NV, int RD, int KC, int RC, bool WARM, int WPB>
hipError_t launch_wpb(const DeviceQP &qp, const WarmStart &warm, int variant_id, int64_t B, const double *x_k, const double *ref,
                      const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                      int32_t *status, int32_t *iters, int n_cu, hipStream_t stream) {
    using SH = Shape<NV, RD, KC, RC, WPB == 8>;
    constexpr size_t lds = kernel_lds_bytes<SH>(WPB);
    static_assert(lds <= 160 * 1024, "shape does not fit the 160 KiB LDS of a CU");
    // > 64 KiB of dynamic LDS needs the opt-in once per device and instantiation
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_kernel<NV, RD, KC, RC, WARM, WPB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    int64_t blocks = (B + WPB - 1) / WPB;
    // one persistent workgroup per CU (the LDS footprint admits no second one): the model is staged once and every wave
    // fetches its next instance as soon as it is done with the current one (grid-stride over the batch)
    const int64_t cap = static_cast<int64_t>(n_cu);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((solve_kernel<NV, RD, KC, RC, WARM, WPB>), dim3(static_cast<unsigned>(blocks)), dim3(WAVE * WPB), lds, stream,
                       qp, warm, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters);
    return hipGetLastError();
}

template <int NV, int RD, int KC, int RC, bool WARM>
hipError_t launch_one(const DeviceQP &qp, const WarmStart &warm, int variant_id, int64_t B, const double *x_k, const double *ref,
                      const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                      int32_t *status, int32_t *iters, int n_cu, hipStream_t stream) {
    if constexpr (fits_two_per_simd<NV, RD, KC, RC>()) {
        if (B > static_cast<int64_t>(n_cu) * 4)
            return launch_wpb<NV, RD, KC, RC, WARM, 8>(qp, warm, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status,
                                                        iters, n_cu, stream);
    }
    return launch_wpb<NV, RD, KC, RC, WARM, waves_per_block<NV, RD, KC, RC>()>(qp, warm, variant_id, B, x_k, ref, variant, u_nom, x_nom0,
                                                                               xu_ss, x_nom, status, iters, n_cu, stream);
}

}  // namespace

// Compiled shapes (NVP, RD, KCP, RC).  Dense-only shapes cover the small problems (config 1);
// the factored shapes cover the cartpole (terminal block of 420 rows, width 5) at N = 10 and N = 20.
#define TMPC_SHAPES(X) \
    X(8, 2, 0, 0) X(8, 4, 0, 0) X(12, 2, 0, 0) X(12, 4, 0, 0) X(16, 2, 0, 0) X(16, 4, 0, 0) \
    X(12, 2, 6, 7) X(24, 3, 6, 7)

size_t lds_bytes(const KernelShape &s) {
#define TMPC_LDS(A, B_, C, D) if (s.nvp == A && s.rd == B_ && s.kcp == C && s.rc == D) return kernel_lds_bytes<Shape<A, B_, C, D, false>>(waves_per_block<A, B_, C, D>());
    TMPC_SHAPES(TMPC_LDS)
#undef TMPC_LDS
    return 0;
}

bool pick_config(int nv, int nd, int kc, int ncc, KernelShape *shape) {
    static const int table[][4] = {
#define TMPC_ROW(A, B_, C, D) {A, B_, C, D},
        TMPC_SHAPES(TMPC_ROW)
#undef TMPC_ROW
    };
    long best = -1;
    for (const auto &t : table) {
        if (nv > t[0] || nd > t[1] * WAVE || ncc > t[3] * WAVE) continue;
        if ((kc > 0) != (t[2] > 0) || kc > t[2]) continue;
        if (kc > 0 && nd > 0 && t[1] == 0) continue;
        const long cost = static_cast<long>(t[0]) * t[0] * t[1] + static_cast<long>(t[2]) * t[2] * t[3] + t[0];
        if (best < 0 || cost < best) { best = cost; shape->nvp = t[0]; shape->rd = t[1]; shape->kcp = t[2]; shape->rc = t[3]; }
    }
    return best >= 0;
}

hipError_t launch_solve(const DeviceQP &qp, const KernelShape &s, const WarmStart &warm, int variant_id, int64_t B,
                        const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                        double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, int n_cu, hipStream_t stream) {
#define TMPC_CASE(A, B_, C, D)                                                                                       \
    if (s.nvp == A && s.rd == B_ && s.kcp == C && s.rc == D)                                                         \
        return warm.z ? launch_one<A, B_, C, D, true>(qp, warm, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, n_cu, stream) \
                      : launch_one<A, B_, C, D, false>(qp, warm, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, n_cu, stream);
    TMPC_SHAPES(TMPC_CASE)
#undef TMPC_CASE
    return hipErrorInvalidValue;
}

}  // namespace tmpc
