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
//      rows of Gs are spread over the lanes (row r lives on lane r % 64, slot r / 64), the
//      per-row state (s, lambda, r_p, 1/s, ...) stays in registers, Gs is staged once per
//      workgroup in LDS (column-major, so a lane-per-row read is conflict free).  The
//      normal matrix M = Hs + Gs' D Gs is accumulated per lane in registers, a few
//      columns of its lower triangle at a time (the full triangle does not fit the 256
//      directly addressable VGPRs next to the row state), and summed across the wave
//      through an LDS transposition.  Its Cholesky factor and the solves are done
//      redundantly by every lane in registers (nv <= 16); the factor is parked in LDS
//      between the predictor and the corrector solve.
//   4. active-set refinement on W = {lambda_i > s_i}: proximal Newton steps on the
//      KKT system of the equality-constrained QP (range-space form, S = G_W Hs^-1 G_W'),
//      accepted only when primal feasible on all rows with non-negative multipliers
//   5. outputs: u_nom, x_nom[0], (x_bar, u_bar) = Mth theta, optionally x_nom
//
// Numerics are float64 throughout: cond(Hs) ~ 3e5 after scaling and the weights
// span 1e-1 .. 5e6 (R vs 10 P), float32 cannot resolve the minimiser.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "tmpc_device.hpp"

namespace tmpc {

namespace {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int WCAP = 32;          // max rows in the refinement's working set
constexpr int RED_ROWS = 16;      // entries per transposition round
constexpr int RED_STRIDE = 65;    // 64 lanes + 1 pad: conflict-free transposed reads

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
    // all LDS traffic of this wave issued so far has completed, and the compiler may
    // not move LDS accesses across this point (waves of a block run different QPs, so
    // a block-wide barrier is not available here)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, WAVE);
    hi = __shfl_xor(hi, m, WAVE);
    return __hiloint2double(hi, lo);
}

// Sum each of acc[0..CNT) over the 64 lanes and leave the totals in out[0..CNT) (LDS).
// Round: 16 entries are written as rows of a [16][65] LDS tile, lane l then adds a
// 16-lane quarter (l>>4) of entry (l&15); the four quarters meet through two shuffles.
template <int CNT>
__device__ __forceinline__ void wave_reduce_to_lds(const double (&acc)[CNT], double *red, double *out, int lane) {
    const int e = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int c0 = 0; c0 < CNT; c0 += RED_ROWS) {
#pragma unroll
        for (int k = 0; k < RED_ROWS; ++k)
            if (c0 + k < CNT) red[k * RED_STRIDE + lane] = acc[c0 + k];
        wave_lds_fence();
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[e * RED_STRIDE + qd * 16 + j];
        t += shfl_xor_d(t, 16);
        t += shfl_xor_d(t, 32);
        if (qd == 0 && c0 + e < CNT) out[c0 + e] = t;
        wave_lds_fence();
    }
}

// ---- packed lower triangle, column-major: (i,j), i >= j, at col_off(j) + i - j
template <int NV>
__host__ __device__ constexpr int col_off(int j) { return j * NV - j * (j - 1) / 2; }

// Column blocks of the lower triangle accumulated per sweep (<= ~40 accumulators each).
template <int NV> struct Blocks;
template <> struct Blocks<8>  { static constexpr int n = 2; static constexpr int b[3] = {0, 3, 8}; };
template <> struct Blocks<12> { static constexpr int n = 3; static constexpr int b[4] = {0, 3, 7, 12}; };
template <> struct Blocks<16> { static constexpr int n = 5; static constexpr int b[6] = {0, 2, 4, 7, 11, 16}; };

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
        const double pinv = 1.0 / pkk;
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
// back substitution; x comes out wave-uniform.  (b is destroyed.)
template <int NV>
__device__ __forceinline__ void rows_backsub(const double (&row)[NV], double b, double dinv, double (&x)[NV]) {
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) {
        const double xi = readlane_d(b * dinv, i);
        x[i] = xi;
        b = fma(-row[i], xi, b);      // lanes r < i consume U[r][i]; the others are already done
    }
}

// per-wave LDS workspace (doubles), see solve_kernel
template <int NV, int RPL>
struct WaveLds {
    static constexpr int NT = NV * (NV + 1) / 2;
    static constexpr int RED = RED_ROWS * RED_STRIDE;                       // transposition tile
    static constexpr int POL = NV * WCAP + WCAP * (WCAP + 1) + 4 * WCAP;   // T, S, y, dy, W(idx)
    static constexpr int BIG = RED > POL ? RED : POL;                       // the two are never live together
    static constexpr int SUMS = NT + 2 * NV + 8;                            // reduced totals of a sweep
    static constexpr int HROW = RPL * WAVE;                                 // right-hand side h, [slot][lane]
    static constexpr int VEC = 8 * NV + 32;                                 // q, z, cost gradient, x_k, ref, scratch
    static constexpr int TOTAL = BIG + SUMS + HROW + VEC;
};

// One sweep over the rows of this lane for columns [J0, J1) of the lower triangle of G'DG.
// FIRST also forms r_p and 1/s; LAST also accumulates G'(d.r_p), G'lam and the gap.
template <int NV, int RPL, int J0, int J1, bool FIRST, bool LAST>
__device__ __forceinline__ void sweep_a(const double *Gt, const double *hw, const double (&z)[NV],
                                        const double (&s)[RPL], const double (&lam)[RPL], double (&rp)[RPL], double (&rs)[RPL],
                                        double *red, double *sums, int lane, int nc) {
    constexpr int NCP = RPL * WAVE;
    constexpr int TRI = col_off<NV>(J1) - col_off<NV>(J0);
    constexpr int CNT = TRI + (LAST ? 2 * NV + 1 : 0);
    constexpr int I0 = (FIRST || LAST) ? 0 : J0;        // first column this sweep has to load
    static_assert(!LAST || col_off<NV>(J1) == NV * (NV + 1) / 2, "the last block must end the triangle");
    double acc[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] = 0.0;
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = lane + k * WAVE;
        double g[NV];
#pragma unroll
        for (int j = I0; j < NV; ++j) g[j] = Gt[j * NCP + r];
        if (FIRST) {
            double gz = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) gz += g[j] * z[j];
            rp[k] = gz + s[k] - hw[k * WAVE + lane];
            rs[k] = (r < nc) ? fast_rcp(s[k]) : 0.0;
        }
        const double d = lam[k] * rs[k];
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const double dg = d * g[j];
#pragma unroll
            for (int i = j; i < NV; ++i) acc[col_off<NV>(j) - col_off<NV>(J0) + i - j] += dg * g[i];
        }
        if (LAST) {
            const double t = d * rp[k];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                acc[TRI + i] += g[i] * t;
                acc[TRI + NV + i] += g[i] * lam[k];
            }
            acc[TRI + 2 * NV] += s[k] * lam[k];
        }
        row_fence();
    }
    // totals: the triangle block lands at its packed position; the last block's vectors and the gap
    // follow the triangle directly (col_off(J1) == NT there)
    wave_reduce_to_lds<CNT>(acc, red, sums + col_off<NV>(J0), lane);
}

template <int NV, int RPL, int BI>
__device__ __forceinline__ void sweep_a_all(const double *Gt, const double *hw, const double (&z)[NV],
                                            const double (&s)[RPL], const double (&lam)[RPL], double (&rp)[RPL],
                                            double (&rs)[RPL], double *red, double *sums, int lane, int nc) {
    using BL = Blocks<NV>;
    if constexpr (BI < BL::n) {
        sweep_a<NV, RPL, BL::b[BI], BL::b[BI + 1], BI == 0, BI == BL::n - 1>(Gt, hw, z, s, lam, rp, rs, red, sums, lane, nc);
        sweep_a_all<NV, RPL, BI + 1>(Gt, hw, z, s, lam, rp, rs, red, sums, lane, nc);
    }
}

template <int NV, int RPL>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK, 1) void solve_kernel(
    const DeviceQP qp, const int variant_id, const int64_t B,
    const double *__restrict__ x_k, const double *__restrict__ ref, const uint8_t *__restrict__ variant,
    double *__restrict__ u_nom, double *__restrict__ x_nom0, double *__restrict__ xu_ss,
    double *__restrict__ x_nom, int32_t *__restrict__ status, int32_t *__restrict__ iters) {
    constexpr int NCP = RPL * WAVE;
    constexpr int NT = NV * (NV + 1) / 2;
    using WL = WaveLds<NV, RPL>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *Gt = smem;                         // [NV][NCP]
    double *Hs = Gt + NV * NCP;                // [NV][NV]
    double *Hinv = Hs + NV * NV;               // [NV][NV]
    double *wbase = Hinv + NV * NV;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid >> 6;
    const int nx = qp.nx, nu = qp.nu, N = qp.N, nc = qp.nc;

    // ---- stage the shared model once per workgroup (coalesced, L2-resident source)
    for (int i = tid; i < NV * NCP; i += blockDim.x) Gt[i] = qp.Gt[i];
    for (int i = tid; i < NV * NV; i += blockDim.x) { Hs[i] = qp.Hs[i]; Hinv[i] = qp.Hinv[i]; }
    __syncthreads();

    double *red = wbase + wave * WL::TOTAL;       // transposition tile / refinement workspace
    double *sums = red + WL::BIG;                 // reduced totals: triangle (column-major packed), vectors, gap
    double *hw = sums + WL::SUMS;                 // h, [slot][lane]
    double *vec = hw + WL::HROW;
    double *qv = vec;                 // [NV] linear term
    double *zv = vec + NV;            // [NV] current z (wave-uniform copy)
    double *cgv = vec + 2 * NV;       // [NV] cost gradient
    double *xin = vec + 3 * NV;       // [2*nx] x_k | ref   (nx <= 16)
    double *tv = vec + 3 * NV + 32;   // [NV] scratch
    double *uv = vec + 4 * NV + 32;   // [NV] scratch

    const int64_t wave_global = static_cast<int64_t>(blockIdx.x) * WAVES_PER_BLOCK + wave;
    const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * WAVES_PER_BLOCK;

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
        for (int k = 0; k < RPL; ++k) {
            const int r = lane + k * WAVE;
            double v = 1.0;
            if (r < nc) {
                v = qp.g0s[r];
                for (int c = 0; c < nx; ++c) v += qp.Es[r * nx + c] * xin[c];
                hn = fmax(hn, fabs(v));
            }
            hw[k * WAVE + lane] = v;
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
        double z[NV];
        double qn = 1.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) { z[j] = zv[j]; qn = fmax(qn, fabs(qv[j])); }

        double s[RPL], lam[RPL];
        double smin = INFINITY;
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const int r = lane + k * WAVE;
            double gz = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) gz += Gt[j * NCP + r] * z[j];
            s[k] = hw[k * WAVE + lane] - gz;
            lam[k] = 0.0;
            if (r < nc) smin = fmin(smin, s[k]);
        }
        smin = wave_min(smin);
        STAMP(0);

        if (infeasible_par) {
            st = TMPC_STATUS_INFEASIBLE;
        } else if (smin >= 0.0) {
            st = TMPC_STATUS_OPTIMAL;
        } else {
            // -------------------------------------------------------- interior point
            {
                const double viol = -smin;
                const double fl = 0.1 * fmax(viol, 1.0);
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    const bool valid = lane + k * WAVE < nc;
                    s[k] = valid ? fmax(s[k], fl) : 1.0;
                    lam[k] = valid ? 1.0 : 0.0;
                }
            }
            double try_tol = qp.tol;
            const double ncd = static_cast<double>(nc);
            int it = 0;
            double rdn_last = 0.0;
            // interior point until the active set can be read off, then the refinement; the pair is
            // repeated (with a tighter hand-over tolerance) only if the refinement cannot certify its set
            for (;;) {
            bool want_polish = false;
            for (; it < qp.max_iter; ++it) {
                it_done = it;
                // ---- sweeps A: residuals, 1/s, lower triangle of G'DG by column blocks, G'(d.rp), G'lam, gap
                double rp[RPL], rs[RPL];
                sweep_a_all<NV, RPL, 0>(Gt, hw, z, s, lam, rp, rs, red, sums, lane, nc);
                STAMP(1);
                double rpn = 0.0, lmax = 0.0;
#pragma unroll
                for (int k = 0; k < RPL; ++k) { rpn = fmax(rpn, fabs(rp[k])); lmax = fmax(lmax, lam[k]); }
                rpn = wave_max(rpn);
                lmax = wave_max(lmax);
                const double gap = sums[NT + 2 * NV];
                const double mu = gap / ncd;
                // cost gradient cg = Hs z + q (lane i computes entry i)
                if (lane < NV) {
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) v += Hs[lane * NV + j] * zv[j];
                    cgv[lane] = v + qv[lane];
                }
                wave_lds_fence();
                double rdn = 0.0, obj = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const double cgj = cgv[j], qj = qv[j];
                    rdn = fmax(rdn, fabs(cgj + sums[NT + NV + j]));
                    obj += z[j] * (0.5 * (cgj - qj) + qj);
                }
                if (!(mu == mu) || !(rdn == rdn)) { st = TMPC_STATUS_NUMERICAL; break; }
                const double objs = fmax(fabs(obj), 1.0);
                STAMP(2);
                const bool try_polish = (rdn <= 1e3 * try_tol * qn) && (rpn <= try_tol * hn) && (gap <= try_tol * objs);
                if (try_polish) { want_polish = true; rdn_last = rdn; break; }
                if (gap <= 1e-15 * objs) { st = TMPC_STATUS_MAX_ITER; break; }
                if (lmax > 1e10) {
                    // Farkas-type certificate: lam blows up, G'lam -> 0, h'lam < 0
                    double hl = 0.0;
#pragma unroll
                    for (int k = 0; k < RPL; ++k) hl += hw[k * WAVE + lane] * lam[k];
                    hl = wave_sum(hl);
                    double gn = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) gn = fmax(gn, fabs(sums[NT + NV + j]));
                    if (hl < 0.0 && gn <= 1e-6 * lmax) { st = TMPC_STATUS_INFEASIBLE; break; }
                }
                // ---- M = Hs + G'DG by rows (lane i holds row i), elimination with the predictor rhs carried along
                double mrow[NV], mdinv = 1.0, rhs_i = 0.0;
                double dz[NV];
                {
                    const int li = lane < NV ? lane : 0;
                    double shift = 0.0;
                    bool spd = false;
                    for (int attempt = 0; attempt < 2 && !spd; ++attempt) {
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            const int lo = li < j ? li : j, hi2 = li < j ? j : li;
                            const double v = Hs[li * NV + j] + sums[lo * NV - lo * (lo - 1) / 2 + hi2 - lo] + (li == j ? shift : 0.0);
                            mrow[j] = (lane < NV) ? v : 0.0;
                        }
                        rhs_i = (lane < NV) ? -cgv[li] - sums[NT + li] : 0.0;
                        double bb = rhs_i;
                        mdinv = 1.0;
                        spd = rows_factor<NV>(mrow, bb, mdinv, lane);
                        if (spd) {
                            rows_backsub<NV>(mrow, bb, mdinv, dz);
                        } else {
                            // non-positive pivot from cancellation: retry once with a 1e-13 * trace shift
                            double trc = 0.0;
#pragma unroll
                            for (int i = 0; i < NV; ++i) trc += Hs[i * NV + i] + sums[col_off<NV>(i)];
                            shift = 1e-13 * trc;
                        }
                    }
                    if (!spd) { st = TMPC_STATUS_NUMERICAL; break; }
                }
                STAMP(3);
                // ---- sweep B: affine step statistics and the corrector's G' products
                double wprod[RPL];
                double rho_aff = 0.0;
                {
                    double accb[2 * NV + 2];
#pragma unroll
                    for (int i = 0; i < 2 * NV + 2; ++i) accb[i] = 0.0;
#pragma unroll
                    for (int k = 0; k < RPL; ++k) {
                        const int r = lane + k * WAVE;
                        const bool valid = r < nc;
                        double g[NV];
                        double gdz = 0.0;
#pragma unroll
                        for (int j = 0; j < NV; ++j) { g[j] = Gt[j * NCP + r]; gdz += g[j] * dz[j]; }
                        const double dsa = valid ? (-rp[k] - gdz) : 0.0;
                        const double dla = valid ? (-lam[k] - lam[k] * rs[k] * dsa) : 0.0;
                        const double rl = valid ? fast_rcp(lam[k]) : 0.0;
                        rho_aff = fmax(rho_aff, fmax(-dsa * rs[k], -dla * rl));
                        wprod[k] = dsa * dla;
                        accb[2 * NV] += s[k] * dla + lam[k] * dsa;
                        accb[2 * NV + 1] += wprod[k];
                        const double c1 = wprod[k] * rs[k];
#pragma unroll
                        for (int j = 0; j < NV; ++j) { accb[j] += g[j] * c1; accb[NV + j] += g[j] * rs[k]; }
                        row_fence();
                    }
                    rho_aff = wave_max(rho_aff);
                    wave_reduce_to_lds<2 * NV + 2>(accb, red, sums + NT, lane);   // overwrites G'(d.rp), G'lam (consumed)
                }
                STAMP(4);
                const double aaff = rho_aff > 1.0 ? 1.0 / rho_aff : 1.0;
                const double mu_aff = (gap + aaff * sums[NT + 2 * NV] + aaff * aaff * sums[NT + 2 * NV + 1]) / ncd;
                double sigma = mu_aff / mu;
                sigma = fmin(sigma * sigma * sigma, 1.0);
                const double smu = sigma * mu;
                {
                    const int li = lane < NV ? lane : 0;
                    double bb = (lane < NV) ? rhs_i + sums[NT + li] - smu * sums[NT + NV + li] : 0.0;
                    rows_forward<NV>(mrow, bb, lane);
                    rows_backsub<NV>(mrow, bb, mdinv, dz);
                }
                STAMP(5);
                // ---- sweep D: final direction, step length, update
                double ds[RPL];
                double om = (1.0 - aaff) * (1.0 - aaff);
                om = fmin(fmax(om, 1e-4), 1e-2);
                const double tau = 1.0 - om;
                double rho = 0.0;
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    const int r = lane + k * WAVE;
                    const bool valid = r < nc;
                    double gdz = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) gdz += Gt[j * NCP + r] * dz[j];
                    ds[k] = valid ? (-rp[k] - gdz) : 0.0;
                    const double rc = s[k] * lam[k] + wprod[k] - smu;
                    const double dl = valid ? (-(rc + lam[k] * ds[k]) * rs[k]) : 0.0;
                    const double rl = valid ? fast_rcp(lam[k]) : 0.0;
                    rho = fmax(rho, fmax(-ds[k] * rs[k], -dl * rl));
                    wprod[k] = dl;                        // the product is consumed; keep dl in its place
                    row_fence();
                }
                rho = wave_max(rho);
                const double alpha = rho > tau ? tau / rho : 1.0;
#pragma unroll
                for (int k = 0; k < RPL; ++k) { s[k] += alpha * ds[k]; lam[k] += alpha * wprod[k]; }
#pragma unroll
                for (int j = 0; j < NV; ++j) z[j] += alpha * dz[j];
#pragma unroll
                for (int j = 0; j < NV; ++j) if (lane == j) zv[j] = z[j];     // z is wave-uniform
                wave_lds_fence();
                it_done = it + 1;
                STAMP(6);
            }
            if (!want_polish) break;
            // ------------------------------------------------ active-set refinement
            bool ok = false;
                {
                    // workspace carved from the (now idle) transposition tile
                    double *T = red;                          // [NV][WCAP]
                    double *S = T + NV * WCAP;                // [WCAP][WCAP+1]
                    double *yv = S + WCAP * (WCAP + 1);       // [WCAP]
                    double *dyv = yv + WCAP;                  // [WCAP]
                    int *Widx = reinterpret_cast<int *>(dyv + WCAP);   // [WCAP] (+ spare)
                    bool inW[RPL];
                    double yall[RPL];
#pragma unroll
                    for (int k = 0; k < RPL; ++k) { inW[k] = (lane + k * WAVE < nc) && (lam[k] > s[k]); yall[k] = lam[k]; }
                    double zp[NV];
#pragma unroll
                    for (int j = 0; j < NV; ++j) zp[j] = z[j];
                    for (int round = 0; round < 6 && !ok; ++round) {
                        // compact the working set: W[0..m)
                        int m = 0;
#pragma unroll
                        for (int k = 0; k < RPL; ++k) {
                            const unsigned long long bal = __ballot(inW[k]);
                            const int pos = m + __popcll(bal & ((1ull << lane) - 1ull));
                            if (inW[k] && pos < WCAP) { Widx[pos] = lane + k * WAVE; yv[pos] = yall[k]; }
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
                            // T = Hinv G_W'  (entry (i,k): i = idx / m, k = idx % m)
                            for (int idx = lane; idx < NV * m; idx += WAVE) {
                                const int i = idx / m, k = idx - i * m;
                                const int r = Widx[k];
                                double v = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += Hinv[i * NV + j] * Gt[j * NCP + r];
                                T[i * WCAP + k] = v;
                            }
                            wave_lds_fence();
                            // S = G_W T (+ delta I)
                            for (int idx = lane; idx < m * m; idx += WAVE) {
                                const int a = idx / m, c2 = idx - a * m;
                                const int r = Widx[a];
                                double v = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) v += Gt[j * NCP + r] * T[j * WCAP + c2];
                                S[a * (WCAP + 1) + c2] = v;
                            }
                            wave_lds_fence();
                            double dmax = 0.0;
                            if (lane < m) dmax = S[lane * (WCAP + 1) + lane];
                            dmax = wave_max(dmax);
                            if (lane < m) S[lane * (WCAP + 1) + lane] += 1e-11 * dmax;
                            wave_lds_fence();
                            // Cholesky of S in LDS, right-looking; lane a owns row a
                            bool spd = true;
                            for (int j = 0; j < m; ++j) {
                                const double pjj = S[j * (WCAP + 1) + j];
                                if (!(pjj > 0.0)) { spd = false; break; }
                                const double piv = sqrt(pjj);
                                double lij = 0.0;
                                if (lane > j && lane < m) lij = S[lane * (WCAP + 1) + j] / piv;
                                wave_lds_fence();
                                if (lane == j) S[j * (WCAP + 1) + j] = piv;
                                if (lane > j && lane < m) S[lane * (WCAP + 1) + j] = lij;
                                wave_lds_fence();
                                // trailing update: row `lane`, columns j+1..lane
                                if (lane > j && lane < m) {
                                    for (int c2 = j + 1; c2 <= lane; ++c2)
                                        S[lane * (WCAP + 1) + c2] -= lij * S[c2 * (WCAP + 1) + j];
                                }
                                wave_lds_fence();
                            }
                            if (!spd) break;
                            // four proximal Newton steps on the KKT system of the working set
                            for (int stp = 0; stp < 4; ++stp) {
                                // r1 = Hs zp + q + G_W' y   (lane i -> entry i)
                                if (lane < NV) {
                                    double v = qv[lane];
#pragma unroll
                                    for (int j = 0; j < NV; ++j) v += Hs[lane * NV + j] * zp[j];
                                    for (int k = 0; k < m; ++k) v += Gt[lane * NCP + Widx[k]] * yv[k];
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
                                if (lane < m) {
                                    const int r = Widx[lane];
                                    double gz = 0.0, gt = 0.0;
#pragma unroll
                                    for (int j = 0; j < NV; ++j) { const double g = Gt[j * NCP + r]; gz += g * zp[j]; gt += g * uv[j]; }
                                    dyv[lane] = gz - hw[(r >> 6) * WAVE + (r & 63)] - gt;
                                }
                                wave_lds_fence();
                                // forward / backward substitution with L (in S), m sequential steps each
                                for (int j = 0; j < m; ++j) {
                                    const double vj = dyv[j] / S[j * (WCAP + 1) + j];
                                    wave_lds_fence();
                                    if (lane == j) dyv[j] = vj;
                                    if (lane > j && lane < m) dyv[lane] -= S[lane * (WCAP + 1) + j] * vj;
                                    wave_lds_fence();
                                }
                                for (int j = m - 1; j >= 0; --j) {
                                    const double vj = dyv[j] / S[j * (WCAP + 1) + j];
                                    wave_lds_fence();
                                    if (lane == j) dyv[j] = vj;
                                    if (lane < j) dyv[lane] -= S[j * (WCAP + 1) + lane] * vj;
                                    wave_lds_fence();
                                }
                                // zp -= t1 + T dy ; y += dy
#pragma unroll
                                for (int j = 0; j < NV; ++j) {
                                    double v = uv[j];
                                    for (int k = 0; k < m; ++k) v += T[j * WCAP + k] * dyv[k];
                                    zp[j] -= v;
                                }
                                if (lane < m) yv[lane] += dyv[lane];
                                wave_lds_fence();
                            }
                        }
                        // ---- verify: primal feasibility on all rows, sign of y on W
                        double ymax = 1.0;
                        for (int k = 0; k < m; ++k) ymax = fmax(ymax, fabs(yv[k]));
                        int nviol = 0, nneg = 0, nloose = 0;
                        double rr[RPL];
                        {
                            int mm = 0;
#pragma unroll
                            for (int k = 0; k < RPL; ++k) {
                                const int r = lane + k * WAVE;
                                double gz = 0.0;
#pragma unroll
                                for (int j = 0; j < NV; ++j) gz += Gt[j * NCP + r] * zp[j];
                                const double hk = hw[k * WAVE + lane];
                                rr[k] = gz - hk;
                                const unsigned long long bal = __ballot(inW[k]);
                                const int pos = mm + __popcll(bal & ((1ull << lane) - 1ull));
                                mm += __popcll(bal);
                                const bool valid = r < nc;
                                const double hi = fmax(fabs(hk), 1.0);
                                bool viol = valid && !inW[k] && rr[k] > 1e-12 * hi;
                                // a working-set row that is not on its bound: the Newton steps have not converged
                                const bool loose = inW[k] && fabs(rr[k]) > 1e-11 * hi;
                                bool neg = false;
                                if (inW[k]) { yall[k] = yv[pos]; neg = yall[k] < -1e-10 * ymax; }
                                nviol += __popcll(__ballot(viol));
                                nneg += __popcll(__ballot(neg));
                                nloose += __popcll(__ballot(loose));
                                if (neg) { inW[k] = false; yall[k] = 0.0; }
                                if (viol) { inW[k] = true; yall[k] = 0.0; }
                            }
                        }
                        wave_lds_fence();
                        if (nloose != 0) break;
                        if (nviol == 0 && nneg == 0) {
                            ok = true;
#pragma unroll
                            for (int j = 0; j < NV; ++j) z[j] = zp[j];
#pragma unroll
                            for (int k = 0; k < RPL; ++k) {
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
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    const int r = lane + k * WAVE;
                    double gz = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) gz += Gt[j * NCP + r] * z[j];
                    if (r < nc) viol = fmax(viol, gz - hw[k * WAVE + lane]);
                }
                viol = wave_max(viol);
                if (viol > 1e-6 * hn) st = TMPC_STATUS_INFEASIBLE;
            }
        }

        // ---------------------------------------------------------------- outputs
        const bool good = st < TMPC_STATUS_INFEASIBLE;
        const double nanv = __longlong_as_double(0x7ff8000000000000ll);
        // zu = Dv .* z -> zv (LDS) so that any lane can read any entry
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < NV; ++j) if (lane == j) zv[j] = (j < qp.nv) ? qp.Dv[j] * z[j] : 0.0;
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

template <int NV, int RPL>
constexpr size_t kernel_lds_bytes() {
    return sizeof(double) * (static_cast<size_t>(NV) * RPL * WAVE + 2 * NV * NV + WAVES_PER_BLOCK * WaveLds<NV, RPL>::TOTAL);
}

template <int NV, int RPL>
hipError_t launch_one(const DeviceQP &qp, int variant_id, int64_t B, const double *x_k, const double *ref,
                      const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                      int32_t *status, int32_t *iters, int n_cu, hipStream_t stream) {
    constexpr size_t lds = kernel_lds_bytes<NV, RPL>();
    // > 64 KiB of dynamic LDS needs the opt-in once per device and instantiation
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_kernel<NV, RPL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    int64_t blocks = (B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t cap = static_cast<int64_t>(n_cu) * 4;     // a few workgroups per CU, grid-stride over the batch
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((solve_kernel<NV, RPL>), dim3(static_cast<unsigned>(blocks)), dim3(WAVE * WAVES_PER_BLOCK), lds, stream,
                       qp, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters);
    return hipGetLastError();
}

}  // namespace

size_t lds_bytes(int nvp, int rpl) {
#define TMPC_LDS(NVV, RPLV) if (nvp == NVV && rpl == RPLV) return kernel_lds_bytes<NVV, RPLV>();
    TMPC_LDS(8, 2) TMPC_LDS(8, 4) TMPC_LDS(8, 8) TMPC_LDS(12, 2) TMPC_LDS(12, 4) TMPC_LDS(12, 8) TMPC_LDS(16, 2) TMPC_LDS(16, 4)
#undef TMPC_LDS
    return 0;
}

bool pick_config(int nv, int nc, int *nvp, int *rpl) {
    static const int nvs[] = {8, 12, 16};
    static const int rpls[] = {2, 4, 8};
    for (int a : nvs) {
        if (nv > a) continue;
        for (int r : rpls) {
            if (nc > r * WAVE) continue;
            if (a == 16 && r == 8) continue;    // not instantiated (LDS/VGPR budget)
            *nvp = a; *rpl = r;
            return true;
        }
    }
    return false;
}

#define TMPC_CASE(NVV, RPLV)                                                                                         \
    if (nvp == NVV && rpl == RPLV)                                                                                   \
        return launch_one<NVV, RPLV>(qp, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, n_cu, stream);

hipError_t launch_solve(const DeviceQP &qp, int nvp, int rpl, int variant_id, int64_t B, const double *x_k,
                        const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss,
                        double *x_nom, int32_t *status, int32_t *iters, int n_cu, hipStream_t stream) {
    TMPC_CASE(8, 2) TMPC_CASE(8, 4) TMPC_CASE(8, 8)
    TMPC_CASE(12, 2) TMPC_CASE(12, 4) TMPC_CASE(12, 8)
    TMPC_CASE(16, 2) TMPC_CASE(16, 4)
    return hipErrorInvalidValue;
}

}  // namespace tmpc
