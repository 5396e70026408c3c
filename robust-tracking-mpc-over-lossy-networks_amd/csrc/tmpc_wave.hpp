// Wave-level device helpers shared by the block kernel (tmpc_block.hip) and the LP kernel (tmpc_lp.hip): DPP / readlane
// reductions over the 64 lanes of a gfx950 wavefront and a division-free reciprocal.
#pragma once
#ifdef TMPC_HOST_SIM
#include "hip_sim.hpp"      // tests/wavesim: the same source compiled for the CPU (sanitizer runs), never in the product
#else
#include <hip/hip_runtime.h>
#endif

#include <utility>

namespace tmpc {
namespace wv {

constexpr int WAVE = 64;

// 1/x to full double precision for normal, finite x: v_rcp_f64 + two Newton steps
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

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
// max / min as the bare instruction.  `fmax` compiles to v_max_f64 preceded by a canonicalising v_max_f64 x, x of every
// operand the compiler cannot prove quiet (2-3 instructions per call in the row sweeps); the instruction itself already
// returns the other operand when one is a NaN, which is all these reductions need.
#ifdef TMPC_HOST_SIM
__device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double vmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double vmax_abs(double a, double b) { return fmax(a, fabs(b)); }
#else
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_abs(double a, double b) {      // max(a, |b|)
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif
struct OpSum { __device__ __forceinline__ static double f(double a, double b) { return a + b; } };
struct OpMin { __device__ __forceinline__ static double f(double a, double b) { return vmin(a, b); } };
struct OpMax { __device__ __forceinline__ static double f(double a, double b) { return vmax(a, b); } };

// all-reduce over the wave: every lane returns the total
template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
    v = Op::f(v, dpp_mov_d<0xB1>(v));    // quad_perm [1,0,3,2]
    v = Op::f(v, dpp_mov_d<0x4E>(v));    // quad_perm [2,3,0,1]
    v = Op::f(v, dpp_mov_d<0x141>(v));   // row_half_mirror
    v = Op::f(v, dpp_mov_d<0x140>(v));   // row_mirror
    const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    return Op::f(Op::f(r0, r1), Op::f(r2, r3));
}

// Orders one wave's LDS traffic for the compiler (the hardware runs the DS instructions of a wave in issue order).
#ifdef TMPC_HOST_SIM
__device__ __forceinline__ void lds_fence() { sim::wave_fence(); }
#else
__device__ __forceinline__ void lds_fence() { asm volatile("" ::: "memory"); }
#endif

// Sum each of acc[0..CNT) over the 64 lanes, totals to out[0..CNT) (LDS).  `red` is a [16][68] tile: 16 entries per round
// are written as rows (lane l at column l + l/16), lane l then adds the 16-lane quarter (l & 3) of entry (l >> 2) and
// the four quarters of a quad meet through two DPP quad permutes.
constexpr int RED_STRIDE = 68;
template <int CNT>
__device__ __forceinline__ void reduce_to_lds(const double (&acc)[CNT], double *red, double *out, int lane) {
    const int e = lane >> 2, qd = lane & 3;
    const int wcol = lane + (lane >> 4);
#pragma unroll
    for (int c0 = 0; c0 < CNT; c0 += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (c0 + k < CNT) red[k * RED_STRIDE + wcol] = acc[c0 + k];
        lds_fence();
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[e * RED_STRIDE + qd * 17 + j];
        t += dpp_mov_d<0xB1>(t);
        t += dpp_mov_d<0x4E>(t);
        if (qd == 0 && c0 + e < CNT) out[c0 + e] = t;
        lds_fence();
    }
}

// n x n symmetric positive definite solve with the rows on the lanes: lane i (< N) holds row i in registers; LDL' by
// Gaussian elimination without pivoting, the pivot row broadcast with v_readlane.  `b` is carried as an extra column.
template <int N>
__device__ __forceinline__ bool rows_factor(double (&row)[N], double &b, double &dinv, int lane) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double pkk = readlane_d(row[k], k);
        ok = ok && (pkk > 0.0);
        const double pinv = fast_rcp(pkk);
        const double f = (lane > k) ? row[k] * pinv : 0.0;
#pragma unroll
        for (int j = k + 1; j < N; ++j) row[j] = fma(-f, readlane_d(row[j], k), row[j]);
        b = fma(-f, readlane_d(b, k), b);
        if (lane > k) row[k] = f;
        if (lane == k) dinv = pinv;
    }
    return ok;
}
template <int N>
__device__ __forceinline__ void rows_forward(const double (&row)[N], double &b, int lane) {
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        const double f = (lane > k) ? row[k] : 0.0;
        b = fma(-f, readlane_d(b, k), b);
    }
}
// back substitution; x_i is returned on lane i
template <int N>
__device__ __forceinline__ double rows_backsub_lane(const double (&row)[N], double b, double dinv, int lane) {
    double xl = 0.0;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        const double bi = b * dinv;
        const double xi = readlane_d(bi, i);
        xl = (lane == i) ? bi : xl;
        b = fma(-row[i], xi, b);
    }
    return xl;
}

// ---- the same three routines for N <= 16 with the pivot row handed round by ONE 64-bit DPP move instead of two
// v_readlane_b32: `v_mov_b64_dpp ... row_newbcast:k` (gfx90a and later; the only DPP control the double-precision ALU
// takes) copies lane k of every 16-lane row to the whole row.  Two instructions per multiply-add of the elimination
// instead of three, and no SGPR round trip.  Every 16-lane row of the wave works on the matrix whose row i sits on its
// lane i: the callers keep theirs on lanes 0 .. N-1 and ignore what the other three rows compute.
template <int K>
__device__ __forceinline__ double row_bcast_d(double v) {
    return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xF, 0xF, false);      // row_newbcast:K
}
template <int N, class F>
__device__ __forceinline__ void static_for_n(F &&f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
// returns whether the matrix of lanes 0 .. N-1 had positive pivots (wave-uniform)
template <int N>
__device__ __forceinline__ bool rows16_factor(double (&row)[N], double &b, double &dinv, int lane) {
    static_assert(N <= 16, "one matrix row per lane of a 16-lane DPP row");
    const int l16 = lane & 15;
    bool ok = true;
    static_for_n<N>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        const double pkk = row_bcast_d<k>(row[k]);
        ok = ok && (pkk > 0.0);
        const double pinv = fast_rcp(pkk);
        const double f = (l16 > k) ? row[k] * pinv : 0.0;
#pragma unroll
        for (int j = k + 1; j < N; ++j) row[j] = fma(-f, row_bcast_d<k>(row[j]), row[j]);
        b = fma(-f, row_bcast_d<k>(b), b);
        if (l16 > k) row[k] = f;
        if (l16 == k) dinv = pinv;
    });
    return __builtin_amdgcn_readfirstlane(static_cast<int>(ok)) != 0;
}
template <int N>
__device__ __forceinline__ void rows16_forward(const double (&row)[N], double &b, int lane) {
    const int l16 = lane & 15;
    static_for_n<N - 1>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        const double f = (l16 > k) ? row[k] : 0.0;
        b = fma(-f, row_bcast_d<k>(b), b);
    });
}
template <int N>
__device__ __forceinline__ double rows16_backsub_lane(const double (&row)[N], double b, double dinv, int lane) {
    const int l16 = lane & 15;
    double xl = 0.0;
    static_for_n<N>([&](auto r_) {
        constexpr int i = N - 1 - decltype(r_)::value;
        const double bi = b * dinv;
        const double xi = row_bcast_d<i>(bi);
        xl = (l16 == i) ? bi : xl;
        b = fma(-row[i], xi, b);
    });
    return xl;
}
// dispatch: the DPP form where a matrix fits a 16-lane row, the readlane form otherwise (-DTMPC_NO_DPP64: diagnostic
// builds with the readlane form throughout)
#ifdef TMPC_NO_DPP64
constexpr int DPP_ROW = 0;
#else
constexpr int DPP_ROW = 16;
#endif
template <int N>
__device__ __forceinline__ bool lanes_factor(double (&row)[N], double &b, double &dinv, int lane) {
    if constexpr (N <= DPP_ROW) return rows16_factor<N>(row, b, dinv, lane);
    else return rows_factor<N>(row, b, dinv, lane);
}
template <int N>
__device__ __forceinline__ void lanes_forward(const double (&row)[N], double &b, int lane) {
    if constexpr (N <= DPP_ROW) rows16_forward<N>(row, b, lane);
    else rows_forward<N>(row, b, lane);
}
template <int N>
__device__ __forceinline__ double lanes_backsub_lane(const double (&row)[N], double b, double dinv, int lane) {
    if constexpr (N <= DPP_ROW) return rows16_backsub_lane<N>(row, b, dinv, lane);
    else return rows_backsub_lane<N>(row, b, dinv, lane);
}

}  // namespace wv
}  // namespace tmpc
