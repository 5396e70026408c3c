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
// acc + (lane K of v's 16-lane row) * m in ONE instruction: v_fmac_f64 is a two-operand (VOP2) instruction on gfx90a and later
// and takes the DPP operand itself (`v_fmac_f64_dpp dst, src0, src1 row_newbcast:K`: dst += dpp(src0) * src1).  The compiler
// does not fuse a 64-bit DPP move into the multiply-add that consumes it (round 3 shipped v_mov_b64_dpp + v_fma_f64: two
// instructions and one more link in every dependent chain of the eliminations), hence inline assembly.
// Hazard (gfx9 family, software managed): a vector-ALU write of a VGPR needs two wait states before a DPP read of it, and the
// compiler's hazard recognizer does not look into inline assembly.  The statements are `asm volatile`, i.e. they stay in
// program order among themselves; NOPS > 0 puts `s_nop NOPS-1` in front where the DPP operand may have been written by the
// statement just before (the dependent chains of the substitutions), and the callers order their statements so that the
// other operands were written at least two instructions earlier (each routine says how).
// (-DTMPC_NO_DPP_FMAC: diagnostic builds with the compiler's two-instruction form)
template <int K, int NOPS>
__device__ __forceinline__ void fmac_bcast(double &acc, double v, double m) {
#if defined(TMPC_HOST_SIM) || defined(TMPC_NO_DPP_FMAC)
    acc = fma(row_bcast_d<K>(v), m, acc);
#else
    if constexpr (NOPS > 0)
        asm volatile("s_nop %3\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(v), "v"(m), "n"(NOPS - 1), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(m), "n"(K));
#endif
}
// the pivot broadcast in the same (volatile) instruction order as the updates: `v` may have been written by the update two
// statements back (last elimination steps), which the compiler cannot see
template <int K, int NOPS>
__device__ __forceinline__ double row_bcast_ordered(double v) {
#if defined(TMPC_HOST_SIM) || defined(TMPC_NO_DPP_FMAC)
    return row_bcast_d<K>(v);
#else
    double r;
    if constexpr (NOPS > 0)
        asm volatile("s_nop %2\n\tv_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(NOPS - 1), "n"(K));
    else
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
#endif
}
// pins a value: everything that produces `v` is issued before the (empty, volatile) statement, which in turn stays in front
// of the volatile statements that follow
__device__ __forceinline__ void pin(double &v) {
#if !defined(TMPC_HOST_SIM) && !defined(TMPC_NO_DPP_FMAC)
    asm volatile("" : "+v"(v));
#endif
}
// returns whether the matrix of lanes 0 .. N-1 had positive pivots (wave-uniform)
// Software pipelined: the update of column k+1 comes FIRST in step k, the next pivot is broadcast two updates later (the two
// wait states of the DPP read) and its reciprocal -- rcp and four dependent multiply-adds, the longest chain of a step -- runs
// while the remaining updates of step k issue.  Order of the other DPP reads: an update of step k reads a register written in
// step k-1, with at least the pivot broadcast and the multiplier's instructions in between.  The inputs of step 0 come from
// LDS loads (guarded by s_waitcnt) and, b, from the caller through pin().
template <int N>
__device__ __forceinline__ bool rows16_factor(double (&row)[N], double &b, double &dinv, int lane) {
    static_assert(N <= 16, "one matrix row per lane of a 16-lane DPP row");
    const int l16 = lane & 15;
    bool ok = true;
    pin(b);
    double pkk = row_bcast_ordered<0, 0>(row[0]);
    double pinv = fast_rcp(pkk);
    static_for_n<N>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        ok = ok && (pkk > 0.0);
        const double pinv_k = pinv;
        const double f = (l16 > k) ? row[k] * pinv_k : 0.0;
        const double nf = -f;
        constexpr int NU = N - 1 - k;              // columns still to update
        // columns k+1, k+2, k+3 (b takes a place when fewer are left), then the next pivot, then the rest
        static_for_n<(NU < 3 ? NU : 3)>([&](auto j_) {
            constexpr int j = k + 1 + decltype(j_)::value;
            fmac_bcast<k, 0>(row[j], row[j], nf);
        });
        if constexpr (NU < 3) fmac_bcast<k, 0>(b, b, nf);
        if constexpr (k + 1 < N) {
            // (two statements lie between the write of row[k+1] and this read of it -- the DPP hazard's two wait states -- except
            // in the last step but one, where only the update of b does: NU = 1 waits)
            pkk = row_bcast_ordered<k + 1, (NU == 1 ? 2 : 0)>(row[k + 1]);
            pinv = fast_rcp(pkk);
        }
        if constexpr (NU >= 3) {
            static_for_n<NU - 3>([&](auto j_) {
                constexpr int j = k + 4 + decltype(j_)::value;
                fmac_bcast<k, 0>(row[j], row[j], nf);
            });
            fmac_bcast<k, 0>(b, b, nf);
        }
        if (l16 > k) row[k] = f;
        if (l16 == k) dinv = pinv_k;
    });
    return __builtin_amdgcn_readfirstlane(static_cast<int>(ok)) != 0;
}
// The substitutions are ONE dependent chain on b -- every step reads through DPP what the step before wrote, which costs the
// two wait states whatever the instruction --: the fused form gains nothing there, and the compiler fills the wait states of
// its own two-instruction form (v_mov_b64_dpp + v_fma_f64) with the selects of the neighbouring steps.
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
// ---- 16 < N <= 32: TWO matrix rows per lane -- lane i of every 16-lane row holds rows i (`ra`) and i + 16 (`rb`) -- so that every
// pivot row sits on a lane of the SAME 16-lane row as the rows it updates and the DPP forms above apply: an update is one
// v_fmac_f64_dpp per row half (two per column while the pivot is among the first sixteen rows, one afterwards) where the
// one-row-per-lane form of round 3 needed two v_readlane_b32 and a multiply-add with a scalar operand, i.e. a scalar-register
// round trip in every link of every chain (N = 26: 1 053 instructions for the elimination against 605 fused updates).
// The matrix lives in LDS (rows of stride LDM: N entries, then 1 / d_i) before and after: `rows32_factor_solve` reads the rows,
// eliminates with the right-hand side `rhs` (LDS, [N]) carried along, leaves the factor in place of the matrix and the solution
// in `x` (LDS, [N]); `rows32_resolve` solves with the stored factor.  Lanes whose second row does not exist (i + 16 >= N) carry
// a copy of row 0 with multipliers forced to zero.
template <int N, int LDM>
__device__ __forceinline__ bool rows32_factor_solve(double *Mf, const double *rhs, double *x, int lane) {
    static_assert(N > 16 && N <= 32, "two matrix rows per lane");
    const int l16 = lane & 15;
    const bool hasb = l16 + 16 < N;
    const int ib = hasb ? l16 + 16 : 0;
    double ra[N], rb[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { ra[j] = Mf[l16 * LDM + j]; rb[j] = Mf[ib * LDM + j]; }
    double ba = rhs[l16], bb = hasb ? rhs[ib] : 0.0;
    double da = 1.0, db = 1.0;
    lds_fence();
    bool ok = true;
    pin(ba);
    pin(bb);
    static_for_n<N>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        constexpr bool inA = k < 16;
        constexpr int kl = k & 15;
        // (the pivot entry was written by the first updates of step k - 1; only the last steps have fewer than two statements behind it)
        double pkk;
        if constexpr (inA) pkk = row_bcast_ordered<kl, (k >= N - 2 ? 2 : 0)>(ra[k]);
        else pkk = row_bcast_ordered<kl, (k >= N - 2 ? 2 : 0)>(rb[k]);
        ok = ok && (pkk > 0.0);
        const double pinv = fast_rcp(pkk);
        double nfa = 0.0;
        if constexpr (inA) nfa = (l16 > k) ? -ra[k] * pinv : 0.0;
        const double nfb = (hasb && l16 + 16 > k) ? -rb[k] * pinv : 0.0;
        static_for_n<N - 1 - k>([&](auto j_) {
            constexpr int j = k + 1 + decltype(j_)::value;
            // (the lower half first: it reads the pivot row's entry of `ra` before the upper half's own update writes that register)
            if constexpr (inA) {
                fmac_bcast<kl, 0>(rb[j], ra[j], nfb);
                fmac_bcast<kl, 0>(ra[j], ra[j], nfa);
            } else {
                fmac_bcast<kl, 0>(rb[j], rb[j], nfb);
            }
        });
        if constexpr (inA) {
            fmac_bcast<kl, 0>(bb, ba, nfb);
            fmac_bcast<kl, 0>(ba, ba, nfa);
            if (l16 > k) ra[k] = -nfa;
            if (l16 == k) da = pinv;
        } else {
            fmac_bcast<kl, 0>(bb, bb, nfb);
            if (l16 == kl) db = pinv;
        }
        if (hasb && l16 + 16 > k) rb[k] = -nfb;
    });
    ok = __builtin_amdgcn_readfirstlane(static_cast<int>(ok)) != 0;
    if (!ok) return false;
    // back substitution (compiler's DPP form: one dependent chain, see rows16_backsub_lane); x_i ends on lane i & 15
    double xa = 0.0, xb = 0.0;
    static_for_n<N>([&](auto r_) {
        constexpr int i = N - 1 - decltype(r_)::value;
        constexpr int il = i & 15;
        if constexpr (i >= 16) {
            const double bi = bb * db;
            const double xi = row_bcast_d<il>(bi);
            xb = (l16 == il) ? bi : xb;
            bb = fma(-rb[i], xi, bb);
            ba = fma(-ra[i], xi, ba);
        } else {
            const double bi = ba * da;
            const double xi = row_bcast_d<il>(bi);
            xa = (l16 == il) ? bi : xa;
            ba = fma(-ra[i], xi, ba);
        }
    });
    if (lane < 16) {
        x[l16] = xa;
        if (hasb) x[ib] = xb;
        // the factor waits in LDS for the corrector's solve
#pragma unroll
        for (int j = 0; j < N; ++j) Mf[l16 * LDM + j] = ra[j];
        Mf[l16 * LDM + N] = da;
        if (hasb) {
#pragma unroll
            for (int j = 0; j < N; ++j) Mf[ib * LDM + j] = rb[j];
            Mf[ib * LDM + N] = db;
        }
    }
    lds_fence();
    return true;
}
template <int N, int LDM>
__device__ __forceinline__ void rows32_resolve(const double *Mf, const double *rhs, double *x, int lane) {
    const int l16 = lane & 15;
    const bool hasb = l16 + 16 < N;
    const int ib = hasb ? l16 + 16 : 0;
    double ra[N], rb[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { ra[j] = Mf[l16 * LDM + j]; rb[j] = Mf[ib * LDM + j]; }
    const double da = Mf[l16 * LDM + N], db = Mf[ib * LDM + N];
    double ba = rhs[l16], bb = hasb ? rhs[ib] : 0.0;
    lds_fence();
    // forward substitution with the stored multipliers
    static_for_n<N - 1>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        constexpr int kl = k & 15;
        const double fb = (hasb && l16 + 16 > k) ? rb[k] : 0.0;
        if constexpr (k < 16) {
            const double fa = (l16 > k) ? ra[k] : 0.0;
            const double xk = row_bcast_d<kl>(ba);
            ba = fma(-fa, xk, ba);
            bb = fma(-fb, xk, bb);
        } else {
            const double xk = row_bcast_d<kl>(bb);
            bb = fma(-fb, xk, bb);
        }
    });
    double xa = 0.0, xb = 0.0;
    static_for_n<N>([&](auto r_) {
        constexpr int i = N - 1 - decltype(r_)::value;
        constexpr int il = i & 15;
        if constexpr (i >= 16) {
            const double bi = bb * db;
            const double xi = row_bcast_d<il>(bi);
            xb = (l16 == il) ? bi : xb;
            bb = fma(-rb[i], xi, bb);
            ba = fma(-ra[i], xi, ba);
        } else {
            const double bi = ba * da;
            const double xi = row_bcast_d<il>(bi);
            xa = (l16 == il) ? bi : xa;
            ba = fma(-ra[i], xi, ba);
        }
    });
    if (lane < 16) {
        x[l16] = xa;
        if (hasb) x[ib] = xb;
    }
    lds_fence();
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
