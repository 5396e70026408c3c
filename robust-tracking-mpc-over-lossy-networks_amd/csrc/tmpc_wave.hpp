// Wave-level device helpers shared by the block kernel (tmpc_block.hip): DPP / readlane
// reductions over the 64 lanes of a gfx950 wavefront and a division-free reciprocal.
#pragma once
#include <hip/hip_runtime.h>

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
struct OpSum { __device__ __forceinline__ static double f(double a, double b) { return a + b; } };
struct OpMin { __device__ __forceinline__ static double f(double a, double b) { return fmin(a, b); } };
struct OpMax { __device__ __forceinline__ static double f(double a, double b) { return fmax(a, b); } };

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

}  // namespace wv
}  // namespace tmpc
