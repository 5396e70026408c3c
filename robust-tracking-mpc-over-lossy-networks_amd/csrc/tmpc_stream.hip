// Streaming interior-point kernel for large batches: 16 lanes per QP (4 QPs per wavefront).
//
// solve_kernel (tmpc_kernels.hip) keeps one QP's row state in the registers of one wave; its
// per-iteration fixed costs (cross-lane sums, the nv x nv solve, dependent-latency chains with
// one wave per SIMD) are paid per QP and bound it at a few percent of the FP64 peak.  Here a QP
// owns a DPP row of 16 lanes instead:
//
//   * each lane walks 1/16 of the constraint rows in a plain loop; all per-row quantities are
//     recomputed per sweep from (s, lambda), which stream from an HBM/L2 workspace laid out
//     [instance][row] (16 consecutive doubles = one 128-byte line per group and load);
//   * every lane accumulates the full lower triangle of G'DG for its rows in registers (the 78 + 24
//     independent chains of NV = 12 give the FP64 pipe its instruction-level parallelism);
//   * the 16 partial sums meet through four DPP steps (quad_perm x2, row_half_mirror, row_mirror),
//     after which EVERY lane of the group holds M and factors it redundantly in registers: no LDS
//     round trip, no lane exchange on the factorisation's critical path;
//   * Gs, g0s, Es, Hs, Hs^-1 are staged once per workgroup in LDS.
//
// The kernel runs the Mehrotra iteration of tmpc_kernels.hip up to the hand-over point and leaves
// (z, s, lambda, status, iterations) in the workspace; solve_kernel is then launched in warm mode
// and performs the active-set refinement (and, should that fail, continues the iteration itself).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "tmpc_device.hpp"

namespace tmpc {

namespace {

constexpr int WAVE = 64;
constexpr int LPQ = 16;               // lanes per QP = one DPP row
constexpr int QPW = WAVE / LPQ;       // QPs per wave
constexpr int SWAVES = 4;             // waves per workgroup

template <int NV>
__host__ __device__ constexpr int col_off(int j) { return j * NV - j * (j - 1) / 2; }

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
// all-reduce over the 16 lanes of a DPP row
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_mov_d<0xB1>(v);
    v += dpp_mov_d<0x4E>(v);
    v += dpp_mov_d<0x141>(v);
    v += dpp_mov_d<0x140>(v);
    return v;
}
__device__ __forceinline__ double group_max(double v) {
    v = fmax(v, dpp_mov_d<0xB1>(v));
    v = fmax(v, dpp_mov_d<0x4E>(v));
    v = fmax(v, dpp_mov_d<0x141>(v));
    v = fmax(v, dpp_mov_d<0x140>(v));
    return v;
}
__device__ __forceinline__ double group_min(double v) { return -group_max(-v); }

// Cholesky in place on a column-major packed lower triangle ((i,j), i >= j, at col_off(j)+i-j);
// reciprocal pivots are left on the diagonal.
template <int NV>
__device__ __forceinline__ bool chol_cm(double (&M)[NV * (NV + 1) / 2]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        double v = M[col_off<NV>(j)];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= M[col_off<NV>(k) + j - k] * M[col_off<NV>(k) + j - k];
        ok = ok && (v > 0.0);
        const double inv = 1.0 / sqrt(v);
        M[col_off<NV>(j)] = inv;
#pragma unroll
        for (int i = j + 1; i < NV; ++i) {
            double t = M[col_off<NV>(j) + i - j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= M[col_off<NV>(k) + i - k] * M[col_off<NV>(k) + j - k];
            M[col_off<NV>(j) + i - j] = t * inv;
        }
    }
    return ok;
}
template <int NV>
__device__ __forceinline__ void solve_cm(const double (&L)[NV * (NV + 1) / 2], double (&b)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= L[col_off<NV>(k) + i - k] * b[k];
        b[i] = t * L[col_off<NV>(i)];
    }
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) {
        double t = b[i];
#pragma unroll
        for (int k = i + 1; k < NV; ++k) t -= L[col_off<NV>(i) + k - i] * b[k];
        b[i] = t * L[col_off<NV>(i)];
    }
}

template <int NV>
__global__ __launch_bounds__(WAVE *SWAVES, 1) void ipm_stream_kernel(
    const DeviceQP qp, const StreamQP sq, const int variant_id, const int64_t B,
    const double *__restrict__ x_k, const double *__restrict__ ref, const uint8_t *__restrict__ variant,
    double *__restrict__ ws_s, double *__restrict__ ws_lam, double *__restrict__ ws_z,
    int32_t *__restrict__ ws_stat, int32_t *__restrict__ ws_iters) {
    constexpr int NT = NV * (NV + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ncp = sq.ncp, nx = qp.nx, nc = qp.nc;
    double *Gt = smem;                      // [NV][ncp]
    double *g0 = Gt + NV * ncp;             // [ncp]
    double *Es = g0 + ncp;                  // [ncp][nx]
    double *Hs = Es + ncp * nx;             // [NV][NV]
    double *Hinv = Hs + NV * NV;            // [NV][NV]
    double *wbase = Hinv + NV * NV;         // per wave: x_k|ref [QPW][32], scratch vector [QPW][2*NV]

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid >> 6;
    for (int i = tid; i < NV * ncp; i += blockDim.x) Gt[i] = sq.Gd[i];
    for (int i = tid; i < ncp; i += blockDim.x) g0[i] = sq.g0d[i];
    for (int i = tid; i < ncp * nx; i += blockDim.x) Es[i] = sq.Esd[i];
    for (int i = tid; i < NV * NV; i += blockDim.x) { Hs[i] = qp.Hs[i]; Hinv[i] = qp.Hinv[i]; }
    __syncthreads();

    const int grp = lane / LPQ, gl = lane % LPQ;
    double *xin = wbase + (wave * QPW + grp) * (32 + 2 * NV);   // [2*nx] x_k | ref
    double *vtmp = xin + 32;                                    // [2*NV]

    const int64_t b = (static_cast<int64_t>(blockIdx.x) * SWAVES + wave) * QPW + grp;
    const bool mine = b < B && (variant ? variant[b] == variant_id : variant_id == 0);
    const int64_t bb = mine ? b : 0;

    // ---------------------------------------------------------------- per-instance data
    if (gl < nx) { xin[gl] = x_k[bb * nx + gl]; xin[nx + gl] = ref[bb * nx + gl]; }
    asm volatile("" ::: "memory");
    bool infeasible_par = qp.always_infeasible != 0;
    for (int r = gl; r < qp.npar; r += LPQ) {
        double v = qp.gp0[r];
        for (int c = 0; c < nx; ++c) v += qp.Ep[r * nx + c] * xin[c];
        if (v < -1e-9 * (1.0 + fabs(qp.gp0[r]))) infeasible_par = true;
    }
    infeasible_par = group_max(infeasible_par ? 1.0 : 0.0) > 0.5;
    if (gl < NV) {
        double v = 0.0;
        if (gl < qp.nv)
            for (int c = 0; c < nx; ++c) v += qp.F1s[gl * nx + c] * xin[c] + qp.F2s[gl * nx + c] * xin[nx + c];
        vtmp[gl] = v;
    }
    asm volatile("" ::: "memory");
    double q[NV], z[NV];
    double qn = 1.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) { q[j] = vtmp[j]; qn = fmax(qn, fabs(q[j])); }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) v -= Hinv[i * NV + j] * q[j];
        z[i] = v;
    }
    const int rpg = (nc + LPQ - 1) / LPQ;
    double *S = ws_s + bb * ncp, *LAM = ws_lam + bb * ncp;
    double hn = 1.0, smin = INFINITY;
    for (int t = 0; t < rpg; ++t) {
        const int r = gl + t * LPQ;
        if (r < nc) {
            double hr = g0[r];
            for (int c = 0; c < nx; ++c) hr += Es[r * nx + c] * xin[c];
            double gz = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) gz += Gt[j * ncp + r] * z[j];
            const double sr = hr - gz;
            hn = fmax(hn, fabs(hr));
            smin = fmin(smin, sr);
            if (mine) S[r] = sr;
        }
    }
    hn = group_max(hn);
    smin = group_min(smin);

    int st = TMPC_STATUS_MAX_ITER;      // hand-over codes: 0 refine, 1 cap, 2 infeasible, 3 numerical, 4 unconstrained optimum
    int it = 0;
    bool done = !mine;
    if (infeasible_par) { st = TMPC_STATUS_INFEASIBLE; done = true; }
    else if (smin >= 0.0) { st = 4; done = true; }
    if (!done) {
        const double fl = 0.1 * fmax(-smin, 1.0);
        for (int t = 0; t < rpg; ++t) {
            const int r = gl + t * LPQ;
            if (r < nc) { S[r] = fmax(S[r], fl); LAM[r] = 1.0; }
        }
    }
    const double ncd = static_cast<double>(nc);
    const double try_tol = qp.tol;
    while (!done) {
        if (it >= qp.max_iter) { st = TMPC_STATUS_MAX_ITER; break; }
        // nothing in this loop stores to LDS, so without a barrier the loads of Hs (144 values for
        // NV = 12) are treated as loop-invariant, hoisted and kept in registers across the loop
        asm volatile("" ::: "memory");
        // ---- sweep A: residuals, G'DG, G'(d.rp), G'lam, gap
        double acc[NT + 2 * NV];
#pragma unroll
        for (int i = 0; i < NT + 2 * NV; ++i) acc[i] = 0.0;
        double gap = 0.0, rpn = 0.0, lmax = 0.0;
        {
        // (s, lambda) of the next row are requested before this row's arithmetic: one wave per SIMD
        // has nothing else to hide the L2/HBM latency behind
        double s_nx = (gl < nc) ? S[gl] : 1.0, l_nx = (gl < nc) ? LAM[gl] : 0.0;
#pragma unroll 1
        for (int t = 0; t < rpg; ++t) {
            const int r = gl + t * LPQ;
            const double sr = s_nx, lr = l_nx;
            if (r + LPQ < nc) { s_nx = S[r + LPQ]; l_nx = LAM[r + LPQ]; }
            if (r < nc) {
                double hr = g0[r];
                for (int c = 0; c < nx; ++c) hr += Es[r * nx + c] * xin[c];
                double g[NV];
                double gz = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) { g[j] = Gt[j * ncp + r]; gz += g[j] * z[j]; }
                const double rp = gz + sr - hr;
                const double d = lr * fast_rcp(sr);
                const double tt = d * rp;
                gap += sr * lr;
                rpn = fmax(rpn, fabs(rp));
                lmax = fmax(lmax, lr);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const double dg = d * g[j];
#pragma unroll
                    for (int i = j; i < NV; ++i) acc[col_off<NV>(j) + i - j] += dg * g[i];
                    acc[NT + j] += g[j] * tt;
                    acc[NT + NV + j] += g[j] * lr;
                }
            }
        }
        }
#pragma unroll
        for (int i = 0; i < NT + 2 * NV; ++i) acc[i] = group_sum(acc[i]);
        gap = group_sum(gap);
        rpn = group_max(rpn);
        lmax = group_max(lmax);
        const double mu = gap / ncd;
        // cost gradient, residual norms, objective
        asm volatile("" ::: "memory");
        double cg[NV];
        double rdn = 0.0, obj = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) v += Hs[i * NV + j] * z[j];
            cg[i] = v + q[i];
            rdn = fmax(rdn, fabs(cg[i] + acc[NT + NV + i]));
            obj += z[i] * (0.5 * v + q[i]);
        }
        if (!(mu == mu) || !(rdn == rdn)) { st = TMPC_STATUS_NUMERICAL; break; }
        const double objs = fmax(fabs(obj), 1.0);
        if ((rdn <= 1e3 * try_tol * qn) && (rpn <= try_tol * hn) && (gap <= try_tol * objs)) { st = 0; break; }
        if (gap <= 1e-15 * objs) { st = TMPC_STATUS_MAX_ITER; break; }
        if (lmax > 1e10) {
            double hl = 0.0;
            for (int t = 0; t < rpg; ++t) {
                const int r = gl + t * LPQ;
                if (r < nc) {
                    double hr = g0[r];
                    for (int c = 0; c < nx; ++c) hr += Es[r * nx + c] * xin[c];
                    hl += hr * LAM[r];
                }
            }
            hl = group_sum(hl);
            double gn = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) gn = fmax(gn, fabs(acc[NT + NV + j]));
            if (hl < 0.0 && gn <= 1e-6 * lmax) { st = TMPC_STATUS_INFEASIBLE; break; }
        }
        // ---- M = Hs + G'DG, Cholesky in place, predictor solve
        asm volatile("" ::: "memory");
        double rhs[NV], dza[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            rhs[j] = -cg[j] - acc[NT + j];
            dza[j] = rhs[j];
#pragma unroll
            for (int i = j; i < NV; ++i) acc[col_off<NV>(j) + i - j] += Hs[i * NV + j];
        }
        {
            double (&L)[NT] = reinterpret_cast<double (&)[NT]>(acc);
            if (!chol_cm<NV>(L)) { st = TMPC_STATUS_NUMERICAL; break; }
            solve_cm<NV>(L, dza);
        }
        // ---- sweep B: affine step statistics and the corrector's G' products
        double accb[2 * NV];
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) accb[i] = 0.0;
        double rho_aff = 0.0, sb1 = 0.0, sb2 = 0.0;
        {
        // (s, lambda) of the next row are requested before this row's arithmetic: one wave per SIMD
        // has nothing else to hide the L2/HBM latency behind
        double s_nx = (gl < nc) ? S[gl] : 1.0, l_nx = (gl < nc) ? LAM[gl] : 0.0;
#pragma unroll 1
        for (int t = 0; t < rpg; ++t) {
            const int r = gl + t * LPQ;
            const double sr = s_nx, lr = l_nx;
            if (r + LPQ < nc) { s_nx = S[r + LPQ]; l_nx = LAM[r + LPQ]; }
            if (r < nc) {
                double hr = g0[r];
                for (int c = 0; c < nx; ++c) hr += Es[r * nx + c] * xin[c];
                double g[NV];
                double gz = 0.0, gdz = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) { g[j] = Gt[j * ncp + r]; gz += g[j] * z[j]; gdz += g[j] * dza[j]; }
                const double rs = fast_rcp(sr);
                const double rp = gz + sr - hr;
                const double dsa = -rp - gdz;
                const double dla = -lr - lr * rs * dsa;
                rho_aff = fmax(rho_aff, fmax(-dsa * rs, -dla * fast_rcp(lr)));
                const double w = dsa * dla;
                sb1 += sr * dla + lr * dsa;
                sb2 += w;
                const double c1 = w * rs;
#pragma unroll
                for (int j = 0; j < NV; ++j) { accb[j] += g[j] * c1; accb[NV + j] += g[j] * rs; }
            }
        }
        }
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) accb[i] = group_sum(accb[i]);
        rho_aff = group_max(rho_aff);
        sb1 = group_sum(sb1);
        sb2 = group_sum(sb2);
        const double aaff = rho_aff > 1.0 ? 1.0 / rho_aff : 1.0;
        const double mu_aff = (gap + aaff * sb1 + aaff * aaff * sb2) / ncd;
        double sigma = mu_aff / mu;
        sigma = fmin(sigma * sigma * sigma, 1.0);
        const double smu = sigma * mu;
        double dz[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) dz[j] = rhs[j] + accb[j] - smu * accb[NV + j];
        {
            double (&L)[NT] = reinterpret_cast<double (&)[NT]>(acc);
            solve_cm<NV>(L, dz);
        }
        // ---- sweep D1: step length
        double om = (1.0 - aaff) * (1.0 - aaff);
        om = fmin(fmax(om, 1e-4), 1e-2);
        const double tau = 1.0 - om;
        double rho = 0.0;
        {
        // (s, lambda) of the next row are requested before this row's arithmetic: one wave per SIMD
        // has nothing else to hide the L2/HBM latency behind
        double s_nx = (gl < nc) ? S[gl] : 1.0, l_nx = (gl < nc) ? LAM[gl] : 0.0;
#pragma unroll 1
        for (int t = 0; t < rpg; ++t) {
            const int r = gl + t * LPQ;
            const double sr = s_nx, lr = l_nx;
            if (r + LPQ < nc) { s_nx = S[r + LPQ]; l_nx = LAM[r + LPQ]; }
            if (r < nc) {
                double hr = g0[r];
                for (int c = 0; c < nx; ++c) hr += Es[r * nx + c] * xin[c];
                double gz = 0.0, gdza = 0.0, gdz = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) { const double g = Gt[j * ncp + r]; gz += g * z[j]; gdza += g * dza[j]; gdz += g * dz[j]; }
                const double rs = fast_rcp(sr);
                const double rp = gz + sr - hr;
                const double dsa = -rp - gdza;
                const double dla = -lr - lr * rs * dsa;
                const double ds = -rp - gdz;
                const double rc = sr * lr + dsa * dla - smu;
                const double dl = -(rc + lr * ds) * rs;
                rho = fmax(rho, fmax(-ds * rs, -dl * fast_rcp(lr)));
            }
        }
        }
        rho = group_max(rho);
        const double alpha = rho > tau ? tau / rho : 1.0;
        // ---- sweep D2: apply the step (row quantities recomputed; cheaper than another 16 B/row of traffic)
        {
        // (s, lambda) of the next row are requested before this row's arithmetic: one wave per SIMD
        // has nothing else to hide the L2/HBM latency behind
        double s_nx = (gl < nc) ? S[gl] : 1.0, l_nx = (gl < nc) ? LAM[gl] : 0.0;
#pragma unroll 1
        for (int t = 0; t < rpg; ++t) {
            const int r = gl + t * LPQ;
            const double sr = s_nx, lr = l_nx;
            if (r + LPQ < nc) { s_nx = S[r + LPQ]; l_nx = LAM[r + LPQ]; }
            if (r < nc) {
                double hr = g0[r];
                for (int c = 0; c < nx; ++c) hr += Es[r * nx + c] * xin[c];
                double gz = 0.0, gdza = 0.0, gdz = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) { const double g = Gt[j * ncp + r]; gz += g * z[j]; gdza += g * dza[j]; gdz += g * dz[j]; }
                const double rs = fast_rcp(sr);
                const double rp = gz + sr - hr;
                const double dsa = -rp - gdza;
                const double dla = -lr - lr * rs * dsa;
                const double ds = -rp - gdz;
                const double rc = sr * lr + dsa * dla - smu;
                const double dl = -(rc + lr * ds) * rs;
                S[r] = sr + alpha * ds;
                LAM[r] = lr + alpha * dl;
            }
        }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) z[j] += alpha * dz[j];
        ++it;
    }
    if (mine) {
        if (gl < NV) {
            double zj = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) zj = (gl == j) ? z[j] : zj;
            ws_z[b * NV + gl] = zj;
        }
        if (gl == 0) { ws_stat[b] = st; ws_iters[b] = it; }
    }
}

template <int NV>
hipError_t launch_stream_one(const DeviceQP &qp, const StreamQP &sq, int variant_id, int64_t B, const double *x_k,
                             const double *ref, const uint8_t *variant, double *ws_s, double *ws_lam, double *ws_z,
                             int32_t *ws_stat, int32_t *ws_iters, hipStream_t stream) {
    const size_t lds = stream_lds_bytes(NV, sq.ncp, qp.nx);
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ipm_stream_kernel<NV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    const int64_t per_block = static_cast<int64_t>(SWAVES) * QPW;
    const int64_t blocks = (B + per_block - 1) / per_block;
    hipLaunchKernelGGL((ipm_stream_kernel<NV>), dim3(static_cast<unsigned>(blocks)), dim3(WAVE * SWAVES), lds, stream,
                       qp, sq, variant_id, B, x_k, ref, variant, ws_s, ws_lam, ws_z, ws_stat, ws_iters);
    return hipGetLastError();
}

}  // namespace

size_t stream_lds_bytes(int nvp, int ncp, int nx) {
    return sizeof(double) * (static_cast<size_t>(nvp) * ncp + ncp + static_cast<size_t>(ncp) * nx + 2 * nvp * nvp +
                             static_cast<size_t>(SWAVES) * QPW * (32 + 2 * nvp));
}

bool stream_supported(int nvp, int ncp, int nx) {
    return (nvp == 8 || nvp == 12) && nx <= 16 && stream_lds_bytes(nvp, ncp, nx) <= 160 * 1024;
}

hipError_t launch_stream(const DeviceQP &qp, const StreamQP &sq, int nvp, int variant_id, int64_t B, const double *x_k,
                         const double *ref, const uint8_t *variant, double *ws_s, double *ws_lam, double *ws_z,
                         int32_t *ws_stat, int32_t *ws_iters, hipStream_t stream) {
    if (nvp == 8) return launch_stream_one<8>(qp, sq, variant_id, B, x_k, ref, variant, ws_s, ws_lam, ws_z, ws_stat, ws_iters, stream);
    if (nvp == 12) return launch_stream_one<12>(qp, sq, variant_id, B, x_k, ref, variant, ws_s, ws_lam, ws_z, ws_stat, ws_iters, stream);
    return hipErrorInvalidValue;
}

}  // namespace tmpc
