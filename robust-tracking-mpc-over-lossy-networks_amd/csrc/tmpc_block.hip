// Workgroup-per-QP solve kernel for gfx950 (MI355X): the general path for condensed QPs that do
// not fit the one-wave-per-QP kernel of tmpc_kernels.hip -- up to 128 decision variables, any
// number of inequality rows (cartpole N = 20, the extended controller's packet-received problem
// with its Z (-) W rows, the synthetic n = 12, m = 4, N = 30 model of BASELINE.json config 5).
//
// Same algorithm as tmpc_kernels.hip / oracle (unconstrained-minimiser
// shortcut, Mehrotra predictor-corrector interior point on the normal equations, active-set
// refinement by proximal Newton steps), different mapping to the machine:
//
//   * one workgroup per QP instance: 256 threads (4 waves, two workgroups per CU) up to 64 variables, 512 threads (8 waves = two per
//     SIMD, one workgroup per CU) for 65 .. 128; persistent workgroups, the first instance of each is its index in the grid, the
//     later ones are drawn from the launch's work counter;
//   * the per-row state (s, lambda, h, G z and four arrays shared by the quantities of an iteration: enum WS_* below) lives in a
//     per-workgroup workspace slice in HBM/L2, laid out [quantity][row]: thread t owns rows t, t + BT, ...; every access is a
//     coalesced stream and the register footprint does not depend on the number of rows;
//   * when every constraint row has its mirror row (box-type sets: all of BASELINE's models) a row of G is a FUNCTIONAL serving
//     both sides (BlockQP::mir): the three G-sized passes of an iteration read half the rows;
//   * M = Hs + G'DG -- nc*nv^2 of the ~nc*nv^2 + nv^3/3 flops of an iteration -- is formed with
//     v_mfma_f64_16x16x4_f64: A = (d .* G)' and B = G are read straight from the row-major copy of G
//     (one f64 per lane and k-step, 4 x 128 B segments per load), the 16x16 tiles of the lower
//     triangle are dealt to the waves as tile rows g and T-1-g, the structural zeros of the condensed rows (stage k acts on
//     u_0 .. u_k only) are skipped by tile (BlockQP::row_start); for small nv the waves also split the rows and add their
//     partial tiles in LDS;
//   * M is factored in LDS by the whole workgroup (blocked right-looking Cholesky: 16 x 16 diagonal blocks in the registers of
//     wave 0, panel by thread per row, trailing update on the matrix cores), L^-1 is formed explicitly (W' in the upper
//     triangle of the same LDS matrix) and the two solves of an iteration are four matrix-vector products by all threads;
//   * G'v products (two vectors per pass) use thread-per-(column pair, row part) over the row-major
//     copy, with Hs z riding along in the pass of the dual residual; G v products thread-per-row-pair over the column-major
//     copy; G z itself is carried along incrementally (G z += alpha G dz).
#ifdef TMPC_HOST_SIM
#include "hip_sim.hpp"      // tests/wavesim: this very source compiled for the CPU under sanitizers (never in the product)
#else
#include <hip/hip_runtime.h>
#endif

#include <atomic>
#include <cmath>
#include <cstdint>
#include <type_traits>

#include "tmpc_device.hpp"
#include "tmpc_wave.hpp"

namespace tmpc {

namespace {

using namespace wv;

// The model record (BlockArgs) is read through the constant address space: every field is a scalar load at its point of use
// (s_load from the scalar cache), not a register that has to survive the kernel.  ARGS_REFRESH() at the phase boundaries hands
// the compiler an opaque copy of the record's address and of the workspace base, so that nothing derived from them is hoisted
// out of the instance loop and kept alive across phases (the same device as TMPC_REFRESH() in tmpc_kernels.hip).
// per-launch arguments of solve_block_kernel
struct BlockLaunch {
    const BlockArgs *args;
    long long *ticks, *dbg;
    double *ws;
    int variant_id;
    int64_t B;
    const double *x_k, *ref;
    const uint8_t *variant;
    double *u_nom, *x_nom0, *xu_ss, *x_nom;
    int32_t *status, *iters;
    unsigned long long *next_item;      // work counter of this launch (zero at its start): instances beyond the grid's first round
};

#ifdef TMPC_HOST_SIM
typedef const BlockLaunch *LaunchPtr;
typedef const BlockArgs *ArgsPtr;
typedef const BlockQP CBlockQP;
typedef const DeviceQP CDeviceQP;
#define ARGS_REFRESH() do { } while (0)
inline double fresh_value(double v) { return v; }
#else
typedef const __attribute__((address_space(4))) BlockLaunch *LaunchPtr;
typedef const __attribute__((address_space(4))) BlockArgs *ArgsPtr;
typedef const __attribute__((address_space(4))) BlockQP CBlockQP;
typedef const __attribute__((address_space(4))) DeviceQP CDeviceQP;
// (through a vector register and v_readfirstlane: the compiler takes phase boundaries behind LDS-derived conditions for divergent
// control flow and would not keep an "s"-constrained value in scalar registers there)
__device__ __forceinline__ int fresh_lane(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ double fresh_value(double v) { asm volatile("" : "+v"(v)); return v; }
template <class P>
__device__ __forceinline__ P fresh_uniform(P p) {
    unsigned lo = static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)), hi = static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return reinterpret_cast<P>((static_cast<uintptr_t>(hi) << 32) | lo);
}
#define ARGS_REFRESH() do { tid = fresh_lane(tid_k); lane = tid & (WAVE - 1); wave = tid >> 6; ap = (ArgsPtr)fresh_uniform(reinterpret_cast<const BlockArgs *>((uintptr_t)ap)); W0 = fresh_uniform(W0); lp = (LaunchPtr)fresh_uniform(reinterpret_cast<const BlockLaunch *>((uintptr_t)lp)); } while (0)
#endif

// threads per workgroup: 256 (4 waves, two workgroups per CU where the LDS allows) up to 64 variables; 512 (8 waves, one
// workgroup per CU = two waves per SIMD) for the 128-variable shape, whose LDS footprint admits one workgroup only
constexpr int block_threads(int tiles) { return tiles >= 8 ? 512 : 256; }
typedef double v4d __attribute__((ext_vector_type(4)));

// rows of the per-workgroup workspace, each [ncp] doubles
// Per-row state of a workgroup in its workspace slice: EIGHT arrays of ncp doubles.  Round 3 kept sixteen (164 KB per workgroup for
// config 5: 32 workgroups of an XCD and the two copies of G overflowed its 4 MB L2, and every pass wrote its arrays through to
// HBM -- 9.4 GB per launch of 16384 instances).  Quantities whose lifetimes within an iteration do not overlap share an array;
// each hand-over is a read and a write of the SAME row by the SAME thread (the statement order in p5_row / p7_row):
//   WS_D   d = lambda / s (P1 .. P5)           then w = ds_aff * dl_aff (P5 .. P7)
//   WS_V1  d * r_p (P1 .. P2)                  then w / s (P5 .. P6)            then dl (P7 .. P8)
//   WS_RS  1 / s (P5 .. P7)                    then ds (P7 .. P8)
//   WS_RP  r_p (P1 .. P7)                      then G dz (P7 .. P8)
// and the refinement's arrays (multipliers, row residuals, membership flags) take the places of r_p, d and d * r_p, which the
// interior-point phase forms anew (P1) when it continues after a refinement that did not certify.
enum { WS_S, WS_LAM, WS_H, WS_GZ, WS_RP, WS_D, WS_V1, WS_RS, WS_COUNT,
       WS_W = WS_D, WS_C1 = WS_V1, WS_DL = WS_V1, WS_DS = WS_RS, WS_GDZ = WS_RP, WS_Y = WS_RP, WS_RR = WS_D, WS_INW = WS_V1 };

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int T>
struct BShape {
    static constexpr int NVP = 16 * T;
    static constexpr int BT = block_threads(T);
    static constexpr int BW = BT / WAVE;                             // waves per workgroup
    static constexpr int LDM = NVP + 1;                              // odd stride: conflict-free column walks
    static constexpr int WCAP = NVP >= 128 ? 128 : NVP + 16;         // working-set rows of the refinement
    static constexpr int LDSS = WCAP + 1;
    static constexpr int BIG = cmax(NVP * LDM, WCAP * LDSS);         // M / its factor, later S / its factor
    static constexpr int PARTS = 2 * BT / NVP;                       // row parts of a G'v pass (a thread owns two columns)
    static constexpr int G = T >= 2 ? T / 2 : 1;                     // tile groups of the MFMA pass
    static constexpr int RSPLIT = BW / G;                            // row parts of the MFMA pass
    static constexpr int NVEC = 10;                                  // nv-vectors
    static constexpr int SMALL = NVEC * NVP + 4 * BT + 2 * WCAP + cmax(NVP, WCAP) + WCAP /*Widx as ints, padded*/ + 32 + 48;
    static constexpr int TOTAL = BIG + SMALL;
    // resident workgroups per CU the LDS footprint allows (capped at 4) = waves per SIMD the register budget is set for
    static constexpr int OCC = (160 * 1024 / 8) / TOTAL >= 2 ? 2 : 1;
};

template <int BW, class OpA, class OpB, class OpC>
__device__ __forceinline__ void block_reduce3(double &a, double &b, double &c, double *red, int wave, int lane) {
    a = wave_reduce<OpA>(a);
    b = wave_reduce<OpB>(b);
    c = wave_reduce<OpC>(c);
    __syncthreads();
    if (lane == 0) { red[wave] = a; red[BW + wave] = b; red[2 * BW + wave] = c; }
    __syncthreads();
    a = red[0]; b = red[BW]; c = red[2 * BW];
#pragma unroll
    for (int w = 1; w < BW; ++w) { a = OpA::f(a, red[w]); b = OpB::f(b, red[BW + w]); c = OpC::f(c, red[2 * BW + w]); }
}
template <int BW, class Op>
__device__ __forceinline__ double block_reduce1(double a, double *red, int wave, int lane) {
    a = wave_reduce<Op>(a);
    __syncthreads();
    if (lane == 0) red[wave] = a;
    __syncthreads();
    a = red[0];
#pragma unroll
    for (int w = 1; w < BW; ++w) a = Op::f(a, red[w]);
    return a;
}

// G_row(r) . v for the thread's row: column-major copy, v in LDS (broadcast reads).  Eight loads in flight per
// step: the operands come from L2, the loop is bound by how many of them are outstanding.
__device__ __forceinline__ double row_dot(const double *__restrict__ Gcm, int ncp, int nv, int r, const double *v, CBlockQP &bq) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (r < bq.nz4) {                      // initial-state row: znx columns only
        for (int a = 0; a < bq.znx; ++a) t0 = fma(Gcm[static_cast<size_t>(bq.zx0 + a) * ncp + r], v[bq.zx0 + a], t0);
        return t0;
    }
    nv = bq.ncols[r];                      // the row is zero beyond (staircase, BlockQP::ncols)
    int j = 0;
    for (; j + 8 <= nv; j += 8) {
        double g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = Gcm[static_cast<size_t>(j + k) * ncp + r];
        t0 = fma(g[0], v[j], t0); t1 = fma(g[1], v[j + 1], t1); t2 = fma(g[2], v[j + 2], t2); t3 = fma(g[3], v[j + 3], t3);
        t0 = fma(g[4], v[j + 4], t0); t1 = fma(g[5], v[j + 5], t1); t2 = fma(g[6], v[j + 6], t2); t3 = fma(g[7], v[j + 7], t3);
    }
    for (; j < nv; ++j) t0 = fma(Gcm[static_cast<size_t>(j) * ncp + r], v[j], t0);
    return (t0 + t1) + (t2 + t3);
}

// The same for the row pair (r, r + 1), r even: 16-byte loads (an 8-byte access runs at 0.54 - 0.70 of the 16-byte rate, and
// the row passes sit at the CU's L1 rate).  Initial-state rows go through row_dot.
__device__ __forceinline__ void row_dot2(const double *__restrict__ Gcm, int ncp, int nv, int r, const double *v, CBlockQP &bq,
                                         double &o0, double &o1) {
    if (r < bq.nz4) { o0 = row_dot(Gcm, ncp, nv, r, v, bq); o1 = row_dot(Gcm, ncp, nv, r + 1, v, bq); return; }
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int n0 = bq.ncols[r], n1 = bq.ncols[r + 1];
    nv = n0 > n1 ? n0 : n1;                 // (rows are ordered by their reach: the two mostly agree; the shorter one reads zeros)
    const v2d *__restrict__ G2 = reinterpret_cast<const v2d *>(Gcm + r);
    const size_t st = static_cast<size_t>(ncp) / 2;
    v2d t0 = {0.0, 0.0}, t1 = {0.0, 0.0}, t2 = {0.0, 0.0}, t3 = {0.0, 0.0};
    int j = 0;
    for (; j + 8 <= nv; j += 8) {
        v2d g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = G2[static_cast<size_t>(j + k) * st];
        t0 += g[0] * v[j]; t1 += g[1] * v[j + 1]; t2 += g[2] * v[j + 2]; t3 += g[3] * v[j + 3];
        t0 += g[4] * v[j + 4]; t1 += g[5] * v[j + 5]; t2 += g[6] * v[j + 6]; t3 += g[7] * v[j + 7];
    }
    for (; j < nv; ++j) t0 += G2[static_cast<size_t>(j) * st] * v[j];
    const v2d t = (t0 + t1) + (t2 + t3);
    o0 = t.x; o1 = t.y;
}

// sum_j Hm[j][c] v[j] + sum_k R[idx[k]][c] w[k] for column c = tid (valid for tid < NVP), by all threads: thread
// (c, part) takes every (BT / NVP)-th term, eight loads in flight, the parts meet in LDS.  Hm (symmetric, [NVP][NVP]) or R
// (rows of NVP) may be null.  With the columns alone (128 threads, four loads in flight) a 128-variable product costs
// 32 dependent L2 round trips.
template <int T>
__device__ __noinline__ double column_sums(const double *__restrict__ Hm, const double *v, const double *__restrict__ R, const int *idx,
                                              const double *w, int m, double *parts, int tid) {
    constexpr int NVP = BShape<T>::NVP, BT = BShape<T>::BT, P = BT / NVP;
    const int c = tid % NVP, part = tid / NVP;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // Hm: thread (column pair, row part), 16-byte loads, eight of the thread's rows in flight (the matrix comes from L2: the
    // product is two round trips for T = 8 instead of four with half the requests)
    constexpr int NV2 = NVP / 2, PH = BT / NV2, RPT = (NVP + PH - 1) / PH;      // row parts, rows per thread
    typedef double v2d __attribute__((ext_vector_type(2)));
    if (Hm != nullptr) {
        const int c2 = tid % NV2, ph = tid / NV2;
        const v2d *__restrict__ H2 = reinterpret_cast<const v2d *>(Hm) + c2;
        constexpr int CH = RPT > 8 ? 8 : RPT;          // rows in flight per batch (T = 8: two batches of eight)
        static_assert(RPT % CH == 0, "whole batches");
        v2d s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
        for (int u0 = 0; u0 < RPT; u0 += CH) {
            v2d hv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = ph + (u0 + u) * PH;
                hv[u] = j < NVP ? H2[static_cast<size_t>(j) * NV2] : v2d{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = ph + (u0 + u) * PH;
                const double vj = j < NVP ? v[j] : 0.0;
                if (u & 1) s1 += hv[u] * vj; else s0 += hv[u] * vj;
            }
        }
        s0 += s1;
        __syncthreads();                   // previous readers of `parts` are done
        parts[P * NVP + ph * NVP + 2 * c2] = s0.x;
        parts[P * NVP + ph * NVP + 2 * c2 + 1] = s0.y;
    }
    if (R != nullptr) {
        int k = part;
        for (; k + 3 * P < m; k += 4 * P) {
            const double r0 = R[static_cast<size_t>(idx[k]) * NVP + c], r1 = R[static_cast<size_t>(idx[k + P]) * NVP + c];
            const double r2 = R[static_cast<size_t>(idx[k + 2 * P]) * NVP + c], r3 = R[static_cast<size_t>(idx[k + 3 * P]) * NVP + c];
            a0 = fma(r0, w[k], a0); a1 = fma(r1, w[k + P], a1); a2 = fma(r2, w[k + 2 * P], a2); a3 = fma(r3, w[k + 3 * P], a3);
        }
        for (; k < m; k += P) a0 = fma(R[static_cast<size_t>(idx[k]) * NVP + c], w[k], a0);
    }
    if (Hm == nullptr) __syncthreads();    // previous readers of `parts` are done
    parts[part * NVP + c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    double sum = 0.0;
    if (tid < NVP) {
#pragma unroll
        for (int p2 = 0; p2 < P; ++p2) sum += parts[p2 * NVP + tid];
        if (Hm != nullptr) {
#pragma unroll
            for (int p2 = 0; p2 < PH; ++p2) sum += parts[P * NVP + p2 * NVP + tid];
        }
    }
    return sum;
}

// out_a = G' va, out_b = G' vb (LDS vectors of NVP entries); va, vb are per-row workspace arrays.
// zext != nullptr: Hs zext is added to BOTH -- the NVP rows of the (symmetric) scaled Hessian sit behind the rows of G in Grm and
// ride along in this pass, which streams at the L1 rate, instead of costing a product of their own (four dependent L2 round
// trips with a handful of loads in flight: 5 % of an instance's time in round 2's form).
template <int T>
__device__ __noinline__ void gt_products(const double *__restrict__ Grm, int nc, const double *va, const double *vb,
                                            double *parts, double *out_a, double *out_b, int tid, CBlockQP &bq, const double *zext = nullptr) {
    constexpr int NVP = BShape<T>::NVP, PARTS = BShape<T>::PARTS, BT = BShape<T>::BT, NV2 = NVP / 2;
    typedef double v2d __attribute__((ext_vector_type(2)));
    // thread = (column pair, row part): 16-byte loads, a wave covers whole rows of G; the operands come from L2 or beyond
    // and the pass is bound by the bytes in flight, so eight rows are outstanding per thread
    const int j2 = tid % NV2, part = tid / NV2;
    v2d a0 = {0.0, 0.0}, b0 = {0.0, 0.0}, a1 = {0.0, 0.0}, b1 = {0.0, 0.0};
    const v2d *__restrict__ G2 = reinterpret_cast<const v2d *>(Grm) + j2;
    const int rs = bq.row_start[(2 * j2) >> 4];   // rows below do not reach this column tile (staircase): not loaded
    const v2d zero2 = {0.0, 0.0};
    const int mir = bq.mir;                // != 0: row r of G is a functional, its weight is va[r] - va[r + mir] (BlockQP)
    auto wt = [&](const double *w, int row) { return mir != 0 ? w[row] - w[row + mir] : w[row]; };
    int r = bq.nz4 + part;                 // general rows; the initial-state rows [0, nz4) follow below
    if constexpr (NV2 >= WAVE) {
        // a wave = one row part (T = 8): sixteen rows per trip, their weights va / vb fetched by sixteen lanes in one gather each
        // and handed round by v_readlane (a load per row and lane would triple the vector-memory instructions of the pass)
        for (; r + 15 * PARTS < nc; r += 16 * PARTS) {
            const int rl = r + (tid & 15) * PARTS;
            const double xal = wt(va, rl), xbl = wt(vb, rl);
            v2d g[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) g[k] = (r + k * PARTS >= rs) ? G2[static_cast<size_t>(r + k * PARTS) * NV2] : zero2;
            // (the weight of row k is lane k's of every 16-lane row: the DPP operand of the multiply-add itself, wv::fmac_bcast --
            // round 3 brought it to scalar registers with two v_readlane_b32 per weight first: 8 instead of 4 instructions per row)
            double xad = xal, xbd = xbl;
            wv::pin(xad);
            wv::pin(xbd);
            double a0x = a0.x, a0y = a0.y, b0x = b0.x, b0y = b0.y, a1x = a1.x, a1y = a1.y, b1x = b1.x, b1y = b1.y;
            wv::static_for_n<8>([&](auto k2_) {
                constexpr int k = 2 * decltype(k2_)::value;
                wv::fmac_bcast<k, (k == 0 ? 2 : 0)>(a0x, xad, g[k].x);
                wv::fmac_bcast<k, (k == 0 ? 2 : 0)>(b0x, xbd, g[k].x);
                wv::fmac_bcast<k, 0>(a0y, xad, g[k].y);
                wv::fmac_bcast<k, 0>(b0y, xbd, g[k].y);
                wv::fmac_bcast<k + 1, 0>(a1x, xad, g[k + 1].x);
                wv::fmac_bcast<k + 1, 0>(b1x, xbd, g[k + 1].x);
                wv::fmac_bcast<k + 1, 0>(a1y, xad, g[k + 1].y);
                wv::fmac_bcast<k + 1, 0>(b1y, xbd, g[k + 1].y);
            });
            a0 = v2d{a0x, a0y}; b0 = v2d{b0x, b0y}; a1 = v2d{a1x, a1y}; b1 = v2d{b1x, b1y};
        }
    } else {
        for (; r + 7 * PARTS < nc; r += 8 * PARTS) {           // several row parts per wave: eight rows in flight, weights per lane
            v2d g[8];
            double xa[8], xb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                g[k] = (r + k * PARTS >= rs) ? G2[static_cast<size_t>(r + k * PARTS) * NV2] : zero2;
                xa[k] = wt(va, r + k * PARTS);
                xb[k] = wt(vb, r + k * PARTS);
            }
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                a0 += g[k] * xa[k]; b0 += g[k] * xb[k];
                a1 += g[k + 1] * xa[k + 1]; b1 += g[k + 1] * xb[k + 1];
            }
        }
    }
    for (; r < nc; r += PARTS) {
        const v2d g0 = (r >= rs) ? G2[static_cast<size_t>(r) * NV2] : zero2;
        a0 += g0 * wt(va, r); b0 += g0 * wt(vb, r);
    }
    if (zext != nullptr) {
        const size_t hs0 = static_cast<size_t>(bq.ngp);          // first row of Hs in Grm
        constexpr int HR = (NVP + PARTS - 1) / PARTS, CH = HR > 8 ? 8 : HR;       // rows of Hs per thread, rows in flight
        static_assert(HR % CH == 0, "whole batches");
        v2d hs = {0.0, 0.0};
#pragma unroll
        for (int u0 = 0; u0 < HR; u0 += CH) {
            v2d hg[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = part + (u0 + u) * PARTS;
                hg[u] = j < NVP ? G2[(hs0 + j) * NV2] : zero2;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = part + (u0 + u) * PARTS;
                hs += hg[u] * (j < NVP ? zext[j] : 0.0);
            }
        }
        a0 += hs; b0 += hs;
    }
    __syncthreads();                       // previous readers of `parts` are done
    a0 += a1; b0 += b1;
    parts[part * NVP + 2 * j2] = a0.x;
    parts[part * NVP + 2 * j2 + 1] = a0.y;
    parts[2 * BT + part * NVP + 2 * j2] = b0.x;
    parts[2 * BT + part * NVP + 2 * j2 + 1] = b0.y;
    __syncthreads();
    if (tid < NVP) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) { sa += parts[p * NVP + tid]; sb += parts[2 * BT + p * NVP + tid]; }
        out_a[tid] = sa;
        out_b[tid] = sb;
    }
    __syncthreads();
    if (bq.nz4 > 0) {
        // initial-state rows: znx (<= 16) columns; the 256 threads are zpad column slots x (256 / zpad) row parts, four
        // rows in flight per thread
        const int zpad = bq.znx <= 4 ? 4 : (bq.znx <= 8 ? 8 : 16), ZP = BT / zpad;
        const int a = tid % zpad, rp2 = tid / zpad;
        double sa0 = 0.0, sb0 = 0.0, sa1 = 0.0, sb1 = 0.0;
        if (a < bq.znx) {
            const double *gcol = Grm + bq.zx0 + a;
            int r2 = rp2;
            for (; r2 + 3 * ZP < bq.nz4; r2 += 4 * ZP) {
                const double g0 = gcol[static_cast<size_t>(r2) * NVP], g1 = gcol[static_cast<size_t>(r2 + ZP) * NVP];
                const double g2 = gcol[static_cast<size_t>(r2 + 2 * ZP) * NVP], g3 = gcol[static_cast<size_t>(r2 + 3 * ZP) * NVP];
                const double x0 = va[r2], x1 = va[r2 + ZP], x2 = va[r2 + 2 * ZP], x3 = va[r2 + 3 * ZP];
                const double y0 = vb[r2], y1 = vb[r2 + ZP], y2 = vb[r2 + 2 * ZP], y3 = vb[r2 + 3 * ZP];
                sa0 = fma(g0, x0, sa0); sb0 = fma(g0, y0, sb0); sa1 = fma(g1, x1, sa1); sb1 = fma(g1, y1, sb1);
                sa0 = fma(g2, x2, sa0); sb0 = fma(g2, y2, sb0); sa1 = fma(g3, x3, sa1); sb1 = fma(g3, y3, sb1);
            }
            for (; r2 < bq.nz4; r2 += ZP) {
                const double g = gcol[static_cast<size_t>(r2) * NVP];
                sa0 = fma(g, va[r2], sa0);
                sb0 = fma(g, vb[r2], sb0);
            }
        }
        parts[tid] = sa0 + sa1;
        parts[BT + tid] = sb0 + sb1;
        __syncthreads();
        if (tid < bq.znx) {
            double ta = 0.0, tb = 0.0;
            for (int q = 0; q < ZP; ++q) { ta += parts[q * zpad + tid]; tb += parts[BT + q * zpad + tid]; }
            out_a[bq.zx0 + tid] += ta;
            out_b[bq.zx0 + tid] += tb;
        }
        __syncthreads();
    }
}

// Tiles of the lower triangle of G'DG in the tile rows RA < RB (RA = -1: RB only), accumulated over the k-steps (4 rows
// each) rpart, rpart + RSPLIT, ... and added into M (LDS).  The rows are ordered by the tiles they reach (staircase of the
// condensed constraints, BlockQP::row_start): tile row t takes k-steps from ks_t on, so the pass runs [ksA, ksB) for
// the tiles of row RA alone and [ksB, nsteps) for both rows.
template <int T, int RA, int RB>
__device__ __forceinline__ void gdg_group(const double *__restrict__ Grm, const double *__restrict__ dvec, int ksA, int ksB, int nsteps,
                                          int rpart, double *M, int lane, int mir, const double *__restrict__ Hsg, double shift) {
    using SH = BShape<T>;
    constexpr int NVP = SH::NVP, LDM = SH::LDM, RS = SH::RSPLIT;
    constexpr int NA = RA + 1, NB = RB + 1;
    constexpr int NAA = NA > 0 ? NA : 1;
    v4d accA[NAA], accB[NB];
#pragma unroll
    for (int t = 0; t < NAA; ++t) accA[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NB; ++t) accB[t] = v4d{0.0, 0.0, 0.0, 0.0};
    const int kq = lane >> 4, c = lane & 15;
    // U k-steps per group, the next group's operands in flight while the current group's MFMAs run:
    // the loads come from L2 (~1-2 k cycles away), one k-step of look-ahead would leave the matrix core idle
    constexpr int U = T >= 4 ? 2 : 4;
    auto pass = [&](auto with_b, int ks_begin, int ks_end) {
        constexpr bool WB = decltype(with_b)::value;
        constexpr int NL = WB ? NB : NAA;                 // tiles of a row of G this pass reads
        double cur[U][NL], nxt[U][NL];
        double dcur[U], dnxt[U];
        auto load_group = [&](int ksg, double (&g)[U][NL], double (&dv)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = ksg + u * RS;
                const bool in = ks < ks_end;
                const size_t row = static_cast<size_t>(4 * (in ? ks : ks_begin) + kq);
#pragma unroll
                for (int t = 0; t < NL; ++t) g[u][t] = Grm[row * NVP + 16 * t + c];
                // (mir != 0: the row is a functional, its weight the sum of its two sides' -- BlockQP)
                dv[u] = in ? (mir != 0 ? dvec[row] + dvec[row + mir] : dvec[row]) : 0.0;
            }
        };
        int ks = ks_begin + rpart;
        if (ks_begin >= ks_end) return;
        load_group(ks, cur, dcur);                  // unconditional (indices are clamped inside): a branch around the
        for (; ks < ks_end; ks += RS * U) {         // prefetch makes the compiler wait for ALL loads before the MFMAs
            load_group(ks + RS * U, nxt, dnxt);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (WB) {
                    const double aB = dcur[u] * cur[u][RB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) accB[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB, cur[u][t], accB[t], 0, 0, 0);
                }
                if constexpr (NA > 0) {
                    const double aA = dcur[u] * cur[u][RA];
#pragma unroll
                    for (int t = 0; t < NA; ++t) accA[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA, cur[u][t], accA[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int t = 0; t < NL; ++t) cur[u][t] = nxt[u][t];
                dcur[u] = dnxt[u];
            }
        }
    };
    if constexpr (NA > 0) pass(std::false_type{}, ksA, ksB);
    pass(std::true_type{}, ksB, nsteps);
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    // The first row part of a tile row WRITES M = Hs + (shift on the diagonal) + its tiles, the others add theirs: the entries of Hs come
    // straight from L2 with all of a lane's loads in flight (round 3 copied the whole NVP x NVP matrix into LDS first, a phase of
    // its own between two barriers: 6 % of an instance).  Only the lower tile triangle is written: nothing reads M above it before
    // the inverse puts W' there.
    if (rpart == 0) {
        double hB[NB][4], hA[NAA][4];
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) hB[t][reg] = Hsg[(16 * RB + kq + 4 * reg) * NVP + 16 * t + c];
        if constexpr (NA > 0) {
#pragma unroll
            for (int t = 0; t < NA; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) hA[t][reg] = Hsg[(16 * RA + kq + 4 * reg) * NVP + 16 * t + c];
        }
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                M[(16 * RB + kq + 4 * reg) * LDM + 16 * t + c] = hB[t][reg] + accB[t][reg] + ((t == RB && kq + 4 * reg == c) ? shift : 0.0);
        if constexpr (NA > 0) {
#pragma unroll
            for (int t = 0; t < NA; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    M[(16 * RA + kq + 4 * reg) * LDM + 16 * t + c] = hA[t][reg] + accA[t][reg] + ((t == RA && kq + 4 * reg == c) ? shift : 0.0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 1; p < RS; ++p) {
        if (rpart == p) {
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) M[(16 * RB + kq + 4 * reg) * LDM + 16 * t + c] += accB[t][reg];
            if constexpr (NA > 0) {
#pragma unroll
                for (int t = 0; t < NA; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) M[(16 * RA + kq + 4 * reg) * LDM + 16 * t + c] += accA[t][reg];
            }
        }
        __syncthreads();
    }
}

// M += Gz' D Gz for the initial-state rows [0, nz4): they touch the znx columns of x_0 only, so their share of G'DG is a
// znx x znx block (10 entries for the cart-pole instead of a pass of all 16-wide tiles over 850 rows).  Thread (pair p of
// the lower triangle, row part q) sums d_r g_ra g_rb over its rows; the parts meet in LDS.
template <int T>
__device__ __forceinline__ void zblock_accumulate(const double *__restrict__ Grm, const double *__restrict__ dvec, CBlockQP &bq,
                                                  double *M, double *parts, int tid) {
    if (bq.nz4 <= 0) return;
    constexpr int NVP = BShape<T>::NVP, LDM = BShape<T>::LDM, BT = BShape<T>::BT;
    const int P = bq.znx * (bq.znx + 1) / 2;           // <= 136
    const int Q = BT / P;
    const int p = tid % P, q = tid / P;
    int a = 0, rem = p;
    while (rem > a) { rem -= a + 1; ++a; }             // p -> (a, b), b <= a
    const int b = rem;
    double t0 = 0.0, t1 = 0.0;
    if (q < Q) {
        int r = q;
        for (; r + 3 * Q < bq.nz4; r += 4 * Q) {            // four rows in flight
            const double *g0 = Grm + static_cast<size_t>(r) * NVP + bq.zx0, *g1 = Grm + static_cast<size_t>(r + Q) * NVP + bq.zx0;
            const double *g2 = Grm + static_cast<size_t>(r + 2 * Q) * NVP + bq.zx0, *g3 = Grm + static_cast<size_t>(r + 3 * Q) * NVP + bq.zx0;
            const double d0 = dvec[r], d1 = dvec[r + Q], d2 = dvec[r + 2 * Q], d3 = dvec[r + 3 * Q];
            const double a0 = g0[a], b0 = g0[b], a1 = g1[a], b1 = g1[b], a2 = g2[a], b2 = g2[b], a3 = g3[a], b3 = g3[b];
            t0 = fma(d0 * a0, b0, t0); t1 = fma(d1 * a1, b1, t1);
            t0 = fma(d2 * a2, b2, t0); t1 = fma(d3 * a3, b3, t1);
        }
        for (; r + Q < bq.nz4; r += 2 * Q) {
            const double *g0 = Grm + static_cast<size_t>(r) * NVP + bq.zx0, *g1 = Grm + static_cast<size_t>(r + Q) * NVP + bq.zx0;
            t0 = fma(dvec[r] * g0[a], g0[b], t0);
            t1 = fma(dvec[r + Q] * g1[a], g1[b], t1);
        }
        if (r < bq.nz4) { const double *g0 = Grm + static_cast<size_t>(r) * NVP + bq.zx0; t0 = fma(dvec[r] * g0[a], g0[b], t0); }
    }
    __syncthreads();
    parts[tid] = t0 + t1;
    __syncthreads();
    if (tid < P) {
        double v = 0.0;
        for (int k = 0; k < Q; ++k) v += parts[k * P + tid];
        M[(bq.zx0 + a) * LDM + bq.zx0 + b] += v;
    }
    __syncthreads();
}

template <int T>
__device__ __forceinline__ void gdg_all(const double *__restrict__ Grm, const double *__restrict__ dvec, CBlockQP &bq, int nsteps,
                                        double *M, int wave, int lane, const double *__restrict__ Hsg, double shift) {
    constexpr int G = BShape<T>::G;
    const int g = wave % G, rpart = wave / G;
    const int ks0 = bq.nz4 / 4;                    // k-steps below belong to the initial-state rows (handled apart)
    const int mir = bq.mir;
    auto first = [&](int t) { const int k = bq.row_start[t] / 4; return k > ks0 ? k : ks0; };
    if constexpr (G == 1) {
        if constexpr (T == 1) gdg_group<T, -1, 0>(Grm, dvec, ks0, ks0, nsteps, rpart, M, lane, mir, Hsg, shift);
        else gdg_group<T, 0, 1>(Grm, dvec, first(0), first(1), nsteps, rpart, M, lane, mir, Hsg, shift);
    } else if constexpr (G == 2) {
        if (g == 0) gdg_group<T, 0, 3>(Grm, dvec, first(0), first(3), nsteps, rpart, M, lane, mir, Hsg, shift);
        else gdg_group<T, 1, 2>(Grm, dvec, first(1), first(2), nsteps, rpart, M, lane, mir, Hsg, shift);
    } else {
        if (g == 0) gdg_group<T, 0, 7>(Grm, dvec, first(0), first(7), nsteps, rpart, M, lane, mir, Hsg, shift);
        else if (g == 1) gdg_group<T, 1, 6>(Grm, dvec, first(1), first(6), nsteps, rpart, M, lane, mir, Hsg, shift);
        else if (g == 2) gdg_group<T, 2, 5>(Grm, dvec, first(2), first(5), nsteps, rpart, M, lane, mir, Hsg, shift);
        else gdg_group<T, 3, 4>(Grm, dvec, first(3), first(4), nsteps, rpart, M, lane, mir, Hsg, shift);
    }
}

#ifdef TMPC_STAMPS
#define ISTAMP(p) do { if (tph_) { __syncthreads(); long long now_ = __builtin_amdgcn_s_memtime(); tph_[p] += now_ - *tlast_; *tlast_ = now_; } } while (0)
#define ISTAMP_ARGS , long long *tph_ = nullptr, long long *tlast_ = nullptr
#else
#define ISTAMP(p) do { } while (0)
#define ISTAMP_ARGS
#endif
// Blocked right-looking Cholesky of the n x n lower triangle at Mx (LDS, odd row stride ld) by the whole workgroup, 16
// columns per step:
//   1. wave 0 factors the 16 x 16 diagonal block in registers (lane i holds row i; the pivot row reaches the other
//      lanes by v_readlane, no LDS round trips in the 16-column chain);
//   2. one thread per row below solves its row of the panel against the block (L21 = A21 L11^-T);
//   3. the trailing lower-triangular tiles take A22 -= L21 L21' on the matrix cores (v_mfma_f64_16x16x4_f64, four
//      k-steps per tile), tiles dealt round-robin to the waves.
// On return the strictly lower triangle of Mx holds L, dinv[j] = 1 / L[j][j].  Returns false (uniformly) on a
// non-positive pivot.  Rows and columns >= n are never read as data (masked to zero / identity).
template <int BWn>
__device__ __forceinline__ bool block_chol(double *Mx, int ld, int n, double *dinv, double *piv, int tid ISTAMP_ARGS) {
    const int lane = tid & (WAVE - 1), wave = tid >> 6;
    const int nblk = (n + 15) >> 4;
    const int li = lane & 15, kq = lane >> 4;
    if (tid == 0) piv[0] = 1.0;
    for (int kb = 0; kb < nblk; ++kb) {
        const int c0 = 16 * kb;
        const int nl = n - c0 < 16 ? n - c0 : 16;
        __syncthreads();
        if (wave == 0) {
            double a[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (li < nl && k <= li) ? Mx[(c0 + li) * ld + c0 + k] : (k == li ? 1.0 : 0.0);
            bool good = true;
            // Column j of L scaled, then a[k] -= a[j] * (a[j] of lane k) for k > j: the broadcast of lane k's entry is the DPP
            // operand of the multiply-add itself (v_fmac_f64_dpp ... row_newbcast:k, wv::fmac_bcast) -- round 3 fetched it with two
            // v_readlane_b32 into scalar registers first: three instructions and a scalar round trip per update of a chain that the
            // other seven waves of the workgroup wait for.  (The first update of a step follows the write of a[j]: two wait states.)
            wv::static_for_n<16>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                double pj = readlane_d(a[j], j);
                if (!(pj > 0.0)) { good = false; pj = 1.0; }
                double inv = __builtin_amdgcn_rsq(pj);        // 1 / sqrt(p_j): v_rsq_f64 + three Newton steps
#pragma unroll
                for (int nr = 0; nr < 3; ++nr) inv = fma(0.5 * inv, fma(-pj * inv, inv, 1.0), inv);
                a[j] *= inv;
                if (lane == 0 && j < nl) dinv[c0 + j] = inv;
                double naj = -a[j];
                wv::pin(naj);          // (a[j] and its negative are in place before the updates, which stay in this order)
                wv::static_for_n<15 - j>([&](auto k_) {
                    constexpr int k = j + 1 + decltype(k_)::value;
                    if constexpr (k == j + 1) wv::fmac_bcast<k, 2>(a[k], a[j], naj);
                    else wv::fmac_bcast<k, 0>(a[k], a[j], naj);
                });
            });
            if (lane < nl) {
#pragma unroll
                for (int k = 0; k < 15; ++k)
                    if (k < lane) Mx[(c0 + lane) * ld + c0 + k] = a[k];
            }
            if (!good && lane == 0) piv[0] = 0.0;
        }
        __syncthreads();
        ISTAMP(15);
        if (piv[0] == 0.0) break;
        if (kb + 1 == nblk) break;
        // panel: row r of A21 against L11 (forward substitution along the row)
        for (int r = c0 + 16 + tid; r < n; r += BWn * WAVE) {
            double x[16];
            double *row = Mx + r * ld + c0;
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = row[k];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double *lj = Mx + (c0 + j) * ld + c0;
                double v = x[j];
#pragma unroll
                for (int k = 0; k < j; ++k) v = fma(-x[k], lj[k], v);
                x[j] = v * dinv[c0 + j];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k] = x[k];
        }
        __syncthreads();
        ISTAMP(3);
        // trailing update of the tiles (ta, tb), tb <= ta, below / right of the panel
        const int mt = nblk - kb - 1, ntile = mt * (mt + 1) / 2, base = c0 + 16;
        for (int t = wave; t < ntile; t += BWn) {
            int ta = 0, rem = t;
            while (rem > ta) { rem -= ta + 1; ++ta; }
            const int tb = rem;
            const int ri = base + 16 * ta + li, rj = base + 16 * tb + li;
            v4d acc;
            double *ct = Mx + (base + 16 * ta + kq) * ld + base + 16 * tb + li;
            const double *pa = Mx + (ri < n ? ri : 0) * ld + c0 + kq, *pb = Mx + (rj < n ? rj : 0) * ld + c0 + kq;
            double av[4], bv[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { av[s4] = pa[4 * s4]; bv[s4] = pb[4 * s4]; }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[reg] = ct[4 * reg * ld];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ri < n ? -av[s4] : 0.0, rj < n ? bv[s4] : 0.0, acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ct[4 * reg * ld] = acc[reg];
        }
        ISTAMP(5);
    }
    __syncthreads();
    const bool ok = piv[0] != 0.0;
    __syncthreads();
    return ok;
}

// W = L^-1 for the factor left by block_chol, by the whole workgroup; W' goes to the strictly UPPER triangle of Mx
// (W[i][k], k < i, at Mx[k * ld + i]; its diagonal is dinv), the strictly lower triangle is used up.  With W the two
// triangular solves of an iteration -- 2 x 2 n dependent steps in one wave -- become two matrix-vector products by all
// threads (block_inv_solve).  Three steps on 16 x 16 tiles:
//   I.   the diagonal blocks: lane c of a wave solves L11 x = e_c (forward substitution, L11 read as LDS broadcasts);
//   II.  the tiles below the diagonal are scaled by their row's diagonal inverse, Lt_ik = W_ii L_ik (MFMA);
//   III. block diagonal d = 1, 2, ..: W_ij = - sum_{k = j}^{i-1} Lt_ik W_kj (MFMA), i - j = d; W_kj, k - j < d, is complete.
template <int BWn>
__device__ __forceinline__ void block_invert(double *Mx, int ld, int n, const double *dinv, int tid ISTAMP_ARGS) {
    const int lane = tid & (WAVE - 1), wave = tid >> 6;
    const int nblk = (n + 15) >> 4;
    const int li = lane & 15, kq = lane >> 4;
    // W'[a][b] for a < b (storage Mx[a * ld + b]), with the diagonal from dinv and zero below; indices >= n act as
    // identity.  Branch-free: both loads are issued with clamped addresses, the selects follow.
    auto wt = [&](int a, int b2) -> double {
        const bool in = a < n && b2 < n;
        const int ac = in ? a : 0, bc = in ? b2 : 0;
        const double up = Mx[ac * ld + bc], dg = dinv[ac];
        const double v = a < b2 ? up : (a == b2 ? dg : 0.0);
        return in ? v : (a == b2 ? 1.0 : 0.0);
    };
    for (int kb = wave; kb < nblk; kb += BWn) {                       // I
        const int c0 = 16 * kb;
        const int nl = n - c0 < 16 ? n - c0 : 16;
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double v = (j == li) ? 1.0 : 0.0;
            if (j < nl) {
                const double *lj = Mx + (c0 + j) * ld + c0;
#pragma unroll
                for (int k = 0; k < j; ++k) v = fma(-lj[k], x[k], v);
                v *= dinv[c0 + j];
            }
            x[j] = v;
        }
        if (lane < nl) {
#pragma unroll
            for (int j = 1; j < 16; ++j)
                if (j > lane && j < nl) Mx[(c0 + lane) * ld + c0 + j] = x[j];
        }
    }
    __syncthreads();
    ISTAMP(12);
    {                                                                 // II
        const int ntile = nblk * (nblk - 1) / 2;
        for (int t = wave; t < ntile; t += BWn) {
            int ti = 1, rem = t;
            while (rem >= ti) { rem -= ti; ++ti; }
            const int tk = rem;                                       // tile (ti, tk), tk < ti
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            double av[4], bv[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int qq = 4 * s4 + kq;
                const int rowb = 16 * ti + qq;
                av[s4] = wt(rowb, 16 * ti + li);                      // W_ii[m = li][q] = W'[q][m]
                const double lv = Mx[(rowb < n ? rowb : 0) * ld + 16 * tk + li];
                bv[s4] = rowb < n ? lv : 0.0;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Mx[(16 * ti + kq + 4 * reg) * ld + 16 * tk + li] = acc[reg];
        }
    }
    __syncthreads();
    ISTAMP(13);
    for (int d = 1; d < nblk; ++d) {                                  // III
        for (int ti = d + wave; ti < nblk; ti += BWn) {
            const int tj = ti - d;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            const int rowa = 16 * ti + li;
            const bool aok = rowa < n;                                // (only the last block row can be partial)
            const double *arow = Mx + (aok ? rowa : 0) * ld + kq;     // Lt_ik[m = li][q] at arow[16 k + q]
            const double *brow = Mx + (16 * tj + li) * ld + kq;       // W_kj[q][n = li] = W'[n][q] at brow[16 k + q], k > j
            double av[4], bv[4], an[4], bn[4];
            const double dj = dinv[16 * tj + li];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {                          // k = j: the diagonal block W_jj (lower triangular)
                const int qq = 4 * s4 + kq;
                const double lv = arow[16 * tj + 4 * s4], up = brow[16 * tj + 4 * s4];
                av[s4] = aok ? lv : 0.0;
                bv[s4] = li < qq ? up : (li == qq ? dj : 0.0);
            }
            for (int tk = tj; tk < ti; ++tk) {
                const int tn = tk + 1 < ti ? tk + 1 : ti - 1;         // the next product's operands while this one runs
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const double lv = arow[16 * tn + 4 * s4];
                    an[s4] = aok ? lv : 0.0;
                    bn[s4] = brow[16 * tn + 4 * s4];
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { av[s4] = an[s4]; bv[s4] = bn[s4]; }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * ti + kq + 4 * reg;               // W_ij[row][col] -> W'[col][row]
                if (row < n) Mx[(16 * tj + li) * ld + row] = -acc[reg];
            }
        }
        __syncthreads();
    }
}

// x = W' (W b) = (L L')^-1 b with W from block_invert: two matrix-vector products, thread (row, part) with BTn / 128
// parts per row, partial sums through `parts` (>= 5 * 128 doubles).  b and x may alias.  Entries n..nfill-1 of x are cleared.
template <int BTn>
__device__ __noinline__ void block_inv_solve(const double *Mx, int ld, int n, const double *dinv, const double *b, double *x,
                                                double *parts, int tid, int nfill = 0) {
    constexpr int Q = BTn / 128;
    const int i = tid & 127, q = tid >> 7;
    double *yv = parts + Q * 128;
    {
        double a0 = 0.0, a1 = 0.0;
        if (i < n) {
            int k = q;
            for (; k + Q < i; k += 2 * Q) { a0 = fma(Mx[k * ld + i], b[k], a0); a1 = fma(Mx[(k + Q) * ld + i], b[k + Q], a1); }
            if (k < i) a0 = fma(Mx[k * ld + i], b[k], a0);
        }
        parts[q * 128 + i] = a0 + a1;
    }
    __syncthreads();
    if (tid < n) {
        double v = dinv[tid] * b[tid];
#pragma unroll
        for (int p = 0; p < Q; ++p) v += parts[p * 128 + tid];
        yv[tid] = v;
    }
    __syncthreads();
    {
        double a0 = 0.0, a1 = 0.0;
        if (i < n) {
            const double *row = Mx + i * ld;
            int k = i + 1 + q;
            for (; k + Q < n; k += 2 * Q) { a0 = fma(row[k], yv[k], a0); a1 = fma(row[k + Q], yv[k + Q], a1); }
            if (k < n) a0 = fma(row[k], yv[k], a0);
        }
        parts[q * 128 + i] = a0 + a1;
    }
    __syncthreads();
    if (tid < n) {
        double v = dinv[tid] * yv[tid];
#pragma unroll
        for (int p = 0; p < Q; ++p) v += parts[p * 128 + tid];
        x[tid] = v;
    } else if (tid < nfill) {
        x[tid] = 0.0;
    }
    __syncthreads();
}

// Diagnostic build only (-DTMPC_STAMPS): per-phase cycle counts of workgroup 0, written to dbg_arg
#ifdef TMPC_STAMPS
#define BSTAMP(p) do { ARGS_REFRESH(); __syncthreads(); long long now_ = __builtin_amdgcn_s_memtime(); tph[p] += now_ - tlast; tlast = now_; } while (0)
#else
#define BSTAMP(p) do { ARGS_REFRESH(); } while (0)      // phase boundary: nothing derived from the model record or the workspace base survives it
#endif

template <int T>
__global__ __launch_bounds__(BShape<T>::BT, BShape<T>::OCC) void solve_block_kernel(const BlockLaunch la) {
    // The launch record is read where it is used, through the kernel-argument segment itself (constant address space), never
    // through `la`: as by-value parameters its fifteen fields were loaded in the prologue and kept for the whole kernel.
#ifdef TMPC_HOST_SIM
    LaunchPtr lp = &la;
#else
    LaunchPtr lp = (LaunchPtr)__builtin_amdgcn_kernarg_segment_ptr();
#endif
    const BlockArgs *const args = lp->args;

    using SH = BShape<T>;
    constexpr int NVP = SH::NVP, LDM = SH::LDM, WCAP = SH::WCAP, LDSS = SH::LDSS, BT = SH::BT;
#ifdef TMPC_HOST_SIM
    ArgsPtr ap = args;
#else
    ArgsPtr ap = (ArgsPtr)args;             // global -> constant address space: the record is read-only for every launch
#endif
#define qp (ap->qp)
#define bq (ap->bq)
#ifdef TMPC_HOST_SIM
    double *smem = sim::lds<double>();
#else
    extern __shared__ __attribute__((aligned(16))) double smem[];
#endif
    double *big = smem;
    double *qv = big + SH::BIG;          // linear term
    double *zv = qv + NVP;               // z
    double *cgv = zv + NVP;              // cost gradient, later the corrector's right-hand side
    double *rhsv = cgv + NVP;            // predictor right-hand side
    double *dzav = rhsv + NVP;           // affine direction
    double *dzv = dzav + NVP;            // final direction
    double *zpv = dzv + NVP;             // refinement iterate
    double *tv = zpv + NVP;              // scratch
    double *uv = tv + NVP;               // scratch
    double *glv = uv + NVP;              // G' lambda
    double *parts = glv + NVP;           // [2][2 BT] partial sums of a G'v pass
    double *yv = parts + 4 * BT;         // [WCAP]
    double *dyv = yv + WCAP;             // [WCAP]
    double *dinv = dyv + WCAP;           // [max(NVP, WCAP)] reciprocal pivots
    int *Widx = reinterpret_cast<int *>(dinv + cmax(NVP, WCAP));   // [WCAP] (ints in a WCAP-double slot)
    double *xin = reinterpret_cast<double *>(Widx) + WCAP;          // [32] x_k | ref
    double *red = xin + 32;                                          // [48] reductions (3 x 8 waves) | pivot | ints
    int *ibc = reinterpret_cast<int *>(red + 40);                    // a few ints

    // (tid and what follows from it -- LDS addresses, the masks of `tid < n` compares -- are re-derived at the phase boundaries too)
    const int tid_k = threadIdx.x;
    int tid = tid_k, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int nx = qp.nx, nu = qp.nu, N = qp.N, nv = qp.nv, nc = qp.nc, ncp = bq.ncp;
    // Rows of G and rows of the problem (BlockQP): with mir != 0 a row of G is a FUNCTIONAL g, serving the constraint rows r
    // (g'z <= h_r) and r + mir (-g'z <= h_{r+mir}); every pass over G then runs over ng = nc / 2 rows.  mir == 0: ng == nc.
    const int mir = bq.mir, ng = bq.ng, ngp = bq.ngp;
    const double *__restrict__ Grm = bq.Grm;
    const double *__restrict__ Gcm = bq.Gcm;
    const double *__restrict__ GHrm = bq.GHrm;
    const double *__restrict__ Gw = bq.Gw;
    const int nsteps = (ng + 3) / 4;
    auto valid_row = [&](int r) { return ((mir != 0 && r >= mir) ? r - mir : r) < ng; };
    // G v for every row: pairs of rows of G (16-byte loads, row_dot2), the mirror rows take the negated product
    auto row_products = [&](const double *vec, auto &&body) {
        for (int f2 = 2 * tid; f2 < ngp; f2 += 2 * BT) {
            double p0, p1;
            row_dot2(Gcm, ngp, nv, f2, vec, bq, p0, p1);
            body(f2, p0);
            body(f2 + 1, p1);
            if (mir != 0) { body(f2 + mir, -p0); body(f2 + 1 + mir, -p1); }
        }
    };

    // the workspace arrays are offsets of W0, formed where they are used (sixteen live pointers in round 2)
    double *W0 = lp->ws + static_cast<size_t>(blockIdx.x) * WS_COUNT * ncp;
#define WSP(k) (W0 + static_cast<size_t>(k) * ncp)
#define s_ WSP(WS_S)
#define lam_ WSP(WS_LAM)
#define h_ WSP(WS_H)
#define gz_ WSP(WS_GZ)
#define rp_ WSP(WS_RP)
#define d_ WSP(WS_D)
#define v1_ WSP(WS_V1)
#define w_ WSP(WS_W)
#define c1_ WSP(WS_C1)
#define rs_ WSP(WS_RS)
#define ds_ WSP(WS_DS)
#define dl_ WSP(WS_DL)
#define gdz_ WSP(WS_GDZ)
#define yall_ WSP(WS_Y)
#define rr_ WSP(WS_RR)
#define inW_ (reinterpret_cast<int *>(WSP(WS_INW)))

    // Work distribution: the first instance of a workgroup is its index in the grid, the later ones are drawn from the launch's
    // counter.  An instance costs anything between the set-up alone (the unconstrained minimiser is feasible: a quarter of
    // config 5) and a dozen iterations; with the static stride of round 2 the launch ended with the unluckiest of 256 sums of 64
    // such times, a quarter above their mean.
    long long *const draw = reinterpret_cast<long long *>(red + 44);
    bool first_item = true;
    for (;;) {
        int64_t b;
        if (first_item) {
            b = blockIdx.x;
            first_item = false;
        } else {
            ARGS_REFRESH();
            __syncthreads();            // (every thread has read the previous draw)
            if (tid == 0) draw[0] = static_cast<long long>(gridDim.x) + static_cast<long long>(atomicAdd(lp->next_item, 1ull));
            __syncthreads();
            b = draw[0];
        }
        if (b >= lp->B) break;
        { const uint8_t *const variant = lp->variant; const int variant_id = lp->variant_id;
          if (variant != nullptr && variant[b] != variant_id) continue;
        if (variant == nullptr && variant_id != 0) continue; }
        __syncthreads();
        if (tid < nx) { xin[tid] = lp->x_k[b * nx + tid]; xin[16 + tid] = lp->ref[b * nx + tid]; }
        __syncthreads();
        int st = TMPC_STATUS_MAX_ITER;
        int it_done = 0;
        const long long t_begin = lp->ticks ? static_cast<long long>(__builtin_amdgcn_s_memrealtime()) : 0;
#ifdef TMPC_STAMPS
        long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_amdgcn_s_memtime();
#endif

        int bad = qp.always_infeasible != 0;
        for (int r = tid; r < qp.npar; r += BT) {
            double v = qp.gp0[r];
            for (int c = 0; c < nx; ++c) v += qp.Ep[r * nx + c] * xin[c];
            if (v < -1e-9 * (1.0 + fabs(qp.gp0[r]))) bad = 1;
        }
        bad = __syncthreads_or(bad);

        double qn_l = 1.0;
        if (tid < NVP) {
            double v = 0.0;
            if (tid < nv)
                for (int c = 0; c < nx; ++c) v += qp.F1s[tid * nx + c] * xin[c] + qp.F2s[tid * nx + c] * xin[16 + c];
            qv[tid] = v;
            qn_l = fmax(qn_l, fabs(v));
        }
        double hn_l = 1.0;
        for (int r = tid; r < ncp; r += BT) {
            double v = bq.g0[r];
            for (int c = 0; c < nx; ++c) v += bq.Es[static_cast<size_t>(r) * nx + c] * xin[c];
            h_[r] = v;
            if (valid_row(r)) hn_l = fmax(hn_l, fabs(v));
        }
        __syncthreads();
        // z = -Hinv q (Hinv symmetric: column read = coalesced)
        {
            const double v = column_sums<T>(qp.Hinv, qv, nullptr, nullptr, nullptr, 0, parts, tid);
            if (tid < NVP) zv[tid] = -v;
        }
        __syncthreads();
        double smin_l = INFINITY;
        row_products(zv, [&](int r, double gzr) {
            const double sv = h_[r] - gzr;
            gz_[r] = gzr;
            s_[r] = sv;
            lam_[r] = 0.0;
            if (valid_row(r)) smin_l = fmin(smin_l, sv);
        });
        block_reduce3<SH::BW, OpMax, OpMax, OpMin>(qn_l, hn_l, smin_l, red, wave, lane);
        const double qn = qn_l, hn = hn_l, smin = smin_l;
        BSTAMP(0);

        if (bad) {
            st = TMPC_STATUS_INFEASIBLE;
        } else if (smin >= 0.0) {
            st = TMPC_STATUS_OPTIMAL;
        } else {
            // ------------------------------------------------------------ interior point
            {
                const double fl = 0.1 * fmax(-smin, 1.0);
                // starting multipliers: the fourth root of the largest single-row multiplier viol_r / (g_r Hs^-1 g_r') over the
                // violated rows, between 1 and 1e3 (tmpc_kernels.hip has the reasoning; the oracle starts alike)
                double l1 = 0.0;
                for (int r = tid; r < ncp; r += BT)
                    if (valid_row(r)) l1 = fmax(l1, -s_[r] * bq.ci[r]);
                l1 = block_reduce1<SH::BW, OpMax>(l1, red, wave, lane);
                const double lam0 = fmin(fmax(sqrt(sqrt(l1)), 1.0), 1e3);
                for (int r = tid; r < ncp; r += BT) {
                    const bool valid = valid_row(r);
                    s_[r] = valid ? fmax(s_[r], fl) : 1.0;
                    lam_[r] = valid ? lam0 : 0.0;
                }
            }
            double try_tol = qp.tol;
            const double ncd = static_cast<double>(nc);
            int it = 0;
            double rdn_last = 0.0;
            bool floor_tried = false;
            for (;;) {
                bool want_polish = false;
                for (; it < qp.max_iter; ++it) {
                    it_done = it;
                    // ---- P1: residuals and scalings per row
                    double gap = 0.0, rpn = 0.0, lmax = 0.0, gzl = 0.0;
                    for (int r = tid; r < ncp; r += BT) {
                        const bool valid = valid_row(r);
                        const double sv = s_[r], lv = lam_[r];
                        const double gzr = gz_[r];
                        const double rp = gzr + sv - h_[r];
                        gzl = fma(gzr, lv, gzl);
                        const double rs = valid ? fast_rcp(sv) : 0.0;
                        const double d = lv * rs;
                        rp_[r] = rp;
                        d_[r] = d;
                        v1_[r] = d * rp;
                        gap += sv * lv;
                        rpn = fmax(rpn, fabs(rp));
                        lmax = fmax(lmax, lv);
                    }
                    block_reduce3<SH::BW, OpSum, OpMax, OpMax>(gap, rpn, lmax, red, wave, lane);
                    gzl = block_reduce1<SH::BW, OpSum>(gzl, red, wave, lane);          // (G z)' lam
                    const double mu = gap / ncd;
                    BSTAMP(1);
                    // ---- P2: dual residual and predictor right-hand side.  Hs z rides along in the pass over G (gt_products):
                    // glv = Hs z + G'lam, tv = Hs z + G'(d.rp); the objective takes z'Hs z = z'glv - (G z)'lam
                    gt_products<T>(Grm, ng, lam_, v1_, parts, glv, tv, tid, bq, zv);
                    double rdn = 0.0, obj = 0.0, dum2 = 0.0;
                    if (tid < NVP) {
                        const double qj = qv[tid], aj = glv[tid];
                        rdn = fabs(qj + aj);
                        obj = zv[tid] * (0.5 * aj + qj);
                        rhsv[tid] = -qj - tv[tid];
                    }
                    block_reduce3<SH::BW, OpMax, OpSum, OpMax>(rdn, obj, dum2, red, wave, lane);
                    obj -= 0.5 * gzl;
                    BSTAMP(2);
                    if (!(mu == mu) || !(rdn == rdn)) { st = TMPC_STATUS_NUMERICAL; break; }
                    const double objs = fmax(fabs(obj), 1.0);
                    const bool try_polish = (rdn <= 1e3 * try_tol * qn) && (rpn <= try_tol * hn) && (gap <= try_tol * objs);
                    if (try_polish) { want_polish = true; rdn_last = rdn; break; }
                    if (gap <= 1e-15 * objs) {
                        // (stalled gap: the refinement gets this iterate as it is, once -- tmpc_kernels.hip)
                        st = TMPC_STATUS_MAX_ITER;
                        if (!floor_tried) { floor_tried = true; want_polish = true; rdn_last = INFINITY; }
                        break;
                    }
                    if (lmax > 1e10) {
                        double hl = 0.0;
                        for (int r = tid; r < ncp; r += BT) hl += h_[r] * lam_[r];          // (padding rows: lambda = 0)
                        hl = block_reduce1<SH::BW, OpSum>(hl, red, wave, lane);
                        // |G'lam| on its own (rare path: a pass without the rows of Hs; tv is rebuilt by the next iteration's P2
                        // and not read before, rhsv is complete)
                        gt_products<T>(Grm, ng, lam_, v1_, parts, glv, tv, tid, bq);
                        double gln = (tid < NVP) ? fabs(glv[tid]) : 0.0;
                        gln = block_reduce1<SH::BW, OpMax>(gln, red, wave, lane);
                        if (hl < 0.0 && gln <= 1e-6 * lmax) { st = TMPC_STATUS_INFEASIBLE; break; }
                    }
                    // ---- P3 + P4: M = Hs + G'DG (MFMA), Cholesky, predictor solve
                    double shift = 0.0;
                    bool spd = false;
                    for (int attempt = 0; attempt < 2 && !spd; ++attempt) {
                        __syncthreads();           // (the previous readers of the LDS matrix are done)
                        BSTAMP(3);
                        gdg_all<T>(Grm, d_, bq, nsteps, big, wave, lane, qp.Hs, shift);
                        zblock_accumulate<T>(Grm, d_, bq, big, parts, tid);
                        BSTAMP(4);
#ifdef TMPC_STAMPS
                        spd = block_chol<SH::BW>(big, LDM, nv, dinv, red + 32, tid, tph, &tlast);
#else
                        spd = block_chol<SH::BW>(big, LDM, nv, dinv, red + 32, tid);
#endif
                        BSTAMP(5);
                        if (!spd) {
                            // 1e-13 * trace(M), as the oracle does: trace(G'DG) = sum of the weights (unit rows)
                            double trc = 0.0;
                            for (int i = 0; i < nv; ++i) trc += qp.Hs[i * NVP + i];
                            double dsum = 0.0;
                            for (int r = tid; r < ncp; r += BT) dsum += d_[r];                // (padding rows: d = 0)
                            dsum = block_reduce1<SH::BW, OpSum>(dsum, red, wave, lane);
                            shift = 1e-13 * (trc + dsum);
                        }
                    }
                    if (!spd) { st = TMPC_STATUS_NUMERICAL; break; }
#ifdef TMPC_STAMPS
                    block_invert<SH::BW>(big, LDM, nv, dinv, tid, tph, &tlast);
                    BSTAMP(14);
#else
                    block_invert<SH::BW>(big, LDM, nv, dinv, tid);
#endif
                    block_inv_solve<BT>(big, LDM, nv, dinv, rhsv, dzav, parts, tid, NVP);
                    BSTAMP(6);
                    // ---- P5: affine step statistics, corrector terms per row
                    double rho_aff = 0.0, sb1 = 0.0, sb2 = 0.0;
                    auto p5_row = [&](int r, double gd) {
                        const bool valid = valid_row(r);
                        const double sv = s_[r], lv = lam_[r], rp = rp_[r], d = d_[r];
                        const double rs = valid ? fast_rcp(sv) : 0.0;
                        const double dsa = valid ? (-rp - gd) : 0.0;
                        const double dla = valid ? (-lv - d * dsa) : 0.0;
                        const double rl = valid ? fast_rcp(lv) : 0.0;
                        rho_aff = fmax(rho_aff, fmax(-dsa * rs, -dla * rl));
                        const double w = dsa * dla;
                        sb1 += sv * dla + lv * dsa;
                        sb2 += w;
                        w_[r] = w;
                        c1_[r] = w * rs;
                        rs_[r] = rs;
                    };
                    row_products(dzav, p5_row);
                    block_reduce3<SH::BW, OpMax, OpSum, OpSum>(rho_aff, sb1, sb2, red, wave, lane);
                    const double aaff = rho_aff > 1.0 ? 1.0 / rho_aff : 1.0;
                    const double mu_aff = (gap + aaff * sb1 + aaff * aaff * sb2) / ncd;
                    double sigma = mu_aff / mu;
                    sigma = fmin(sigma * sigma * sigma, 1.0);
                    const double smu = sigma * mu;
                    BSTAMP(7);
                    // ---- P6: corrector right-hand side and solve
                    gt_products<T>(Grm, ng, c1_, rs_, parts, tv, uv, tid, bq);
                    if (tid < NVP) cgv[tid] = rhsv[tid] + tv[tid] - smu * uv[tid];
                    __syncthreads();
                    block_inv_solve<BT>(big, LDM, nv, dinv, cgv, dzv, parts, tid, NVP);
                    BSTAMP(8);
                    // ---- P7: final direction per row, step length
                    double om = (1.0 - aaff) * (1.0 - aaff);
                    om = fmin(fmax(om, 1e-4), 1e-2);
                    const double tau = 1.0 - om;
                    double rho = 0.0;
                    auto p7_row = [&](int r, double gd) {
                        const bool valid = valid_row(r);
                        const double sv = s_[r], lv = lam_[r], rp = rp_[r], rs = rs_[r];
                        const double dsk = valid ? (-rp - gd) : 0.0;
                        const double rc = sv * lv + w_[r] - smu;
                        const double dlk = valid ? (-(rc + lv * dsk) * rs) : 0.0;
                        const double rl = valid ? fast_rcp(lv) : 0.0;
                        rho = fmax(rho, fmax(-dsk * rs, -dlk * rl));
                        gdz_[r] = gd;
                        ds_[r] = dsk;
                        dl_[r] = dlk;
                    };
                    row_products(dzv, p7_row);
                    rho = block_reduce1<SH::BW, OpMax>(rho, red, wave, lane);
                    const double alpha = rho > tau ? tau / rho : 1.0;
                    // ---- P8: update
                    for (int r = tid; r < ncp; r += BT) {
                        s_[r] += alpha * ds_[r];
                        lam_[r] += alpha * dl_[r];
                        gz_[r] += alpha * gdz_[r];
                    }
                    if (tid < NVP) zv[tid] += alpha * dzv[tid];
                    __syncthreads();
                    it_done = it + 1;
                    BSTAMP(9);
                }
                if (!want_polish) break;
                // ------------------------------------------------ active-set refinement
                bool ok = false;
                {
                    double *S = big;
                    for (int r = tid; r < ncp; r += BT) {
                        const double lv = lam_[r];
                        inW_[r] = (valid_row(r) && lv > s_[r]) ? 1 : 0;
                        yall_[r] = lv;
                    }
                    if (tid < NVP) zpv[tid] = zv[tid];
                    int loose_retries = 0;        // rounds that only repeat the Newton steps on an unchanged working set (tmpc_kernels.hip)
                    for (int round = 0; round < 6 + loose_retries && !ok; ++round) {
                        __syncthreads();
                        if (wave == 0) {
                            int m0 = 0;
                            for (int r0 = 0; r0 < ncp; r0 += WAVE) {
                                const int r = r0 + lane;
                                const bool in = inW_[r] != 0;
                                const unsigned long long bal = __ballot(in);
                                const int pos = m0 + __popcll(bal & ((1ull << lane) - 1ull));
                                if (in && pos < WCAP) { Widx[pos] = r; yv[pos] = yall_[r]; }
                                m0 += __popcll(bal);
                            }
                            if (lane == 0) ibc[0] = m0;
                        }
                        __syncthreads();
                        const int m = ibc[0];
                        if (m > WCAP) break;
                        if (m == 0) {
                            const double v = column_sums<T>(qp.Hinv, qv, nullptr, nullptr, nullptr, 0, parts, tid);
                            if (tid < NVP) zpv[tid] = -v;
                            __syncthreads();
                        } else {
                            // S = G_W Hinv G_W' (lower triangle): rows of G Hinv (precomputed) . rows of G
                            for (int idx = tid; idx < m * m; idx += BT) {
                                const int a = idx / m, c2 = idx - a * m;
                                if (c2 > a) continue;
                                const double *ga = GHrm + static_cast<size_t>(Widx[a]) * NVP;
                                const double *gc = Gw + static_cast<size_t>(Widx[c2]) * NVP;
                                double v0 = 0.0, v1 = 0.0;
                                #pragma unroll 4
            for (int j = 0; j + 1 < NVP; j += 2) { v0 = fma(ga[j], gc[j], v0); v1 = fma(ga[j + 1], gc[j + 1], v1); }
                                S[a * LDSS + c2] = v0 + v1;
                            }
                            __syncthreads();
                            double dmax = (tid < m) ? S[tid * LDSS + tid] : 0.0;
                            dmax = block_reduce1<SH::BW, OpMax>(dmax, red, wave, lane);
                            if (tid < m) S[tid * LDSS + tid] += 1e-11 * dmax;
                            if (!block_chol<SH::BW>(S, LDSS, m, dinv, red + 32, tid)) break;
                            block_invert<SH::BW>(S, LDSS, m, dinv, tid);
                            double dz_prev = 0.0;
                            for (int stp = 0; stp < 12; ++stp) {      // (nearly parallel working rows need more than the usual two)
                                // r1 = Hs zp + q + G_W' y
                                {
                                    const double v = column_sums<T>(qp.Hs, zpv, Gw, Widx, yv, m, parts, tid);
                                    if (tid < NVP) tv[tid] = qv[tid] + v;
                                }
                                __syncthreads();
                                // t1 = Hinv r1
                                {
                                    const double v = column_sums<T>(qp.Hinv, tv, nullptr, nullptr, nullptr, 0, parts, tid);
                                    if (tid < NVP) uv[tid] = v;
                                }
                                __syncthreads();
                                // dy rhs: (G_W zp - h_W) - G_W t1
                                if (tid < m) {
                                    const int r = Widx[tid];
                                    const double *g = Gw + static_cast<size_t>(r) * NVP;
                                    double gz = 0.0, gt = 0.0;
#pragma unroll 8
                                    for (int j = 0; j < NVP; ++j) { gz = fma(g[j], zpv[j], gz); gt = fma(g[j], uv[j], gt); }
                                    dyv[tid] = gz - h_[r] - gt;
                                }
                                __syncthreads();
                                block_inv_solve<BT>(S, LDSS, m, dinv, dyv, dyv, parts, tid);
                                // zp -= t1 + Hinv G_W' dy ; y += dy
                                double dzl = 0.0, zl = 1.0;
                                const double ghd = column_sums<T>(nullptr, nullptr, GHrm, Widx, dyv, m, parts, tid);
                                if (tid < NVP) {
                                    const double v = uv[tid] + ghd;
                                    const double zn2 = zpv[tid] - v;
                                    zpv[tid] = zn2;
                                    dzl = fabs(v);
                                    zl = fmax(fabs(zn2), 1.0);
                                }
                                if (tid < m) yv[tid] += dyv[tid];
                                double dum = 0.0;
                                block_reduce3<SH::BW, OpMax, OpMax, OpMax>(dzl, zl, dum, red, wave, lane);
                                if (stp >= 1) {       // the step no longer moves the iterate, or what is left after it cannot (tmpc_kernels.hip)
                                    const double rho = dzl / fmax(dz_prev, 1e-300);
                                    if (dzl <= 1e-14 * zl || (loose_retries == 0 && rho < 0.5 && dzl * rho <= 0.5e-15 * zl)) break;
                                }
                                dz_prev = dzl;
                            }
                        }
                        // ---- verify: primal feasibility on all rows, sign of y on W
                        double ymax = (tid < m) ? fabs(yv[tid]) : 1.0;
                        ymax = fmax(ymax, 1.0);
                        ymax = block_reduce1<SH::BW, OpMax>(ymax, red, wave, lane);
                        if (tid < m) yall_[Widx[tid]] = yv[tid];
                        __syncthreads();
                        double nviol = 0.0, nneg = 0.0, nloose = 0.0;
                        row_products(zpv, [&](int r, double gzr) {
                          {
                            const bool valid = valid_row(r);
                            const double hk = h_[r];
                            const double rr = gzr - hk;
                            rr_[r] = rr;
                            const bool in = inW_[r] != 0;
                            const double hi = fmax(fabs(hk), 1.0);
                            const bool viol = valid && !in && rr > 1e-12 * hi;
                            const bool loose = in && fabs(rr) > 1e-11 * hi;
                            const bool neg = in && yall_[r] < -1e-10 * ymax;
                            nviol += viol ? 1.0 : 0.0;
                            nneg += neg ? 1.0 : 0.0;
                            nloose += loose ? 1.0 : 0.0;
                            if (neg) { inW_[r] = 0; yall_[r] = 0.0; }
                            if (viol) { inW_[r] = 1; yall_[r] = 0.0; }
                          }
                        });
                        block_reduce3<SH::BW, OpSum, OpSum, OpSum>(nviol, nneg, nloose, red, wave, lane);
                        // rows of W off their bound with nothing left to correct: not converged, give up (see tmpc_kernels.hip)
                        if (nloose != 0.0 && nviol == 0.0 && nneg == 0.0) {
                            if (loose_retries >= 2) break;
                            ++loose_retries;          // the set is right, its nearly parallel rows need more steps
                            continue;
                        }
                        if (nviol == 0.0 && nneg == 0.0) {
                            ok = true;
                            if (tid < NVP) zv[tid] = zpv[tid];
                            for (int r = tid; r < ncp; r += BT) {
                                const double rr = rr_[r];
                                lam_[r] = inW_[r] ? fmax(yall_[r], 0.0) : 0.0;
                                s_[r] = rr < 0.0 ? -rr : 0.0;
                                gz_[r] = rr + h_[r];
                            }
                            __syncthreads();
                        }
                    }
                }
                BSTAMP(10);
                if (ok) { st = TMPC_STATUS_OPTIMAL; break; }
                // (the product is formed here, from an opaque copy: hoisted to the top of the instance it was a register that lived -- and at
                // T = 8 was spilled -- through the whole solve for the sake of this rare exit)
                if (try_tol <= 1e-12) { st = (rdn_last <= 1e-9 * fresh_value(qn)) ? TMPC_STATUS_OPTIMAL : TMPC_STATUS_MAX_ITER; break; }
                try_tol *= 1e-2;
            }
            // (iteration cap: the last iterate goes out under TMPC_STATUS_MAX_ITER, see tmpc_kernels.hip; INFEASIBLE needs the
            // Farkas-type certificate)
        }

        // ---------------------------------------------------------------- outputs
        const bool good = st < TMPC_STATUS_INFEASIBLE;
        const double nanv = __longlong_as_double(0x7ff8000000000000ll);
        __syncthreads();
        // z_full = [u | theta | x_0 ..]: Dv .* z, or Tzs z + Txf x_k when a further equality was eliminated at set-up
        // (terminal equality of the tracking MPC; tmpc_condense.hpp)
        const double *zo = tv;
        if (qp.Tzs != nullptr) {
            for (int i = tid; i < qp.nvf; i += BT) {          // nvf <= nv + nx <= 2 BT entries of `parts` (idle here)
                double v = 0.0;
                for (int c2 = 0; c2 < nx; ++c2) v += qp.Txf[i * nx + c2] * xin[c2];
                for (int j = 0; j < nv; ++j) v += qp.Tzs[i * nv + j] * zv[j];
                parts[i] = v;
            }
            zo = parts;
        } else if (tid < NVP) {
            tv[tid] = (tid < nv) ? qp.Dv[tid] * zv[tid] : 0.0;     // unscaled z
        }
        __syncthreads();
        for (int i = tid; i < N * nu; i += BT) lp->u_nom[b * N * nu + i] = good ? zo[i] : nanv;
        if (tid < nx + nu && lp->xu_ss) {
            double v = 0.0;
            for (int j = 0; j < qp.nth; ++j) v += qp.Mth[tid * qp.nth + j] * zo[qp.off_theta + j];
            lp->xu_ss[b * (nx + nu) + tid] = good ? v : nanv;
        }
        if (tid < nx) {
            const double x0 = (qp.off_x0 >= 0) ? zo[qp.off_x0 + tid] : xin[tid];
            if (lp->x_nom0) lp->x_nom0[b * nx + tid] = good ? x0 : nanv;
            uv[tid] = x0;
        }
        if (double *const x_nom = lp->x_nom) {
            __syncthreads();
            if (tid < nx) x_nom[b * (N + 1) * nx + tid] = good ? uv[tid] : nanv;
            for (int i = 0; i < N; ++i) {
                double v = 0.0;
                if (tid < nx) {
                    for (int j = 0; j < nx; ++j) v += qp.A[tid * nx + j] * uv[j];
                    for (int j = 0; j < nu; ++j) v += qp.B[tid * nu + j] * zo[i * nu + j];
                }
                __syncthreads();
                if (tid < nx) { uv[tid] = v; x_nom[b * (N + 1) * nx + (i + 1) * nx + tid] = good ? v : nanv; }
                __syncthreads();
            }
        }
        if (tid == 0) { lp->status[b] = st; lp->iters[b] = it_done; }
        if (lp->ticks && tid == 0) lp->ticks[b] = static_cast<long long>(__builtin_amdgcn_s_memrealtime()) - t_begin;
#ifdef TMPC_STAMPS
        BSTAMP(11);
        if (blockIdx.x == 0 && tid == 0 && lp->dbg && it_done > 0) { for (int p_ = 0; p_ < 16; ++p_) lp->dbg[p_] = tph[p_]; }
#endif
    }
}
#undef qp
#undef bq
#undef WSP
#undef s_
#undef lam_
#undef h_
#undef gz_
#undef rp_
#undef d_
#undef v1_
#undef w_
#undef c1_
#undef rs_
#undef ds_
#undef dl_
#undef gdz_
#undef yall_
#undef rr_
#undef inW_

#ifdef TMPC_HOST_SIM
unsigned long sim_rendezvous_total = 0;
// tests/wavesim: one workgroup on the host execution model takes the whole batch (grid of one)
template <int T>
hipError_t launch_block_t(const DeviceQP &qp, const BlockQP &bq, const BlockArgs *dargs, double *ws, int ws_blocks, int variant_id, int64_t B,
                          const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                          double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, WorkCounter *wc, hipStream_t stream) {
    constexpr size_t lds = sizeof(double) * BShape<T>::TOTAL;
    static_assert(lds <= 160 * 1024, "block shape does not fit the 160 KiB LDS of a CU");
    (void)ws_blocks; (void)stream; (void)dargs; (void)wc;
    unsigned long long counter = 0, *next_item = &counter;
    const BlockArgs host_args{qp, bq};
    sim::Dim3 bi, gd;
    bi.x = bi.y = bi.z = 0;
    sim_rendezvous_total += sim::run_block(BShape<T>::BT, lds, bi, gd, [&]() {
        solve_block_kernel<T>(BlockLaunch{&host_args, qp.ticks, qp.dbg, ws, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, next_item});
    });
    return hipSuccess;
}
template <int T>
int block_occupancy_t() { return 1; }
#else
template <int T>
hipError_t launch_block_t(const DeviceQP &qp, const BlockQP &bq, const BlockArgs *dargs, double *ws, int ws_blocks, int variant_id, int64_t B,
                          const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                          double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, WorkCounter *wc, hipStream_t stream) {
    constexpr size_t lds = sizeof(double) * BShape<T>::TOTAL;
    static_assert(lds <= 160 * 1024, "block shape does not fit the 160 KiB LDS of a CU");
    static std::atomic<bool> attr_set[64] = {};       // (the size is a compile-time constant here: setting it twice is harmless)
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_block_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id].store(true, std::memory_order_release);
    }
    if (dargs == nullptr || wc == nullptr || wc->ring == nullptr) return hipErrorInvalidValue;
    (void)bq;
    // a fresh (zero) word of the counter ring per launch; the ring is cleared in one piece when it has gone round (tmpc_device.hpp)
    if (wc->pos >= wc->size) {
        hipError_t e0 = hipMemsetAsync(wc->ring, 0, sizeof(unsigned long long) * wc->size, stream);
        if (e0 != hipSuccess) return e0;
        wc->pos = 0;
    }
    unsigned long long *const next_item = wc->ring + wc->pos;
    ++wc->pos;
    int64_t blocks = B < ws_blocks ? B : ws_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((solve_block_kernel<T>), dim3(static_cast<unsigned>(blocks)), dim3(BShape<T>::BT), lds, stream,
                       BlockLaunch{dargs, qp.ticks, qp.dbg, ws, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, next_item});
    return hipGetLastError();
}

template <int T>
int block_occupancy_t() {
    constexpr size_t lds = sizeof(double) * BShape<T>::TOTAL;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_block_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, solve_block_kernel<T>, BShape<T>::BT, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

#endif

}  // namespace

int block_tiles(int nv) {
    if (nv <= 16) return 1;
    if (nv <= 32) return 2;
    if (nv <= 64) return 4;
    if (nv <= 128) return 8;
    return 0;
}

int block_workspace_rows() { return WS_COUNT; }
#ifdef TMPC_HOST_SIM
unsigned long sim_rendezvous_count() { return sim_rendezvous_total; }
#endif

size_t block_lds_bytes(int tiles) {
    switch (tiles) {
        case 1: return sizeof(double) * BShape<1>::TOTAL;
        case 2: return sizeof(double) * BShape<2>::TOTAL;
        case 4: return sizeof(double) * BShape<4>::TOTAL;
        case 8: return sizeof(double) * BShape<8>::TOTAL;
    }
    return 0;
}

int block_occupancy(int tiles) {
    switch (tiles) {
        case 1: return block_occupancy_t<1>();
        case 2: return block_occupancy_t<2>();
        case 4: return block_occupancy_t<4>();
        case 8: return block_occupancy_t<8>();
    }
    return 1;
}

hipError_t launch_block(const DeviceQP &qp, const BlockQP &bq, const BlockArgs *dargs, int tiles, double *ws, int ws_blocks, int variant_id, int64_t B,
                        const double *x_k, const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0,
                        double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, WorkCounter *wc, hipStream_t stream) {
    switch (tiles) {
        case 1: return launch_block_t<1>(qp, bq, dargs, ws, ws_blocks, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, wc, stream);
        case 2: return launch_block_t<2>(qp, bq, dargs, ws, ws_blocks, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, wc, stream);
        case 4: return launch_block_t<4>(qp, bq, dargs, ws, ws_blocks, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, wc, stream);
        case 8: return launch_block_t<8>(qp, bq, dargs, ws, ws_blocks, variant_id, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters, wc, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace tmpc
