// Batched low-dimensional linear programs for the OFFLINE stage (gfx950):
//
//     val_b = max c_b' x   s.t.  H x <= h  (+ relax_by on row relax_b)
//
// one shared polytope (H, h) with nr rows in d <= 32 dimensions, one objective per instance.  These are the support-function
// LPs behind the set computations that produce the (H, h) blocks of the MPC problem: the reference evaluates them one at a
// time with scipy.optimize.linprog (reference utils_polytope.py:12-23 `support`, :19 the linprog call), several thousand per
// model -- the Gilbert-Tan recursion (utils_polytope.py:247-268), the redundancy removal behind `pc.reduce`
// (TubeRegulatorMPC.py:74) and the Pontryagin differences (utils_polytope.py:25-38).
//
// One wavefront solves one LP.  Rows are strided over the 64 lanes; the per-row state (s, lam, G x, the corrector term and
// the step) lives in a per-wave global workspace that stays in L2, H is stored transposed so that a row sweep is coalesced.
//
//   1. Mehrotra predictor-corrector on the normal equations M = H' diag(lam / s) H (d x d, rows of M on the lanes,
//      LDL' by readlane elimination), infeasible start -- same iteration as the QP kernels without the Hessian.  Pivots
//      that have become round-off are skipped (lp_factor): on an optimal face of dimension >= 1 M loses its rank.
//   2. Hand-over at tol: the rows with lam > s are the IPM's guess of the optimal face.  Non-negative least squares over
//      them (y >= 0 minimising |c - H' y|) picks the working rows -- the ones with y > 0 -- and gives a certificate that
//      needs no vertex: the iterate is strictly feasible, y is dual feasible, so the value lies within sum y_i s_i +
//      |c - H' y| |x| of c . x (accepted below 1e-11).  Primal active-set steps then make the answer exact: project onto the face,
//      multipliers by least squares; a violated row replaces the working row it is (nearly) parallel to, a negative
//      multiplier leaves, a remaining component of c along the face is followed to the blocking row.  Accept when the
//      point is feasible, the multipliers are non-negative and c is in the cone of the working rows.
//   3. Otherwise the IPM continues with a 100 times tighter tolerance and hands over again (1e-8, 1e-10, 1e-12); at
//      1e-12 its own iterate is accepted if the dual residual is below 1e-11 as well.
//   4. Otherwise -- c keeps a component along the optimal face that the skipped pivots cannot remove -- the IPM runs on
//      until the gap has collapsed (the iterate then lies ON its active rows) and feasible ascent directions take over:
//      non-negative least squares over the active rows gives y >= 0 and p = c - H_A' y with a_j . p <= 0 on all of them,
//      x moves along p to the first row in the way.  No cycling at degenerate vertices, multipliers non-negative by
//      construction.  It ends without certificate (status MAX_ITER, the IPM iterate is returned) when |p| ~ 1e-8 has to be
//      followed over a distance ~1: the round-off of c - H_A' y (1e-16) then decides on which side of an active row x lands.
#ifdef TMPC_HOST_SIM
#include "hip_sim.hpp"      // tests/wavesim: this very source compiled for the CPU under sanitizers (never in the product)
#else
#include <hip/hip_runtime.h>
#endif

#include <cmath>
#include <cstdint>
#include <utility>

#include "tmpc_device.hpp"
#include "tmpc_wave.hpp"

namespace tmpc {

namespace {

using namespace wv;

constexpr int LP_WPB = 4;        // waves (= LPs in flight) per workgroup
constexpr int LP_ARR = 6;        // per-row workspace arrays: s, lam, Hx, w, ds, dlam
// dual feasibility of an accepted answer, for |c| = 1: multipliers >= -DUAL_TOL_Y max(y), |c - H_W' y|_inf <= DUAL_TOL_P
// (HiGHS behind the reference's linprog calls: 1e-7).  What c keeps along the face is paid for with the length of the
// face: on the terminal sets of the synthetic model (d = 28) a residual of 8e-10 left 2e-8 of the value behind, so
// 1e-10 keeps the value to a few 1e-9.  Primal feasibility: 1e-11 max(|h_r|, 1).
constexpr double DUAL_TOL_Y = 1e-10, DUAL_TOL_P = 1e-10;
// duality gap, relative to max(|val|, 1) in kernel units, of an answer that is accepted without being a point of the face
constexpr double GAP_TOL = 1e-11;

template <int D> __host__ __device__ constexpr int lp_col_off(int j) { return j * D - j * (j - 1) / 2; }

template <int D>
struct LpLds {
    static constexpr int NT = D * (D + 1) / 2;
    static constexpr int RED = 16 * RED_STRIDE;
    static constexpr int SUMS = NT + 2 * D + 8;
    static constexpr int VEC = 8 * D;             // c, x, dx_aff, dx, xp, p, t, y
    static constexpr int GW = D * D;              // working rows
    static constexpr int IDX = 64 + D + 8;        // candidate ids, working ids (ints, stored in double slots)
    static constexpr int TOTAL = RED + SUMS + VEC + GW + IDX;
};

// column blocks of the lower triangle of M accumulated per row sweep (<= ~160 accumulators each)
template <int D> struct LpBlocks { static constexpr int n = 1; static constexpr int b[2] = {0, D}; };
template <> struct LpBlocks<32> { static constexpr int n = 5; static constexpr int b[6] = {0, 3, 7, 12, 18, 32}; };

__device__ __forceinline__ double lp_hrow(const LpDevice &p, int rel, int r) {
    return p.h[r] + (r == rel ? p.relax_by * p.rscale[r] : 0.0);
}

// LDL' of the normal matrix (rows on the lanes, as wv::rows_factor) that SKIPS a pivot lost to cancellation: once the pivot
// has fallen below piv_tol times the diagonal entry it started from, what is left of it is round-off of the eliminations.
// On an optimal face of dimension >= 1 (c in the span of fewer than d rows: the usual case for the support functions of
// a polytope with nearly parallel rows) M = H' diag(lam / s) H loses its rank as the gap closes; dividing by such a pivot
// sends the step along the face by O(1), and the dual residual picks up eps |M| |dx| per iteration and never comes back.
// A skipped pivot fixes dx_k = 0 (row and column k leave the system): the iterate stops drifting along the face, whose
// points are all optimal.
template <int N>
__device__ __forceinline__ bool lp_factor(double (&row)[N], double &b, double &dinv, double piv_tol, int lane) {
    bool ok = true;
    double diag0 = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) diag0 = (lane == k) ? row[k] : diag0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double pkk = readlane_d(row[k], k), d0 = readlane_d(diag0, k);
        ok = ok && (pkk == pkk);
        const bool skip = !(pkk > piv_tol * d0);
        const double pinv = skip ? 0.0 : fast_rcp(pkk);
        const double f = (lane > k) ? row[k] * pinv : 0.0;
#pragma unroll
        for (int j = k + 1; j < N; ++j) row[j] = fma(-f, readlane_d(row[j], k), row[j]);
        b = fma(-f, readlane_d(b, k), b);
        if (lane > k) row[k] = f;
        if (lane == k) dinv = pinv;
    }
    return ok;
}

// Columns [J0, J1) of M = H' diag(lam / s) H over this lane's rows, summed over the wave into `sums` (packed lower
// triangle, column-major).  The FIRST sweep also forms H'(d . r_p), H'lam, the gap, |r_p|_inf and max lam.
template <int D, int J0, int J1, bool FIRST>
__device__ __forceinline__ void sweep_a(const LpDevice &p, const double *__restrict__ Ht, int rel, const double *s_, const double *lam_,
                                        const double *gz_, double *red, double *sums, double &rpn, double &lmax, int lane) {
    constexpr int NT = D * (D + 1) / 2;
    constexpr int C0 = lp_col_off<D>(J0), CM = lp_col_off<D>(J1) - C0;
    constexpr int CNT = CM + (FIRST ? 2 * D + 1 : 0);
    const int nr = p.nr, nrp = p.nrp;
    double acc[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) acc[k] = 0.0;
    for (int r = lane; r < nrp; r += WAVE) {
        const bool valid = r < nr;
        const double sv = s_[r], lv = lam_[r];
        const double dd = valid ? lv * fast_rcp(sv) : 0.0;
        double g[D];
#pragma unroll
        for (int j = J0; j < D; ++j) g[j] = Ht[static_cast<size_t>(j) * nrp + r];
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const double dg = dd * g[j];
#pragma unroll
            for (int i = j; i < D; ++i) acc[lp_col_off<D>(j) - C0 + i - j] = fma(dg, g[i], acc[lp_col_off<D>(j) - C0 + i - j]);
        }
        if constexpr (FIRST) {
            const double rp = valid ? gz_[r] + sv - lp_hrow(p, rel, r) : 0.0;
            const double drp = dd * rp;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                acc[CM + j] = fma(drp, g[j], acc[CM + j]);
                acc[CM + D + j] = fma(lv, g[j], acc[CM + D + j]);
            }
            acc[CM + 2 * D] = fma(sv, lv, acc[CM + 2 * D]);
            rpn = fmax(rpn, fabs(rp));
            lmax = fmax(lmax, lv);
        }
    }
    if constexpr (FIRST) {
        static_assert(J0 == 0, "the first block starts at column 0");
        // the block's columns, then the 2 D + 1 extra totals behind the whole triangle
        double accm[CM], acce[2 * D + 1];
#pragma unroll
        for (int k = 0; k < CM; ++k) accm[k] = acc[k];
#pragma unroll
        for (int k = 0; k < 2 * D + 1; ++k) acce[k] = acc[CM + k];
        reduce_to_lds<CM>(accm, red, sums, lane);
        reduce_to_lds<2 * D + 1>(acce, red, sums + NT, lane);
    } else {
        reduce_to_lds<CNT>(acc, red, sums + C0, lane);
    }
}

// The working rows widx[0 .. m) of a face: lane a < m loads row a (gw), all of them go to GW [m][D] in LDS, and
// S = GW GW' (m x m, identity beyond) is factored with its rows on the lanes.  False: the rows are dependent.
template <int D>
__device__ __forceinline__ bool lp_face_factor(const double *__restrict__ Ht, int nrp, const int *widx, int m, double *GW, double (&gw)[D],
                                               double (&srow)[D], double &sdinv, int lane) {
    const bool inw = lane < m;
    const int wr = inw ? widx[lane] : 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        gw[k] = inw ? Ht[static_cast<size_t>(k) * nrp + wr] : 0.0;
        if (lane < D) GW[lane * D + k] = gw[k];
    }
    lds_fence();
#pragma unroll
    for (int c2 = 0; c2 < D; ++c2) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(gw[k], GW[c2 * D + k], v);
        srow[c2] = (inw && c2 < m) ? v : ((c2 == lane) ? 1.0 : 0.0);
    }
    double bdummy = 0.0;
    sdinv = 1.0;
    return rows_factor<D>(srow, bdummy, sdinv, lane);
}

// One outer step of non-negative least squares (Lawson, Hanson) for  min |c - H_W' y|, y >= 0:  row j joins the passive
// set W (widx[0 .. m), multipliers yv), the least-squares multipliers z of the new set are computed (tv; refined twice), and
// y moves towards z -- all the way if z > 0, otherwise to where the first multiplier reaches zero; that row leaves and the
// step is repeated on the smaller set.  y >= 0 throughout, |c - H_W' y| does not rise.  False (W, y as before): j depends on
// W to working precision, or comes out again at once (a_j . p > 0 was round-off).  pv is scratch.
template <int D>
__device__ __forceinline__ bool lp_nnls_add(const double *__restrict__ Ht, int nrp, int d, int j, const double *cv, int *widx, int &m,
                                            double *GW, double *yv, double *tv, double *pv, int lane) {
    if (lane == 0) { widx[m] = j; yv[m] = 0.0; }
    ++m;
    lds_fence();
    for (int inner = 0; inner <= d; ++inner) {
        const bool inw = lane < m;
        double gw[D], srow[D], sdinv = 1.0;
        if (!lp_face_factor<D>(Ht, nrp, widx, m, GW, gw, srow, sdinv, lane)) {
            if (inner == 0) { --m; lds_fence(); return false; }
            break;                                                  // (a subset of a set that factored: not expected)
        }
        if (lane < D) tv[lane] = 0.0;
        lds_fence();
        for (int sweep = 0; sweep < 3; ++sweep) {
            if (lane < D) {
                double v = cv[lane];
                for (int a = 0; a < m; ++a) v = fma(-GW[a * D + lane], tv[a], v);
                pv[lane] = v;
            }
            lds_fence();
            double b2 = 0.0;
            if (inw) {
#pragma unroll
                for (int k = 0; k < D; ++k) b2 = fma(gw[k], pv[k], b2);
            }
            rows_forward<D>(srow, b2, lane);
            const double zs = rows_backsub_lane<D>(srow, b2, sdinv, lane);
            if (inw) tv[lane] += zs;
            lds_fence();
        }
        const double zl = inw ? tv[lane] : INFINITY, yl = inw ? yv[lane] : 0.0;
        const double zmin = wave_reduce<OpMin>(zl);
        if (zmin > 0.0) {
            if (inw) yv[lane] = zl;
            lds_fence();
            break;
        }
        const double al = (inw && zl <= 0.0) ? yl / (yl - zl) : INFINITY;
        const double alpha = wave_reduce<OpMin>(al);
        const unsigned long long bl = __ballot(inw && al == alpha);
        const int i = __ffsll(static_cast<long long>(bl)) - 1;
        if (i == m - 1 && inner == 0) { --m; lds_fence(); return false; }
        const double yn = inw ? fmax(yl + alpha * (zl - yl), 0.0) : 0.0;
        const int wmove = inw ? widx[lane] : 0;
        lds_fence();
        if (inw && lane < i) yv[lane] = yn;
        if (inw && lane > i) { yv[lane - 1] = yn; widx[lane - 1] = wmove; }
        --m;
        lds_fence();
    }
    return true;
}

// p = c - H_W' y, projected onto the null space of the passive rows twice more (gw, srow, sdinv: lp_face_factor of W):
// the subtraction leaves round-off of size eps |c| along the rows of W, as large as p itself near the end.  |p|_inf.
template <int D>
__device__ __forceinline__ double lp_face_residual(int m, const double *cv, const double *GW, const double *yv, double *tv, double *pv,
                                                   const double (&gw)[D], const double (&srow)[D], double sdinv, int lane) {
    const bool inw = lane < m;
    double pl_ = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        if (pass > 0) {
            double b2 = 0.0;
            if (inw) {
#pragma unroll
                for (int k = 0; k < D; ++k) b2 = fma(gw[k], pv[k], b2);
            }
            rows_forward<D>(srow, b2, lane);
            const double dl = rows_backsub_lane<D>(srow, b2, sdinv, lane);
            if (lane < D) tv[lane] = inw ? dl : 0.0;
            lds_fence();
        }
        if (lane < D) {
            double v = pass == 0 ? cv[lane] : pv[lane];
            const double *coef = pass == 0 ? yv : tv;
            for (int a = 0; a < m; ++a) v = fma(-GW[a * D + lane], coef[a], v);
            pv[lane] = v;
            pl_ = fabs(v);
        }
        lds_fence();
        if (m == 0) break;
    }
    return wave_reduce<OpMax>(pl_);
}

// STAGED: H' (D x nrp doubles) is copied to LDS once per workgroup and every row sweep reads it from there; otherwise
// the sweeps read it from global memory (L2)
template <int D, bool STAGED>
__global__ __launch_bounds__(WAVE *LP_WPB, 1) void lp_kernel(LpDevice p, int64_t B, const double *__restrict__ C,
                                                              const int32_t *__restrict__ relax, double *__restrict__ ws,
                                                              double *__restrict__ val, double *__restrict__ xout,
                                                              int32_t *__restrict__ status, int32_t *__restrict__ iters) {
    using L = LpLds<D>;
    constexpr int NT = L::NT;
#ifdef TMPC_HOST_SIM
    double *smem = sim::lds<double>();
#else
    extern __shared__ __attribute__((aligned(16))) double smem[];
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *red = smem + wave * L::TOTAL;
    double *sums = red + L::RED;
    double *cv = sums + L::SUMS;                 // objective, unit norm
    double *xv = cv + D, *dxav = xv + D, *dxv = dxav + D, *xpv = dxv + D, *pv = xpv + D, *tv = pv + D, *yv = tv + D;
    double *GW = cv + L::VEC;                    // [D][D] working rows
    int *cidx = reinterpret_cast<int *>(GW + L::GW);   // [64] candidate rows
    int *widx = cidx + 64;                       // [D] working rows

    const int d = p.d, nr = p.nr, nrp = p.nrp;
    const double *__restrict__ Ht = p.Ht;
    if constexpr (STAGED) {
        double *hl = smem + LP_WPB * L::TOTAL;
        for (int i = threadIdx.x; i < D * nrp; i += blockDim.x) hl[i] = p.Ht[i];
        __syncthreads();
        Ht = hl;
    }
    const int slot = blockIdx.x * LP_WPB + wave, nslots = gridDim.x * LP_WPB;
    double *s_ = ws + static_cast<size_t>(slot) * LP_ARR * nrp;
    double *lam_ = s_ + nrp, *gz_ = lam_ + nrp, *w_ = gz_ + nrp, *ds_ = w_ + nrp, *dl_ = ds_ + nrp;

    // Work distribution: the first LP of a wave is its slot, the later ones are drawn from the launch's counter (LPs of one batch
    // differ by a factor of three in their iteration counts, and the hard ones -- hand-overs that fail, ascent steps -- by much more)
    bool first_item = true;
    for (;;) {
        int64_t b;
        if (first_item) {
            b = slot;
            first_item = false;
        } else {
            unsigned long long drawn = 0;
            if (lane == 0) drawn = atomicAdd(p.next_item, 1ull);
            b = nslots + static_cast<int64_t>((static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(drawn >> 32)))) << 32) |
                                              static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(drawn & 0xffffffffull))));
        }
        if (b >= B) break;
        // ---- objective
        const double cl = lane < d ? C[b * d + lane] : 0.0;
        const double cn = sqrt(wave_reduce<OpSum>(cl * cl));
        const int rel = relax ? relax[b] : -1;
        if (!(cn > 0.0) || !(cn < INFINITY)) {
            if (lane == 0) { val[b] = cn == 0.0 ? 0.0 : NAN; status[b] = cn == 0.0 ? TMPC_STATUS_OPTIMAL : TMPC_STATUS_NUMERICAL; iters[b] = 0; }
            if (xout && lane < d) xout[b * d + lane] = 0.0;
            continue;
        }
        if (lane < D) { cv[lane] = cl / cn; xv[lane] = 0.0; }
        lds_fence();
        auto hrow = [&](int r) { return p.h[r] + (r == rel ? p.relax_by * p.rscale[r] : 0.0); };
        // ---- starting point: x = 0, slack floored, unit multipliers
        double smin = INFINITY, hn = 1.0;
        for (int r = lane; r < nr; r += WAVE) { const double hr = hrow(r); smin = fmin(smin, hr); hn = fmax(hn, fabs(hr)); }
        smin = wave_reduce<OpMin>(smin);
        hn = wave_reduce<OpMax>(hn);
        {
            const double fl = 0.1 * fmax(-smin, 1.0);
            for (int r = lane; r < nrp; r += WAVE) {
                const bool valid = r < nr;
                s_[r] = valid ? fmax(hrow(r), fl) : 1.0;
                lam_[r] = valid ? 1.0 : 0.0;
                gz_[r] = 0.0;
            }
        }
        const double ncd = static_cast<double>(nr);
        int st = TMPC_STATUS_MAX_ITER, it = 0;
        double try_tol = p.tol, rdn_last = 0.0;
        double value = NAN;
        bool from_polish = false, collapse = false;

        for (;;) {
            bool want_polish = false, stalled = false;
            for (; it < p.max_iter; ++it) {
                // ---- pass A: residuals, M = H' D H, H'(d.rp), H'lam (for d > 16 the lower triangle of M is accumulated
                // in column blocks, one sweep over the rows each, to stay inside the register file)
                double rpn = 0.0, lmax = 0.0;
                [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                    (sweep_a<D, LpBlocks<D>::b[Ks], LpBlocks<D>::b[Ks + 1], Ks == 0>(p, Ht, rel, s_, lam_, gz_, red, sums, rpn, lmax, lane), ...);
                }(std::make_integer_sequence<int, LpBlocks<D>::n>{});
                rpn = wave_reduce<OpMax>(rpn);
                lmax = wave_reduce<OpMax>(lmax);
                const double gap = sums[NT + 2 * D], mu = gap / ncd;
                double rdl = 0.0, objl = 0.0, xal = 0.0;
                if (lane < D) {
                    rdl = fabs(sums[NT + D + lane] - cv[lane]);
                    objl = cv[lane] * xv[lane];
                    xal = fabs(xv[lane]);
                }
                const double rdn = wave_reduce<OpMax>(rdl), obj = wave_reduce<OpSum>(objl), xn = wave_reduce<OpMax>(xal);
                rdn_last = rdn;
                if (!(mu == mu) || !(rdn == rdn)) { st = TMPC_STATUS_NUMERICAL; break; }
                const double objs = fmax(fabs(obj), 1.0);
#ifdef TMPC_LP_DEBUG
                if (lane == 0) printf("lp %lld it %d rdn %.3e rpn %.3e gap %.3e obj %.12e tol %.1e lmax %.2e\n", (long long)b, it, rdn, rpn, gap, obj, try_tol, lmax);
#endif
                if (!collapse && rdn <= 1e3 * try_tol && rpn <= try_tol * hn && gap <= try_tol * objs) { want_polish = true; break; }
                // the gap has collapsed and the dual residual has not followed: what is left of it lies along directions of the
                // optimal face that the factorisation cannot resolve any more (lp_factor); only vertex steps can remove it
                if (rpn <= try_tol * hn && gap <= 1e-6 * try_tol * objs) { want_polish = stalled = true; break; }
                if (xn > 1e9) { st = TMPC_STATUS_UNBOUNDED; break; }
                if (lmax > 1e10) {
                    double hl = 0.0;
                    for (int r = lane; r < nr; r += WAVE) hl += hrow(r) * lam_[r];
                    hl = wave_reduce<OpSum>(hl);
                    if (hl < 0.0) { st = TMPC_STATUS_INFEASIBLE; break; }
                }
                // ---- factor M, predictor.  No regularisation: near the solution the weak directions of M carry the dual
                // residual, a shift large enough to matter for the pivots would freeze them; pivots that are round-off are
                // skipped instead (lp_factor)
                double mrow[D], mdinv = 1.0;
                const double rhs_i = lane < D ? cv[lane] - sums[NT + lane] : 0.0;
                double bb = rhs_i;
                bool spd = false;
                {
#ifdef TMPC_LP_DEBUG
                    static const double piv_tol = getenv("LP_PIV") ? atof(getenv("LP_PIV")) : 1e-12;
#else
                    constexpr double piv_tol = 1e-12;
#endif
                    const int i = lane < D ? lane : 0;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const int lo = i < j ? i : j, hi = i < j ? j : i;
                        double v = sums[lp_col_off<D>(lo) + hi - lo];
                        if (i == j && i >= d) v = 1.0;
                        mrow[j] = (lane < D) ? v : 0.0;
                    }
                    spd = lp_factor<D>(mrow, bb, mdinv, piv_tol, lane);
                }
                if (!spd) { st = TMPC_STATUS_NUMERICAL; break; }
                {
                    const double xl = rows_backsub_lane<D>(mrow, bb, mdinv, lane);
                    if (lane < D) dxav[lane] = xl;
                }
                lds_fence();
                // ---- pass B: affine step, corrector terms
                double accb[2 * D + 2];
#pragma unroll
                for (int k = 0; k < 2 * D + 2; ++k) accb[k] = 0.0;
                double rho_aff = 0.0;
                for (int r = lane; r < nrp; r += WAVE) {
                    const bool valid = r < nr;
                    const double sv = s_[r], lv = lam_[r];
                    double g[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) g[j] = Ht[static_cast<size_t>(j) * nrp + r];
                    double gd = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) gd = fma(g[j], dxav[j], gd);
                    const double rp = valid ? gz_[r] + sv - hrow(r) : 0.0;
                    const double rs = valid ? fast_rcp(sv) : 0.0;
                    const double dd = lv * rs;
                    const double dsa = valid ? (-rp - gd) : 0.0;
                    const double dla = valid ? (-lv - dd * dsa) : 0.0;
                    const double rl = valid ? fast_rcp(lv) : 0.0;
                    rho_aff = fmax(rho_aff, fmax(-dsa * rs, -dla * rl));
                    const double w = dsa * dla;
                    w_[r] = w;
                    const double c1 = w * rs;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        accb[j] = fma(c1, g[j], accb[j]);
                        accb[D + j] = fma(rs, g[j], accb[D + j]);
                    }
                    accb[2 * D] += sv * dla + lv * dsa;
                    accb[2 * D + 1] += w;
                }
                reduce_to_lds<2 * D + 2>(accb, red, sums, lane);     // M's totals are in registers by now
                rho_aff = wave_reduce<OpMax>(rho_aff);
                const double aaff = rho_aff > 1.0 ? 1.0 / rho_aff : 1.0;
                const double mu_aff = (gap + aaff * sums[2 * D] + aaff * aaff * sums[2 * D + 1]) / ncd;
                double sigma = mu_aff / mu;
                sigma = fmin(sigma * sigma * sigma, 1.0);
                const double smu = sigma * mu;
                {
                    double b2 = lane < D ? rhs_i + sums[lane] - smu * sums[D + lane] : 0.0;
                    rows_forward<D>(mrow, b2, lane);
                    const double xl = rows_backsub_lane<D>(mrow, b2, mdinv, lane);
                    if (lane < D) dxv[lane] = xl;
                }
                lds_fence();
                // ---- pass C: final direction, step length
                double om = (1.0 - aaff) * (1.0 - aaff);
                om = fmin(fmax(om, 1e-4), 1e-2);
                const double tau = 1.0 - om;
                double rho = 0.0;
                for (int r = lane; r < nrp; r += WAVE) {
                    const bool valid = r < nr;
                    const double sv = s_[r], lv = lam_[r];
                    double gd = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) gd = fma(Ht[static_cast<size_t>(j) * nrp + r], dxv[j], gd);
                    const double rp = valid ? gz_[r] + sv - hrow(r) : 0.0;
                    const double rs = valid ? fast_rcp(sv) : 0.0;
                    const double dsk = valid ? (-rp - gd) : 0.0;
                    const double dlk = valid ? (-lv + (smu - w_[r]) * rs - lv * rs * dsk) : 0.0;
                    const double rl = valid ? fast_rcp(lv) : 0.0;
                    rho = fmax(rho, fmax(-dsk * rs, -dlk * rl));
                    ds_[r] = dsk;
                    dl_[r] = dlk;
                }
                rho = wave_reduce<OpMax>(rho);
                const double alpha = rho > tau ? tau / rho : 1.0;
                // ---- pass D: update
                for (int r = lane; r < nr; r += WAVE) {
                    const double sv = s_[r], gz = gz_[r];
                    const double rp = gz + sv - hrow(r);
                    const double dsk = ds_[r];
                    gz_[r] = gz + alpha * (-rp - dsk);
                    s_[r] = sv + alpha * dsk;
                    lam_[r] += alpha * dl_[r];
                }
                if (lane < D) xv[lane] += alpha * dxv[lane];
                lds_fence();
            }
            if (!want_polish) break;

            // ------------------------------------------------------------ hand-over: primal active-set steps
            bool ok = false;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // (s, lam) are read across lanes from here on
            if (stalled) {
                // ---- the collapsed iterate lies ON its active rows (their slack is round-off): feasible ascent directions.
                // Non-negative least squares (Lawson, Hanson) over the active rows A(x): y >= 0 minimising |c - H_A' y|, the
                // residual p then has a_j . p <= 0 on all of A(x) -- an ascent direction that no active row blocks, however
                // many of them meet at x.  p = 0: x is optimal, y its certificate.  Otherwise x moves along p to the first
                // row in the way (a step of positive length: the value rises, no face comes back), which joins A(x).
                // The multipliers stay non-negative by construction (the step from y to the least-squares solution of a new
                // passive set stops where the first one reaches zero, that row leaves), |p| never rises while x stays.
                // p = c - H_W' y carries round-off of size eps |c| along the passive rows, as large as p itself near the
                // end: it is projected onto their null space once more before use, so that p stays a direction of the face
                // (and c . p = |p|^2 > 0) down to |p| ~ 1e-13.
                constexpr double ACT = 1e-11;       // slack <= ACT max(|h_r|, 1): active (the primal feasibility tolerance)
                int m = 0;
                if (lane < D) { xpv[lane] = xv[lane]; yv[lane] = 0.0; }
                lds_fence();
                const int max_rounds = 8 * d + 32;
                for (int round = 0; round < max_rounds; ++round) {
                    double gw[D], srow[D], sdinv = 1.0;
                    if (!lp_face_factor<D>(Ht, nrp, widx, m, GW, gw, srow, sdinv, lane)) break;
                    const double pn = lp_face_residual<D>(m, cv, GW, yv, tv, pv, gw, srow, sdinv, lane);
                    if (pn <= DUAL_TOL_P) { ok = true; break; }
                    // ---- one sweep over the rows: the active row that wants in most, the first row in the way
                    double wbest = 0.0, tbest = INFINITY;
                    int wrow = -1, trow = -1;
                    bool infeas = false;
                    for (int r = lane; r < nr; r += WAVE) {
                        double gx = 0.0, gp = 0.0;
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            const double g = Ht[static_cast<size_t>(j) * nrp + r];
                            gx = fma(g, xpv[j], gx);
                            gp = fma(g, pv[j], gp);
                        }
                        bool isw = false;
                        for (int a = 0; a < m; ++a) isw = isw || (widx[a] == r);
                        const double hr = hrow(r);
                        const double sl = hr - gx;
                        const double hi = fmax(fabs(hr), 1.0);
                        const bool active = sl <= ACT * hi;
                        infeas = infeas || sl < -ACT * hi;
                        if (!isw && active && gp > wbest) { wbest = gp; wrow = r; }
                        if (!isw && !active && gp > 1e-14) {
                            const double t = sl / gp;
                            if (t < tbest) { tbest = t; trow = r; }
                        }
                    }
                    const double wmax = wave_reduce<OpMax>(wbest);
#ifdef TMPC_LP_DEBUG
                    {
                        double ol = lane < D ? cv[lane] * xpv[lane] : 0.0;
                        ol = wave_reduce<OpSum>(ol);
                        const double tm = wave_reduce<OpMin>(tbest);
                        double ymn = lane < m ? yv[lane] : INFINITY;
                        ymn = wave_reduce<OpMin>(ymn);
                        if (lane == 0) printf("lp %lld   nnls %d m %d pn %.3e wmax %.3e obj %.12e tmin %.3e ymin %.3e\n", (long long)b, round, m, pn, wmax, ol, tm, ymn);
                    }
#endif
                    // a long step along a short p took x across a row (what p keeps of the round-off of c - GW' y, times the
                    // length of the step): no certificate from here
                    if (__ballot(infeas) != 0ull) break;
                    if (wmax > 1e-14) {
                        // a_j . p > 0 on an active row: j joins the passive set
                        if (m >= d) break;
                        const unsigned long long bal = __ballot(wbest == wmax);
                        const int j = __builtin_amdgcn_readlane(wrow, __ffsll(static_cast<long long>(bal)) - 1);
                        const bool bad = !lp_nnls_add<D>(Ht, nrp, d, j, cv, widx, m, GW, yv, tv, pv, lane);
                        // an active row in the way that the passive set cannot take (dependent on it to working precision):
                        // no direction to go on with
                        if (bad) break;
                        continue;
                    }
                    // ---- no active row in the way: along p to the first inactive one
                    const double tmin = wave_reduce<OpMin>(tbest);
                    if (!(tmin < INFINITY)) { st = TMPC_STATUS_UNBOUNDED; break; }
                    if (lane < D) xpv[lane] += tmin * pv[lane];
                    lds_fence();
                }
            } else {
                // candidates: rows with lam > s, at most one per lane
                int ncand = 0;
                for (int r0 = 0; r0 < nrp; r0 += WAVE) {
                    const int r = r0 + lane;
                    const bool f = r < nr && lam_[r] > s_[r];
                    const unsigned long long bal = __ballot(f);
                    const int pos = ncand + __popcll(bal & ((1ull << lane) - 1ull));
                    if (f && pos < WAVE) cidx[pos] = r;
                    ncand += __popcll(bal);
                }
                lds_fence();
                int m = 0;
                if (ncand <= WAVE) {
                    // the working set to start from: non-negative least squares over the candidates, y >= 0 minimising
                    // |c - H_cand' y| (the rows that end with y > 0).  Among 43 candidates in 28 dimensions "the most active
                    // ones that are independent" are nearly dependent rows with multipliers of +-30; here a row comes in
                    // only while it reduces the residual
                    const bool have = lane < ncand;
                    const int myr = have ? cidx[lane] : 0;
                    bool alive = have;
                    if (lane < D) yv[lane] = 0.0;
                    lds_fence();
                    double pn_sel = INFINITY;
                    bool cert = false;
                    for (int pick = 0;; ++pick) {
                        double gw[D], srow[D], sdinv = 1.0;
                        if (!lp_face_factor<D>(Ht, nrp, widx, m, GW, gw, srow, sdinv, lane)) break;
                        pn_sel = lp_face_residual<D>(m, cv, GW, yv, tv, pv, gw, srow, sdinv, lane);
                        if (pick >= 2 * d + 8 || m >= d) break;
                        bool isw = false;
                        for (int a = 0; a < m; ++a) isw = isw || (widx[a] == myr);
                        double gp = 0.0;                        // (the candidate's row is read again per pick: D registers less)
#pragma unroll
                        for (int j = 0; j < D; ++j) gp = fma(Ht[static_cast<size_t>(j) * nrp + myr], pv[j], gp);
                        const double key = (alive && !isw) ? gp : -INFINITY;
                        const double wmax = wave_reduce<OpMax>(key);
                        if (!(wmax > 1e-13) || !(wmax > 1e-3 * pn_sel)) break;
                        const unsigned long long bal = __ballot(key == wmax);
                        const int pl = __ffsll(static_cast<long long>(bal)) - 1;
                        const int j = __builtin_amdgcn_readlane(myr, pl);
                        if (!lp_nnls_add<D>(Ht, nrp, d, j, cv, widx, m, GW, yv, tv, pv, lane) && lane == pl) alive = false;
                    }
                    // ---- certificate without a vertex: x (the iterate: strictly feasible) and y >= 0 on the rows of W with
                    // c = H_W' y + p bound the value from both sides, c . x <= val <= c . x + sum y_i s_i + |p . (x* - x)|.
                    // On a degenerate face (rows active with multiplier zero) no projection onto W is feasible, but once the
                    // gap of the iterate is at 1e-12 this closes by itself.
                    {
                        double gl = lane < m ? yv[lane] * fmax(s_[widx[lane]], 0.0) : 0.0;
                        gl = wave_reduce<OpSum>(gl);
                        double x1 = lane < D ? fabs(xv[lane]) : 0.0, ol = lane < D ? cv[lane] * xv[lane] : 0.0;
                        x1 = wave_reduce<OpSum>(x1);
                        ol = wave_reduce<OpSum>(ol);
#ifdef TMPC_LP_DEBUG
                        if (lane == 0) printf("lp %lld   selection m %d pn %.3e gapW %.3e |x|_1 %.3e\n", (long long)b, m, pn_sel, gl, x1);
#endif
                        cert = pn_sel <= DUAL_TOL_P && gl + pn_sel * fmax(x1, 1.0) <= GAP_TOL * fmax(fabs(ol), 1.0);
                    }
                    {
                    lds_fence();
                    if (lane < D) xpv[lane] = xv[lane];
                    lds_fence();
                    // (with the certificate in hand: one projection onto the face of W, for the vertex value if it is feasible)
                    const int max_rounds = cert ? 1 : d + 12;
                    for (int round = 0; round < max_rounds; ++round) {
                        // ---- projection onto the face of the working rows, multipliers by least squares
                        const bool inw = lane < m;
                        const int wr = inw ? widx[lane] : 0;
                        double gw[D];
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            gw[j] = inw ? Ht[static_cast<size_t>(j) * nrp + wr] : 0.0;
                            if (lane < D) GW[lane * D + j] = gw[j];
                        }
                        const double hw = inw ? hrow(wr) : 0.0;
                        lds_fence();
                        double srow[D], sdinv = 1.0;
#pragma unroll
                        for (int c2 = 0; c2 < D; ++c2) {
                            double v = 0.0;
#pragma unroll
                            for (int j = 0; j < D; ++j) v = fma(gw[j], GW[c2 * D + j], v);
                            srow[c2] = (inw && c2 < m) ? v : ((c2 == lane) ? 1.0 : 0.0);
                        }
                        double bdummy = 0.0;
                        if (!rows_factor<D>(srow, bdummy, sdinv, lane)) {
#ifdef TMPC_LP_DEBUG
                            if (lane == 0) printf("lp %lld   round %d m %d: working rows dependent\n", (long long)b, round, m);
#endif
                            break;
                        }
                        if (lane < D) yv[lane] = 0.0;
                        lds_fence();
                        for (int sweep = 0; sweep < 2; ++sweep) {
                            // xp -= GW' S^-1 (GW xp - hW)
                            double b1 = 0.0;
                            if (inw) {
#pragma unroll
                                for (int j = 0; j < D; ++j) b1 = fma(gw[j], xpv[j], b1);
                                b1 -= hw;
                            }
                            rows_forward<D>(srow, b1, lane);
                            const double tl = rows_backsub_lane<D>(srow, b1, sdinv, lane);
                            if (lane < D) tv[lane] = inw ? tl : 0.0;
                            lds_fence();
                            if (lane < D) {
                                double v = 0.0;
                                for (int a = 0; a < m; ++a) v = fma(GW[a * D + lane], tv[a], v);
                                xpv[lane] -= v;
                            }
                            lds_fence();
                            // y += S^-1 GW (c - GW' y)
                            if (lane < D) {
                                double v = cv[lane];
                                for (int a = 0; a < m; ++a) v = fma(-GW[a * D + lane], yv[a], v);
                                pv[lane] = v;
                            }
                            lds_fence();
                            double b2 = 0.0;
                            if (inw) {
#pragma unroll
                                for (int j = 0; j < D; ++j) b2 = fma(gw[j], pv[j], b2);
                            }
                            rows_forward<D>(srow, b2, lane);
                            const double yl = rows_backsub_lane<D>(srow, b2, sdinv, lane);
                            if (inw) yv[lane] += yl;
                            lds_fence();
                        }
                        double pl_ = 0.0, ymin_l = INFINITY, yabs_l = 0.0;
                        if (lane < D) {
                            double v = cv[lane];
                            for (int a = 0; a < m; ++a) v = fma(-GW[a * D + lane], yv[a], v);
                            pv[lane] = v;
                            pl_ = fabs(v);
                        }
                        if (inw) { ymin_l = yv[lane]; yabs_l = fabs(yv[lane]); }
                        lds_fence();
                        const double pn = wave_reduce<OpMax>(pl_);
                        const double ymin = wave_reduce<OpMin>(ymin_l), ymax = fmax(wave_reduce<OpMax>(yabs_l), 1.0);
                        // ---- one sweep over all rows: worst violation, and the ratio test along p
                        double vbest = 0.0, tbest = INFINITY;
                        int vrow = -1, trow = -1;
                        for (int r = lane; r < nr; r += WAVE) {
                            double gx = 0.0, gp = 0.0;
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                const double g = Ht[static_cast<size_t>(j) * nrp + r];
                                gx = fma(g, xpv[j], gx);
                                gp = fma(g, pv[j], gp);
                            }
                            bool isw = false;
                            for (int a = 0; a < m; ++a) isw = isw || (widx[a] == r);
                            const double hr = hrow(r);
                            const double rr = gx - hr;
                            const double hi = fmax(fabs(hr), 1.0);
                            if (!isw && rr > 1e-11 * hi && rr / hi > vbest) { vbest = rr / hi; vrow = r; }
                            if (!isw && gp > 1e-13) {
                                const double t = fmax(-rr, 0.0) / gp;
                                if (t < tbest) { tbest = t; trow = r; }
                            }
                        }
                        const double vmax = wave_reduce<OpMax>(vbest);
#ifdef TMPC_LP_DEBUG
                        if (lane == 0) printf("lp %lld   round %d m %d ncand %d pn %.3e ymin %.3e ymax %.3e vmax %.3e\n", (long long)b, round, m, ncand, pn, ymin, ymax, vmax);
#endif
                        if (vmax > 0.0) {
                            // a violated row enters; it replaces the working row it is nearly parallel to
                            const unsigned long long bal = __ballot(vbest == vmax);
                            const int j = __builtin_amdgcn_readlane(vrow, __ffsll(static_cast<long long>(bal)) - 1);
                            double dotl = 0.0;
                            if (inw) {
#pragma unroll
                                for (int k = 0; k < D; ++k) dotl = fma(gw[k], Ht[static_cast<size_t>(k) * nrp + j], dotl);
                                dotl = fabs(dotl);
                            }
                            const double dmax = wave_reduce<OpMax>(dotl);
                            if (m > 0 && dmax > 0.99) {
                                const unsigned long long b2 = __ballot(inw && dotl == dmax);
                                const int i = __ffsll(static_cast<long long>(b2)) - 1;
                                if (lane == 0) widx[i] = j;
                            } else if (m < d) {
                                if (lane == 0) widx[m] = j;
                                ++m;
                            } else {
                                break;
                            }
                            lds_fence();
                            continue;
                        }
                        if (m > 0 && ymin < -DUAL_TOL_Y * ymax) {
                            // the most negative multiplier leaves
                            const unsigned long long bal = __ballot(inw && ymin_l == ymin);
                            const int i = __ffsll(static_cast<long long>(bal)) - 1;
                            const int moved = (lane > i && lane < m) ? widx[lane] : 0;
                            lds_fence();
                            if (lane > i && lane < m) widx[lane - 1] = moved;
                            --m;
                            lds_fence();
                            continue;
                        }
                        if (pn > DUAL_TOL_P) {
                            // c has a component along the face: follow it to the blocking row
                            const double tmin = wave_reduce<OpMin>(tbest);
                            if (!(tmin < INFINITY)) { st = TMPC_STATUS_UNBOUNDED; break; }
                            if (m >= d) break;
                            const unsigned long long bal = __ballot(tbest == tmin);
                            const int j = __builtin_amdgcn_readlane(trow, __ffsll(static_cast<long long>(bal)) - 1);
                            if (lane < D) xpv[lane] += tmin * pv[lane];
                            if (lane == 0) widx[m] = j;
                            ++m;
                            lds_fence();
                            continue;
                        }
                        ok = true;
                        break;
                    }
                    if (!ok && cert && st != TMPC_STATUS_UNBOUNDED) {
                        if (lane < D) xpv[lane] = xv[lane];
                        lds_fence();
                        ok = true;
                    }
                    }
                }
            }
#ifdef TMPC_LP_DEBUG
            if (lane == 0) printf("lp %lld hand-over at it %d tol %.1e: ok %d st %d\n", (long long)b, it, try_tol, (int)ok, st);
#endif
            if (st == TMPC_STATUS_UNBOUNDED) break;
            if (ok) { st = TMPC_STATUS_OPTIMAL; from_polish = true; break; }
            if (stalled) break;                                          // MAX_ITER: the iterate, without a certificate
            if (try_tol <= 1e-12) {
                if (rdn_last <= DUAL_TOL_P) { st = TMPC_STATUS_OPTIMAL; break; }
                collapse = true;                                         // on until the gap has collapsed, then the steps above
            } else {
                try_tol *= 1e-2;
            }
        }

        // ---- outputs (scaled back: x by hm, the value by |c| hm)
        {
            const double *xs = from_polish ? xpv : xv;
            double ol = lane < D ? cv[lane] * xs[lane] : 0.0;
            ol = wave_reduce<OpSum>(ol);
            value = cn * p.hm * ol;
            const bool good = st == TMPC_STATUS_OPTIMAL || st == TMPC_STATUS_MAX_ITER;
            if (lane == 0) {
                val[b] = good ? value : (st == TMPC_STATUS_UNBOUNDED ? INFINITY : NAN);
                status[b] = st;
                iters[b] = it;
            }
            if (xout && lane < d) xout[b * d + lane] = good ? p.hm * xs[lane] : NAN;
        }
        lds_fence();
    }
}

template <int D, bool STAGED>
hipError_t launch_lp_ds(const LpDevice &p, int64_t B, int nblocks, const double *C, const int32_t *relax, double *ws, double *val,
                        double *xout, int32_t *status, int32_t *iters, hipStream_t stream) {
    const size_t lds = (static_cast<size_t>(LpLds<D>::TOTAL) * LP_WPB + (STAGED ? static_cast<size_t>(D) * p.nrp : 0)) * sizeof(double);
#ifdef TMPC_HOST_SIM
    // tests/wavesim: one workgroup (four waves, four LPs in flight) on the host execution model takes the whole batch
    (void)nblocks; (void)stream;
    sim::Dim3 bi, gd;
    bi.x = bi.y = bi.z = 0;
    unsigned long long counter = 0;
    LpDevice ps = p;
    ps.next_item = &counter;
    sim::run_block(WAVE * LP_WPB, lds, bi, gd, [&]() { lp_kernel<D, STAGED>(ps, B, C, relax, ws, val, xout, status, iters); });
    return hipSuccess;
#else
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lp_kernel<D, STAGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((lp_kernel<D, STAGED>), dim3(nblocks), dim3(WAVE * LP_WPB), lds, stream, p, B, C, relax, ws, val, xout, status, iters);
    return hipGetLastError();
#endif
}

template <int D>
hipError_t launch_lp_d(const LpDevice &p, int64_t B, int nblocks, const double *C, const int32_t *relax, double *ws, double *val,
                       double *xout, int32_t *status, int32_t *iters, hipStream_t stream) {
    // stage H' in LDS when it fits beside the per-wave workspaces (and leaves room for a second workgroup on the CU)
    const size_t hbytes = static_cast<size_t>(D) * p.nrp * sizeof(double);
    const size_t base = static_cast<size_t>(LpLds<D>::TOTAL) * LP_WPB * sizeof(double);
    if (hbytes <= 64 * 1024 && base + hbytes <= 156 * 1024)
        return launch_lp_ds<D, true>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
    return launch_lp_ds<D, false>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
}

}  // namespace

int lp_padded_dim(int d) { return d <= 4 ? 4 : (d <= 8 ? 8 : (d <= 12 ? 12 : (d <= 16 ? 16 : (d <= 32 ? 32 : -1)))); }
int lp_waves_per_block() { return LP_WPB; }
int lp_workspace_arrays() { return LP_ARR; }

hipError_t launch_lp(const LpDevice &p, int64_t B, int nblocks, const double *C, const int32_t *relax, double *ws, double *val,
                     double *xout, int32_t *status, int32_t *iters, hipStream_t stream) {
    // (tests/wavesim: the sanitizer builds stop at TMPC_LP_MAX_DIM = 8 or 12; d = 32 alone takes them half an hour)
#ifndef TMPC_LP_MAX_DIM
#define TMPC_LP_MAX_DIM 32
#endif
    switch (lp_padded_dim(p.d)) {
    case 4: return launch_lp_d<4>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
    case 8: return launch_lp_d<8>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
#if TMPC_LP_MAX_DIM >= 12
    case 12: return launch_lp_d<12>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
#endif
#if TMPC_LP_MAX_DIM >= 32
    case 16: return launch_lp_d<16>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
    case 32: return launch_lp_d<32>(p, B, nblocks, C, relax, ws, val, xout, status, iters, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}

}  // namespace tmpc
