// C ABI of libtmpc_hip.so (declared in include/tmpc.h).  Host side only: condenses the
// problem (tmpc_condense.cpp), keeps it resident in HBM and enqueues the solve kernels
// (tmpc_kernels.hip) on the handle's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "tmpc_condense.hpp"
#include "tmpc_device.hpp"

namespace {

thread_local std::string g_create_error;

struct Variant {
    tmpc::Condensed c;
    tmpc::DeviceQP d{};
    tmpc::KernelShape shape;
    bool wave_ok = false;        // a compiled one-wave-per-QP shape covers this variant
    tmpc::DeviceQP db{};         // same model with Hs / Hinv padded for the block kernel
    tmpc::BlockQP bq{};
    const tmpc::BlockArgs *bargs = nullptr;   // {db, bq} in device memory: what solve_block_kernel reads (tmpc_device.hpp)
    int tiles = 0;               // block kernel: NVP / 16 (0: not available)
    std::vector<void *> dev;     // device allocations of this variant
    std::vector<size_t> dev_bytes;       // their sizes (tmpc_debug_dump_layout)
};

}  // namespace

struct tmpc_handle {
    int device = 0;
    int n_cu = 0;
    int nvariants = 0;
    int nx = 0, nu = 0, N = 0;
    Variant v[2];
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // event pairs of the launches since the last tmpc_kernel_ms_total(reset): per-launch device time
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t pool_used = 0;
    std::vector<double> hA, hB, hK, hKanc;   // host copies for the closed-loop entry point
    // closed-loop state lives in one grow-only arena (25 hipMalloc / hipFree per call cost several milliseconds)
    char *mc_arena = nullptr;
    size_t mc_arena_bytes = 0;
    int plant = TMPC_PLANT_LINEAR, plant_substeps = 10;
    int actuator = TMPC_ACTUATOR_CONSISTENT;
    tmpc::WorkCounter wc;        // work counters of the wave kernel's launches (tmpc_device.hpp)
    long long mc_capture = -1;   // trajectory recorded by the next tmpc_mc_run (-1: none)
    double *mc_cap_dev = nullptr; // its record in the arena: [mc_cap_T][2 nx + nu]
    int mc_cap_T = 0;
    int mc_warm = 0;             // closed loop: hand every solve the working set of the trajectory's previous solve of the same variant
    int mc_fused = TMPC_MC_FUSED_AUTO;   // closed loop: one fused launch for all T steps (tmpc_mc_set_fused)
    int mc_last_fused = 0;       // what the last tmpc_mc_run did
    double plant_par[7] = {0, 0, 0, 0, 0, 0, 0};
    int kernel_path = TMPC_PATH_AUTO;
    int blk_blocks = 0;          // workgroups the block-kernel workspace is sized for
    int blk_ncp = 0;
    double *blk_ws = nullptr;
    // staging buffers for the host-pointer entry point: ONE device block, inputs [x | ref | variant] then outputs
    // [u | x0 | ss | status | iters | x_nom], and a pinned host mirror of it -- a call moves its inputs with one DMA and its
    // outputs with one (round 3: nine hipMemcpyAsync from / to pageable memory per call, 60 % of the time of a call at batch 1)
    int64_t cap = 0;
    char *stage_dev = nullptr, *stage_pin = nullptr;
    size_t stage_in_bytes = 0, stage_out_bytes = 0, stage_out_core = 0;      // (core = the outputs without x_nom)
    size_t off_r = 0, off_var = 0, off_x0 = 0, off_ss = 0, off_st = 0, off_it = 0, off_xn = 0;      // offsets within the input / output parts
    double *d_x = nullptr, *d_r = nullptr, *d_u = nullptr, *d_x0 = nullptr, *d_ss = nullptr, *d_xn = nullptr;
    uint8_t *d_var = nullptr;
    int32_t *d_st = nullptr, *d_it = nullptr;
    // (s, lambda) of every resident wave at its hand-over to the refinement (DeviceQP::save)
    float *save_buf = nullptr;
    size_t save_bytes = 0;
    // per-solve durations (tmpc_set_solve_timing): one tick count per instance of the last call
    int want_ticks = 0;
    int64_t ticks_cap = 0, ticks_n = 0;
    long long *d_ticks = nullptr;
    long long *mc_tick_sum = nullptr, *mc_tick_max = nullptr;      // inside mc_arena, of the last tmpc_mc_run
    double *mc_err2_phys = nullptr;                                 // likewise (nonlinear plant only)
    int mc_rng_on = 0;                                              // tmpc_mc_set_device_rng
    uint64_t mc_rng_seed = 0;
    int64_t mc_rng_first = 0;
    std::vector<double> mc_w_bound;
    int64_t mc_phys_B = 0;
    int64_t mc_tick_B = 0;
    std::string err;
};

namespace {

#define HIP_TRY(h, expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return TMPC_E_DEVICE;                                                          \
        }                                                                                  \
    } while (0)

template <class T>
int upload(tmpc_handle *h, Variant &v, const T *src, size_t n, const T **dst) {
    void *p = nullptr;
    if (h->device < 0) {
        // host-only handle: the layouts the kernels read are kept in host memory (tmpc_debug_layout; tests/wavesim runs
        // the kernel sources on the CPU against them)
        p = std::malloc((n ? n : 1) * sizeof(T));
        if (!p) { h->err = "out of memory"; return TMPC_E_NOMEM; }
        v.dev.push_back(p);
        v.dev_bytes.push_back(n * sizeof(T));
        if (n) std::memcpy(p, src, n * sizeof(T));
        *dst = static_cast<const T *>(p);
        return TMPC_OK;
    }
    HIP_TRY(h, hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    v.dev.push_back(p);
    v.dev_bytes.push_back(n * sizeof(T));
    if (n) HIP_TRY(h, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return TMPC_OK;
}

int upload_common(tmpc_handle *h, Variant &v, const tmpc_problem &p, tmpc::DeviceQP &d, int NVP) {
    const tmpc::Condensed &c = v.c;
    const int nx = c.nx;
    std::vector<double> Hs(static_cast<size_t>(NVP) * NVP, 0.0), Hinv(Hs.size(), 0.0);
    for (int i = 0; i < NVP; ++i)
        for (int j = 0; j < NVP; ++j) {
            const bool in = i < c.nv && j < c.nv;
            Hs[static_cast<size_t>(i) * NVP + j] = in ? c.Hs(i, j) : (i == j ? 1.0 : 0.0);
            Hinv[static_cast<size_t>(i) * NVP + j] = in ? c.Hinv(i, j) : (i == j ? 1.0 : 0.0);
        }
    d.nx = c.nx; d.nu = c.nu; d.N = c.N; d.nv = c.nv; d.nc = c.nc; d.npar = c.npar; d.nth = c.nth;
    d.off_theta = c.off_theta; d.off_x0 = c.off_x0; d.off_aux = c.off_aux;
    d.max_iter = p.max_iter > 0 ? p.max_iter : 60;
    d.tol = p.tol > 0 ? p.tol : 1e-7;
    d.always_infeasible = c.always_infeasible ? 1 : 0;
    d.dbg = nullptr;
    d.ticks = nullptr;
    d.save = nullptr;
    int rc;
    if ((rc = upload(h, v, Hs.data(), Hs.size(), &d.Hs))) return rc;
    if ((rc = upload(h, v, Hinv.data(), Hinv.size(), &d.Hinv))) return rc;
    if ((rc = upload(h, v, c.F1s.a.data(), c.F1s.a.size(), &d.F1s))) return rc;
    if ((rc = upload(h, v, c.F2s.a.data(), c.F2s.a.size(), &d.F2s))) return rc;
    if ((rc = upload(h, v, c.gp0.data(), c.gp0.size(), &d.gp0))) return rc;
    if ((rc = upload(h, v, c.Ep.a.data(), c.Ep.a.size(), &d.Ep))) return rc;
    if ((rc = upload(h, v, c.Dv.data(), c.Dv.size(), &d.Dv))) return rc;
    d.Tzs = d.Txf = nullptr;
    d.nvf = c.nvf;
    if (!c.Tz.a.empty()) {
        std::vector<double> Tzs(c.Tz.a.size());
        for (int i = 0; i < c.Tz.r; ++i)
            for (int j = 0; j < c.Tz.c; ++j) Tzs[static_cast<size_t>(i) * c.Tz.c + j] = c.Tz(i, j) * c.Dv[j];
        if ((rc = upload(h, v, Tzs.data(), Tzs.size(), &d.Tzs))) return rc;
        if ((rc = upload(h, v, c.Tx.a.data(), c.Tx.a.size(), &d.Txf))) return rc;
    }
    if ((rc = upload(h, v, c.Mth.a.data(), c.Mth.a.size(), &d.Mth))) return rc;
    if ((rc = upload(h, v, p.A, static_cast<size_t>(nx) * nx, &d.A))) return rc;
    if ((rc = upload(h, v, p.B, static_cast<size_t>(nx) * c.nu, &d.B))) return rc;
    return TMPC_OK;
}

// 1 / (g_r Hs^-1 g_r') for every row of the scaled problem: the multiplier of the QP with row r alone is (violation of row r at
// the unconstrained minimiser) times this -- the scale of the multipliers the interior-point phase starts from
std::vector<double> single_row_curvature_inv(const tmpc::Condensed &c) {
    std::vector<double> ci(c.nc, 0.0), t(c.nv);
    for (int r = 0; r < c.nc; ++r) {
        double q = 0.0;
        for (int i = 0; i < c.nv; ++i) {
            double v = 0.0;
            for (int j = 0; j < c.nv; ++j) v += c.Hinv(i, j) * c.Gs(r, j);
            q += v * c.Gs(r, i);
        }
        ci[r] = q > 0.0 ? 1.0 / q : 0.0;
    }
    return ci;
}

// one-wave-per-QP path (tmpc_kernels.hip): functionals (a row and, where it exists, its mirror row) in 64-wide slots of four
// kinds -- dense paired, dense single, factored paired, factored single -- and per row side the right-hand side data
int upload_wave(tmpc_handle *h, Variant &v, const tmpc_problem &p) {
    const tmpc::Condensed &c = v.c;
    const int nx = c.nx;
    auto in_fact = [&](int r) { return c.ncc > 0 && r >= c.fb0 && r < c.fb0 + c.ncc; };
    // three layouts, most structured first: pairs + factored block; single rows + factored block; single rows, all dense
    bool use_pairs = false, use_fact = false;
    std::vector<std::pair<int, int>> dpair, cpair;     // (row, mirror row)
    std::vector<int> dsing, csing;
    bool found = false;
    for (int attempt = 0; attempt < 3 && !found; ++attempt) {
        use_pairs = attempt == 0;
        use_fact = attempt <= 1 && c.ncc > 0;
        dpair.clear(); cpair.clear(); dsing.clear(); csing.clear();
        for (int r = 0; r < c.nc; ++r) {
            const bool f = use_fact && in_fact(r);
            const int q = (use_pairs && !c.mirror.empty()) ? c.mirror[r] : -1;
            if (q >= 0 && q < r) continue;                       // second member of a pair: placed with the first
            if (q >= 0) (f ? cpair : dpair).emplace_back(r, q);
            else (f ? csing : dsing).push_back(r);
        }
        found = tmpc::pick_config(c.nv, static_cast<int>(dpair.size()), static_cast<int>(dsing.size()), use_fact ? c.kc : 0,
                                  static_cast<int>(cpair.size()), static_cast<int>(csing.size()), &v.shape);
    }
    if (found) {
        // the dense functionals in use must fit the LDS next to the workspaces of the shape's waves
        const int last = dsing.empty() ? static_cast<int>(dpair.size()) : v.shape.dp * 64 + static_cast<int>(dsing.size());
        if (tmpc::lds_bytes(v.shape, 4 * ((last + 3) / 4)) > 160 * 1024) found = false;
    }
    if (!found) return TMPC_OK;                                  // wave_ok stays false: the block kernel takes the variant
    const int NVP = v.shape.nvp, DP = v.shape.dp, DS = v.shape.ds, KCP = v.shape.kcp, CP = v.shape.cp, CS = v.shape.cs;
    const int NDP = (DP + DS) * 64, NCCP = (CP + CS) * 64, RS = 2 * DP + DS + 2 * CP + CS;
    const int kc = use_fact ? c.kc : 0;
    const int LDG = 16 * ((NVP + 1 + 15) / 16) + 1;        // Shape::LDG of tmpc_kernels.hip
    std::vector<double> Gt(static_cast<size_t>(NDP) * LDG + 1, 0.0), Hct(static_cast<size_t>(KCP) * NCCP + 1, 0.0),
        Psi(static_cast<size_t>(KCP) * NVP + 1, 0.0), g0p(static_cast<size_t>(RS) * 64, 1.0), Esp(static_cast<size_t>(RS) * 64 * nx, 0.0),
        cip(static_cast<size_t>(RS) * 64, 0.0);
    const std::vector<double> ci_rows = single_row_curvature_inv(c);
    std::vector<uint32_t> vmask(64, 0u);
    std::vector<int32_t> row_of(static_cast<size_t>(RS) * 64, -1);
    auto put_side = [&](int side, int lane, int row) {
        const size_t sl = static_cast<size_t>(side) * 64 + lane;
        g0p[sl] = c.g0s[row];
        cip[sl] = ci_rows[row];
        for (int j = 0; j < nx; ++j) Esp[static_cast<size_t>(j) * RS * 64 + sl] = c.Es(row, j);
        vmask[lane] |= 1u << side;
        row_of[sl] = row;
    };
    auto put_dense = [&](int fslot, int lane, int row) {
        for (int j = 0; j < c.nv; ++j) Gt[static_cast<size_t>(fslot * 64 + lane) * LDG + j] = c.Gs(row, j);
    };
    auto put_fact = [&](int fslot, int lane, int row) {
        for (int a = 0; a < kc; ++a) Hct[static_cast<size_t>(a) * NCCP + fslot * 64 + lane] = c.Hc(row - c.fb0, a);
    };
    for (size_t f = 0; f < dpair.size(); ++f) {
        const int k = static_cast<int>(f / 64), lane = static_cast<int>(f % 64);
        put_dense(k, lane, dpair[f].first);
        put_side(2 * k, lane, dpair[f].first);
        put_side(2 * k + 1, lane, dpair[f].second);
    }
    for (size_t f = 0; f < dsing.size(); ++f) {
        const int k = static_cast<int>(f / 64), lane = static_cast<int>(f % 64);
        put_dense(DP + k, lane, dsing[f]);
        put_side(2 * DP + k, lane, dsing[f]);
    }
    const int cb = 2 * DP + DS;
    for (size_t f = 0; f < cpair.size(); ++f) {
        const int k = static_cast<int>(f / 64), lane = static_cast<int>(f % 64);
        put_fact(k, lane, cpair[f].first);
        put_side(cb + 2 * k, lane, cpair[f].first);
        put_side(cb + 2 * k + 1, lane, cpair[f].second);
    }
    for (size_t f = 0; f < csing.size(); ++f) {
        const int k = static_cast<int>(f / 64), lane = static_cast<int>(f % 64);
        put_fact(CP + k, lane, csing[f]);
        put_side(cb + 2 * CP + k, lane, csing[f]);
    }
    for (int a = 0; a < kc; ++a)
        for (int j = 0; j < c.nv; ++j) Psi[static_cast<size_t>(a) * NVP + j] = c.Psi(a, j);
    tmpc::DeviceQP &d = v.d;
    int rc;
    if ((rc = upload_common(h, v, p, d, NVP))) return rc;
    d.nd = use_fact ? c.nd : c.nc; d.ncc = use_fact ? c.ncc : 0; d.kc = kc;
    {
        // k-steps (4 functionals each) of the MFMA pass over the dense functionals: up to the last one in use
        const int last = dsing.empty() ? static_cast<int>(dpair.size()) : DP * 64 + static_cast<int>(dsing.size());
        d.nks = (last + 3) / 4;
    }
    if ((rc = upload(h, v, Gt.data(), Gt.size(), &d.Gt))) return rc;
    if ((rc = upload(h, v, Hct.data(), Hct.size(), &d.Hct))) return rc;
    if ((rc = upload(h, v, Psi.data(), Psi.size(), &d.Psi))) return rc;
    if ((rc = upload(h, v, g0p.data(), g0p.size(), &d.g0p))) return rc;
    if ((rc = upload(h, v, Esp.data(), Esp.size(), &d.Esp))) return rc;
    if ((rc = upload(h, v, cip.data(), cip.size(), &d.cip))) return rc;
    if ((rc = upload(h, v, vmask.data(), vmask.size(), &d.vmask))) return rc;
    if ((rc = upload(h, v, row_of.data(), row_of.size(), &d.row_of))) return rc;
#ifdef TMPC_STAMPS
    {
        void *dbgp = nullptr;
        HIP_TRY(h, hipMalloc(&dbgp, 16 * sizeof(long long)));
        HIP_TRY(h, hipMemset(dbgp, 0, 16 * sizeof(long long)));
        v.dev.push_back(dbgp);
        d.dbg = static_cast<long long *>(dbgp);
    }
#endif
    v.wave_ok = true;
    return TMPC_OK;
}

// workgroup-per-QP path (tmpc_block.hip): every row dense, nv padded to a multiple of 16
int upload_block(tmpc_handle *h, Variant &v, const tmpc_problem &p) {
    const tmpc::Condensed &c = v.c;
    v.tiles = tmpc::block_tiles(c.nv);
    if (v.tiles == 0) return TMPC_OK;
    const int NVP = 16 * v.tiles, nx = c.nx;
    // the Z rows of a free initial state touch x_0 only: the block kernel treats them as a narrow class when there are many
    const int nz4 = (c.off_x0 >= 0 && c.nz >= 256) ? (c.nz / 4) * 4 : 0;
    // Functionals (tmpc_device.hpp, BlockQP): when every row has its mirror row (the two sides of a box-type constraint; exact
    // to 1e-13 after scaling, Condensed::mirror) a row of G serves both, and the G-sized passes read half the rows.
    // TMPC_BLOCK_PAIRS=0 (developer knob) keeps a row of G per constraint row.
    bool paired = nz4 == 0 && c.nc >= 2 && static_cast<int>(c.mirror.size()) == c.nc;
    for (int r = 0; paired && r < c.nc; ++r) paired = c.mirror[r] >= 0 && c.mirror[r] < c.nc && c.mirror[r] != r && c.mirror[c.mirror[r]] == r;
    if (const char *e = std::getenv("TMPC_BLOCK_PAIRS")) paired = paired && std::atoi(e) != 0;
    // rows of G: constraint rows, or the first member of every pair
    std::vector<int> grow;
    for (int r = 0; r < c.nc; ++r)
        if (!paired || c.mirror[r] > r) grow.push_back(r);
    const int ng = static_cast<int>(grow.size()), ngp = (ng + 63) / 64 * 64;
    const int mir = paired ? ngp : 0, ncp = paired ? 2 * ngp : ngp;
    // Staircase of the condensed constraints: the rows of stage k act on u_0 .. u_k only, so the leading rows of G are
    // zero beyond a few 16-column tiles.  The general rows are ordered by the number of tiles they reach (stable), and the
    // kernel skips the tiles / columns a row does not touch (exact: the skipped entries are zero).
    std::vector<int> ext(ng, 1), order(ng);
    for (int k = 0; k < ng; ++k) {
        int last = 0;
        for (int j = 0; j < c.nv; ++j)
            if (c.Gs(grow[k], j) != 0.0) last = j;
        ext[k] = last / 16 + 1;
        order[k] = k;
    }
    std::stable_sort(order.begin() + nz4, order.end(), [&](int a, int b) { return ext[a] < ext[b]; });
    // (Grm: the NVP rows of the scaled Hessian, identity on the padding, follow the rows of G -- gt_products adds Hs z in its pass)
    std::vector<double> Grm(static_cast<size_t>(ngp + NVP) * NVP, 0.0), Gcm(static_cast<size_t>(ngp) * NVP, 0.0), g0(ncp, 1.0),
        Es(static_cast<size_t>(ncp) * nx, 0.0);
    for (int i = 0; i < NVP; ++i)
        for (int j = 0; j < NVP; ++j)
            Grm[static_cast<size_t>(ngp + i) * NVP + j] = (i < c.nv && j < c.nv) ? c.Hs(i, j) : (i == j ? 1.0 : 0.0);
    std::vector<double> Gw(paired ? static_cast<size_t>(ncp) * NVP : 0, 0.0), GH(static_cast<size_t>(ncp) * NVP, 0.0);
    std::vector<int32_t> ncols(ngp, c.nv);
    std::vector<double> ci(ncp, 0.0);
    const std::vector<double> ci_rows = single_row_curvature_inv(c);
    for (int t = 0; t <= 8; ++t) v.bq.row_start[t] = ng;
    for (int rr = ng - 1; rr >= nz4; --rr)
        for (int t = 0; t < ext[order[rr]] && t <= 8; ++t) v.bq.row_start[t] = rr;
    v.bq.row_start[0] = nz4;
    for (int rr = 0; rr < ng; ++rr) {
        const int r = grow[order[rr]];
        ncols[rr] = rr < nz4 ? c.nv : std::min(c.nv, 16 * ext[order[rr]]);
        for (int j = 0; j < c.nv; ++j) {
            const double g = c.Gs(r, j);
            Grm[static_cast<size_t>(rr) * NVP + j] = g;
            Gcm[static_cast<size_t>(j) * ngp + rr] = g;
            double t = 0.0;
            for (int k = 0; k < c.nv; ++k) t += c.Gs(r, k) * c.Hinv(k, j);
            GH[static_cast<size_t>(rr) * NVP + j] = t;
            if (paired) {
                // the lower side is the negated functional (not the mirror row's own entries, which agree to 1e-13): every
                // pass sees the same row
                Gw[static_cast<size_t>(rr) * NVP + j] = g;
                Gw[static_cast<size_t>(rr + mir) * NVP + j] = -g;
                GH[static_cast<size_t>(rr + mir) * NVP + j] = -t;
            }
        }
        g0[rr] = c.g0s[r];
        ci[rr] = ci_rows[r];
        for (int j = 0; j < nx; ++j) Es[static_cast<size_t>(rr) * nx + j] = c.Es(r, j);
        if (paired) {
            const int q = c.mirror[r];
            g0[rr + mir] = c.g0s[q];
            ci[rr + mir] = ci_rows[q];
            for (int j = 0; j < nx; ++j) Es[static_cast<size_t>(rr + mir) * nx + j] = c.Es(q, j);
        }
    }
    int rc;
    if ((rc = upload_common(h, v, p, v.db, NVP))) return rc;
    v.db.nd = c.nc; v.db.ncc = 0; v.db.kc = 0; v.db.nks = 0;
    v.db.Gt = v.db.Hct = v.db.Psi = v.db.g0p = v.db.Esp = v.db.cip = nullptr;
    v.db.vmask = nullptr; v.db.row_of = nullptr;
    v.bq.ncp = ncp;
    v.bq.nz4 = nz4;
    v.bq.zx0 = c.off_x0 >= 0 ? c.off_x0 : 0;
    v.bq.znx = nx;
    v.bq.mir = mir; v.bq.ng = ng; v.bq.ngp = ngp;
    if ((rc = upload(h, v, Grm.data(), Grm.size(), &v.bq.Grm))) return rc;
    if ((rc = upload(h, v, Gcm.data(), Gcm.size(), &v.bq.Gcm))) return rc;
    if (paired) { if ((rc = upload(h, v, Gw.data(), Gw.size(), &v.bq.Gw))) return rc; }
    else v.bq.Gw = v.bq.Grm;
    if ((rc = upload(h, v, GH.data(), GH.size(), &v.bq.GHrm))) return rc;
    if ((rc = upload(h, v, g0.data(), g0.size(), &v.bq.g0))) return rc;
    if ((rc = upload(h, v, Es.data(), Es.size(), &v.bq.Es))) return rc;
    if ((rc = upload(h, v, ncols.data(), ncols.size(), &v.bq.ncols))) return rc;
    if ((rc = upload(h, v, ci.data(), ci.size(), &v.bq.ci))) return rc;
#ifdef TMPC_STAMPS
    {
        void *dbgp = nullptr;
        HIP_TRY(h, hipMalloc(&dbgp, 16 * sizeof(long long)));
        HIP_TRY(h, hipMemset(dbgp, 0, 16 * sizeof(long long)));
        v.dev.push_back(dbgp);
        v.db.dbg = static_cast<long long *>(dbgp);
    }
#endif
    {
        const tmpc::BlockArgs rec{v.db, v.bq};
        if ((rc = upload(h, v, &rec, 1, &v.bargs))) return rc;
    }
    return TMPC_OK;
}

int upload_variant(tmpc_handle *h, Variant &v, const tmpc_problem &p) {
    int rc;
    if ((rc = upload_wave(h, v, p))) return rc;
    if ((rc = upload_block(h, v, p))) return rc;
    if (!v.wave_ok && v.tiles == 0) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "condensed QP (nv=%d, rows=%d) is outside the compiled kernels (nv <= 128)", v.c.nv, v.c.nc);
        h->err = buf;
        return TMPC_E_UNSUPPORTED;
    }
    return TMPC_OK;
}

// block-kernel workspace: one slice per resident workgroup
int ensure_block_ws(tmpc_handle *h) {
    int ncp = 0, occ = 64;
    for (int k = 0; k < h->nvariants; ++k)
        if (h->v[k].tiles) { ncp = std::max(ncp, h->v[k].bq.ncp); occ = std::min(occ, tmpc::block_occupancy(h->v[k].tiles)); }
    if (ncp == 0 || h->blk_ws) return TMPC_OK;
    int occ_max = 1;
    for (int k = 0; k < h->nvariants; ++k)
        if (h->v[k].tiles) occ_max = std::max(occ_max, tmpc::block_occupancy(h->v[k].tiles));
    const int blocks = h->n_cu * occ_max;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->blk_ws),
                         static_cast<size_t>(blocks) * tmpc::block_workspace_rows() * ncp * sizeof(double)));
    h->blk_blocks = blocks;
    h->blk_ncp = ncp;
    return TMPC_OK;
}

bool use_block(const tmpc_handle *h, const Variant &v) {
    if (h->kernel_path == TMPC_PATH_BLOCK) return v.tiles != 0;
    if (h->kernel_path == TMPC_PATH_WAVE) return !v.wave_ok;
    return !v.wave_ok;
}

void free_staging(tmpc_handle *h) {
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    if (h->stage_pin) (void)hipHostFree(h->stage_pin);
    h->stage_dev = h->stage_pin = nullptr;
    h->d_x = h->d_r = h->d_u = h->d_x0 = h->d_ss = h->d_xn = nullptr;
    h->d_var = nullptr; h->d_st = h->d_it = nullptr;
    h->cap = 0;
}

// offsets and sub-buffers for a batch of B (tightly packed for THIS batch, whatever the capacity: the DMAs of a call move
// exactly its bytes); returns the total
size_t layout_staging(tmpc_handle *h, int64_t B) {
    const size_t nx = h->nx, nu = h->nu, N = h->N, b = static_cast<size_t>(B);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    // inputs
    h->off_r = up(b * nx * sizeof(double));
    h->off_var = h->off_r + up(b * nx * sizeof(double));
    h->stage_in_bytes = h->off_var + up(b);
    // outputs (x_nom last: it is optional and by far the largest)
    h->off_x0 = up(b * N * nu * sizeof(double));
    h->off_ss = h->off_x0 + up(b * nx * sizeof(double));
    h->off_st = h->off_ss + up(b * (nx + nu) * sizeof(double));
    h->off_it = h->off_st + up(b * sizeof(int32_t));
    h->stage_out_core = h->off_it + up(b * sizeof(int32_t));
    h->off_xn = h->stage_out_core;
    h->stage_out_bytes = h->off_xn + up(b * (N + 1) * nx * sizeof(double));
    if (h->stage_dev != nullptr) {
        char *in = h->stage_dev, *out = h->stage_dev + h->stage_in_bytes;
        h->d_x = reinterpret_cast<double *>(in);
        h->d_r = reinterpret_cast<double *>(in + h->off_r);
        h->d_var = reinterpret_cast<uint8_t *>(in + h->off_var);
        h->d_u = reinterpret_cast<double *>(out);
        h->d_x0 = reinterpret_cast<double *>(out + h->off_x0);
        h->d_ss = reinterpret_cast<double *>(out + h->off_ss);
        h->d_st = reinterpret_cast<int32_t *>(out + h->off_st);
        h->d_it = reinterpret_cast<int32_t *>(out + h->off_it);
        h->d_xn = reinterpret_cast<double *>(out + h->off_xn);
    }
    return h->stage_in_bytes + h->stage_out_bytes;
}

int ensure_staging(tmpc_handle *h, int64_t B) {
    if (B > h->cap) {
        if (h->stream) HIP_TRY(h, hipStreamSynchronize(h->stream));
        free_staging(h);
        const size_t total = layout_staging(h, B);
        HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->stage_dev), total));
        // the pinned mirror is a convenience, not a requirement: without it (or beyond 64 MB) the call copies from / to the
        // caller's buffers directly
        if (total <= (64u << 20)) {
            if (hipHostMalloc(reinterpret_cast<void **>(&h->stage_pin), total, hipHostMallocDefault) != hipSuccess) {
                h->stage_pin = nullptr;
                (void)hipGetLastError();
            }
        }
        h->cap = B;
    }
    (void)layout_staging(h, B);
    return TMPC_OK;
}

// The next pair of timing events of the handle's pool (tmpc_last_kernel_ms / tmpc_kernel_ms_total read them); records the first one.
// Beyond 4096 pairs the last one is reused.
int begin_timed_launch(tmpc_handle *h) {
    hipEvent_t e0 = h->pool.back().first, e1 = h->pool.back().second;
    if (h->pool_used < 4096) {
        if (h->pool_used == h->pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(h, hipEventCreate(&a));
            HIP_TRY(h, hipEventCreate(&b));
            h->pool.emplace_back(a, b);
        }
        e0 = h->pool[h->pool_used].first;
        e1 = h->pool[h->pool_used].second;
        ++h->pool_used;
    }
    h->ev0 = e0; h->ev1 = e1;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    return TMPC_OK;
}

int enqueue(tmpc_handle *h, int64_t B, const double *x_k, const double *ref, const uint8_t *variant, double *u_nom,
            double *x_nom0, double *xu_ss, double *x_nom, int32_t *status, int32_t *iters, int32_t *const *ws = nullptr,
            bool variants_valid = false) {
    { const int rce = begin_timed_launch(h); if (rce) return rce; }
    if (variant != nullptr && !variants_valid)      // (the closed loop's selector is its own gamma flags: always 0 or 1)
        HIP_TRY(h, tmpc::launch_mark_invalid_variants(variant, h->nvariants, B, h->nx, h->nu, h->N, u_nom, x_nom0, xu_ss, x_nom, status,
                                                      iters, h->stream));
    long long *ticks = nullptr;
    if (h->want_ticks) {
        if (B > h->ticks_cap) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (h->d_ticks) (void)hipFree(h->d_ticks);
            h->d_ticks = nullptr; h->ticks_cap = 0;
            HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->d_ticks), static_cast<size_t>(B) * sizeof(long long)));
            h->ticks_cap = B;
        }
        ticks = h->d_ticks;
        h->ticks_n = B;
        HIP_TRY(h, hipMemsetAsync(ticks, 0, static_cast<size_t>(B) * sizeof(long long), h->stream));
    }
    for (int k = 0; k < h->nvariants; ++k) {
        if (k == 1 && variant == nullptr) break;        // no per-instance selector: everything is variant 0
        Variant &v = h->v[k];
        v.d.ticks = ticks;
        v.db.ticks = ticks;
        if (use_block(h, v)) {
            int rcw = ensure_block_ws(h);
            if (rcw) return rcw;
            // the workspace slices are sized for the largest variant; a slice is addressed with this variant's ncp
            HIP_TRY(h, tmpc::launch_block(v.db, v.bq, v.bargs, v.tiles, h->blk_ws, h->blk_blocks, k, B, x_k, ref, variant, u_nom, x_nom0,
                                          xu_ss, x_nom, status, iters, &h->wc, h->stream));
            continue;
        }
        if (tmpc::parks_in_lds(v.shape)) {
            v.d.save = nullptr;
        } else {
            // one save slot per resident wave (at most 8 per CU), sized for this variant's row sides
            const size_t rs = static_cast<size_t>(2 * v.shape.dp + v.shape.ds + 2 * v.shape.cp + v.shape.cs);
            const size_t need = static_cast<size_t>(h->n_cu) * 8 * 2 * rs * 64 * sizeof(float);
            if (need > h->save_bytes) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                if (h->save_buf) (void)hipFree(h->save_buf);
                h->save_buf = nullptr; h->save_bytes = 0;
                HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->save_buf), need));
                h->save_bytes = need;
            }
            v.d.save = h->save_buf;
        }
        HIP_TRY(h, tmpc::launch_solve(v.d, v.shape, k, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status,
                                      iters, ws ? ws[k] : nullptr, ws ? ws[k] : nullptr, &h->wc, h->n_cu, h->stream));
    }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return TMPC_OK;
}

}  // namespace

// device memory of tmpc_lp_batch: one grow-only allocation per host thread, carved per call (the Gilbert-Tan recursion
// makes hundreds of small calls; ten hipMalloc / hipFree pairs each cost more than the kernel).  Lives until the
// process ends.
namespace {
struct LpArena {
    int device = -1;
    char *p = nullptr;
    size_t cap = 0, off = 0;
    hipError_t reserve(int dev, size_t bytes) {
        off = 0;
        if (dev == device && bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0; device = dev;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    template <class T> T *take(size_t count) {
        T *q = reinterpret_cast<T *>(p + off);
        off += (count * sizeof(T) + 255) / 256 * 256;
        return q;
    }
};
thread_local LpArena g_lp_arena;
size_t lp_round(size_t bytes) { return (bytes + 255) / 256 * 256; }
}  // namespace

extern "C" {

int tmpc_abi_version(void) { return TMPC_ABI_VERSION; }

const char *tmpc_last_error(const tmpc_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tmpc_create(const tmpc_problem *p, int device, tmpc_handle **out) {
    if (!p || !out) { g_create_error = "tmpc_create: NULL argument"; return TMPC_E_INVALID; }
    *out = nullptr;
    if (p->nx <= 0 || p->nu <= 0 || p->N <= 0 || p->nx > 16) {
        g_create_error = "tmpc_create: need 0 < nx <= 16, nu > 0, N > 0";
        return TMPC_E_INVALID;
    }
    tmpc_handle *h = new (std::nothrow) tmpc_handle();
    if (!h) { g_create_error = "out of memory"; return TMPC_E_NOMEM; }
    h->device = device; h->nx = p->nx; h->nu = p->nu; h->N = p->N;
    h->hA.assign(p->A ? p->A : nullptr, p->A ? p->A + p->nx * p->nx : nullptr);
    h->hB.assign(p->B ? p->B : nullptr, p->B ? p->B + p->nx * p->nu : nullptr);
    if (p->K) h->hK.assign(p->K, p->K + p->nu * p->nx);
    if (p->K_anc) h->hKanc.assign(p->K_anc, p->K_anc + p->nu * p->nx);
    h->nvariants = p->extended ? 2 : 1;
    int rc = TMPC_OK;
    try {
        for (int k = 0; k < h->nvariants && rc == TMPC_OK; ++k) {
            const std::string msg = tmpc::condense(*p, k, h->v[k].c);
            if (!msg.empty()) { h->err = "tmpc_create: " + msg; rc = TMPC_E_INVALID; }
        }
        if (rc == TMPC_OK && device >= 0) {
            hipError_t e = hipSetDevice(device);
            hipDeviceProp_t prop;
            if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
            if (e != hipSuccess) {
                h->err = std::string("tmpc_create: no usable HIP device: ") + hipGetErrorString(e);
                rc = TMPC_E_DEVICE;
            } else {
                h->n_cu = prop.multiProcessorCount;
                auto setup = [&]() -> int {
                    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
                    h->wc.size = 4096;
                    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->wc.ring), h->wc.size * sizeof(unsigned long long)));
                    HIP_TRY(h, hipMemset(h->wc.ring, 0, h->wc.size * sizeof(unsigned long long)));
                    for (int i = 0; i < 256; ++i) {       // timing events are created up front, not in the solve path
                        hipEvent_t a = nullptr, b = nullptr;
                        HIP_TRY(h, hipEventCreate(&a));
                        HIP_TRY(h, hipEventCreate(&b));
                        h->pool.emplace_back(a, b);
                    }
                    for (int k = 0; k < h->nvariants; ++k) {
                        int r2 = upload_variant(h, h->v[k], *p);
                        if (r2) return r2;
                    }
                    return TMPC_OK;
                };
                rc = setup();
            }
        }
        if (rc == TMPC_OK && device < 0) {
            // host-only handle: the layouts are laid out for the debug dumps only.  A problem no kernel covers (nv > 128) still
            // gets its handle -- tmpc_get_condensed and the oracle-side tests use it -- and the dump calls answer UNSUPPORTED
            // (wave_ok = false, tiles = 0).
            for (int k = 0; k < h->nvariants && rc == TMPC_OK; ++k) {
                rc = upload_variant(h, h->v[k], *p);
                if (rc == TMPC_E_UNSUPPORTED) { rc = TMPC_OK; h->err.clear(); }
            }
        }
    } catch (const std::exception &ex) {
        h->err = std::string("tmpc_create: ") + ex.what();
        rc = TMPC_E_NOMEM;
    }
    if (rc != TMPC_OK) {
        g_create_error = h->err;
        tmpc_destroy(h);
        return rc;
    }
    *out = h;
    return TMPC_OK;
}

void tmpc_destroy(tmpc_handle *h) {
    if (!h) return;
    if (h->device < 0) {
        for (int k = 0; k < 2; ++k)
            for (void *p : h->v[k].dev) std::free(p);
        delete h;
        return;
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_staging(h);
    {
        void *wsp[] = {h->blk_ws, h->mc_arena, h->wc.ring, h->d_ticks, h->save_buf};
        for (void *q2 : wsp) if (q2) (void)hipFree(q2);
    }
    for (int k = 0; k < 2; ++k)
        for (void *p : h->v[k].dev) (void)hipFree(p);
    for (auto &pr : h->pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int tmpc_solve_batch_device(tmpc_handle *h, int64_t B, const double *x_k, const double *ref, const uint8_t *variant,
                            double *u_nom, double *x_nom0, double *xu_ss, double *x_nom, int32_t *status, int32_t *iters) {
    if (!h) return TMPC_E_INVALID;
    if (B < 0 || !x_k || !ref || !u_nom || !status || !iters) { h->err = "tmpc_solve_batch_device: NULL argument"; return TMPC_E_INVALID; }
    if (B == 0) return TMPC_OK;
    if (h->device < 0) { h->err = "host-only handle (device < 0): nothing can be solved without the GPU"; return TMPC_E_DEVICE; }
    HIP_TRY(h, hipSetDevice(h->device));
    return enqueue(h, B, x_k, ref, variant, u_nom, x_nom0, xu_ss, x_nom, status, iters);
}

int tmpc_solve_batch(tmpc_handle *h, int64_t B, const double *x_k, const double *ref, const uint8_t *variant,
                     double *u_nom, double *x_nom0, double *xu_ss, double *x_nom, int32_t *status, int32_t *iters) {
    if (!h) return TMPC_E_INVALID;
    if (B < 0 || !x_k || !ref || !u_nom || !status || !iters) { h->err = "tmpc_solve_batch: NULL argument"; return TMPC_E_INVALID; }
    if (B == 0) return TMPC_OK;
    if (variant)
        for (int64_t i = 0; i < B; ++i)
            if (variant[i] >= h->nvariants) { h->err = "tmpc_solve_batch: variant id out of range"; return TMPC_E_INVALID; }
    if (h->device < 0) { h->err = "host-only handle (device < 0): nothing can be solved without the GPU"; return TMPC_E_DEVICE; }
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_staging(h, B);
    if (rc) return rc;
    const size_t nx = h->nx, nu = h->nu, N = h->N, b = static_cast<size_t>(B);
    if (h->stage_pin != nullptr) {
        // through the pinned mirror: one DMA in, one out
        char *pin_in = h->stage_pin, *pin_out = h->stage_pin + h->stage_in_bytes;
        std::memcpy(pin_in, x_k, b * nx * sizeof(double));
        std::memcpy(pin_in + h->off_r, ref, b * nx * sizeof(double));
        size_t in_bytes = h->off_r + b * nx * sizeof(double);
        if (variant) { std::memcpy(pin_in + h->off_var, variant, b); in_bytes = h->off_var + b; }
        HIP_TRY(h, hipMemcpyAsync(h->stage_dev, pin_in, in_bytes, hipMemcpyHostToDevice, h->stream));
        rc = enqueue(h, B, h->d_x, h->d_r, variant ? h->d_var : nullptr, h->d_u, h->d_x0, h->d_ss, x_nom ? h->d_xn : nullptr,
                     h->d_st, h->d_it);
        if (rc) return rc;
        const size_t out_bytes = x_nom ? h->off_xn + b * (N + 1) * nx * sizeof(double) : h->off_it + b * sizeof(int32_t);
        HIP_TRY(h, hipMemcpyAsync(pin_out, h->stage_dev + h->stage_in_bytes, out_bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(u_nom, pin_out, b * N * nu * sizeof(double));
        if (x_nom0) std::memcpy(x_nom0, pin_out + h->off_x0, b * nx * sizeof(double));
        if (xu_ss) std::memcpy(xu_ss, pin_out + h->off_ss, b * (nx + nu) * sizeof(double));
        std::memcpy(status, pin_out + h->off_st, b * sizeof(int32_t));
        std::memcpy(iters, pin_out + h->off_it, b * sizeof(int32_t));
        if (x_nom) std::memcpy(x_nom, pin_out + h->off_xn, b * (N + 1) * nx * sizeof(double));
        return TMPC_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_x, x_k, b * nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_r, ref, b * nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (variant) HIP_TRY(h, hipMemcpyAsync(h->d_var, variant, b, hipMemcpyHostToDevice, h->stream));
    rc = enqueue(h, B, h->d_x, h->d_r, variant ? h->d_var : nullptr, h->d_u, h->d_x0, h->d_ss, x_nom ? h->d_xn : nullptr,
                 h->d_st, h->d_it);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(u_nom, h->d_u, b * N * nu * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (x_nom0) HIP_TRY(h, hipMemcpyAsync(x_nom0, h->d_x0, b * nx * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (xu_ss) HIP_TRY(h, hipMemcpyAsync(xu_ss, h->d_ss, b * (nx + nu) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (x_nom) HIP_TRY(h, hipMemcpyAsync(x_nom, h->d_xn, b * (N + 1) * nx * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(status, h->d_st, b * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(iters, h->d_it, b * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_set_kernel_path(tmpc_handle *h, int path) {
    if (!h) return TMPC_E_INVALID;
    if (path != TMPC_PATH_AUTO && path != TMPC_PATH_WAVE && path != TMPC_PATH_BLOCK) { h->err = "tmpc_set_kernel_path: unknown path"; return TMPC_E_INVALID; }
    for (int k = 0; k < h->nvariants; ++k) {
        if (path == TMPC_PATH_WAVE && !h->v[k].wave_ok && h->device >= 0) { h->err = "tmpc_set_kernel_path: no wave-per-QP shape covers this problem"; return TMPC_E_UNSUPPORTED; }
        if (path == TMPC_PATH_BLOCK && h->v[k].tiles == 0 && h->device >= 0) { h->err = "tmpc_set_kernel_path: the block kernel needs nv <= 128"; return TMPC_E_UNSUPPORTED; }
    }
    h->kernel_path = path;
    return TMPC_OK;
}

int tmpc_get_kernel_path(const tmpc_handle *h, int variant) {
    if (!h || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    return use_block(h, h->v[variant]) ? TMPC_PATH_BLOCK : TMPC_PATH_WAVE;       // (host-only handles included: the choice is made at tmpc_create)
}

int tmpc_debug_dump_layout(const tmpc_handle *h, int variant, const char *path) {
    if (!h || !path || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    if (h->device >= 0) return TMPC_E_UNSUPPORTED;            // the arrays of a device handle live in HBM
    const Variant &v = h->v[variant];
    if (!v.wave_ok) return TMPC_E_UNSUPPORTED;
    FILE *f = std::fopen(path, "wb");
    if (!f) return TMPC_E_INVALID;
    const int32_t shp[6] = {v.shape.nvp, v.shape.dp, v.shape.ds, v.shape.kcp, v.shape.cp, v.shape.cs};
    const uint64_t qp_bytes = sizeof(tmpc::DeviceQP);
    const int32_t tag[2] = {tmpc::DUMP_TAG, tmpc::DUMP_FORMAT};
    std::fwrite(tag, 4, 2, f);
    std::fwrite(shp, 4, 6, f);
    std::fwrite(&qp_bytes, 8, 1, f);
    std::fwrite(&v.d, sizeof(tmpc::DeviceQP), 1, f);
    // every array the structure points to, in the order of its fields: byte count, bytes (0: null pointer)
    const void *ptrs[] = {v.d.Gt, v.d.Hct, v.d.Psi, v.d.Hs, v.d.Hinv, v.d.F1s, v.d.F2s, v.d.g0p, v.d.Esp, v.d.vmask, v.d.row_of,
                          v.d.gp0, v.d.Ep, v.d.Dv, v.d.Tzs, v.d.Txf, v.d.Mth, v.d.A, v.d.B, v.d.cip};
    for (const void *q : ptrs) {
        uint64_t n = 0;
        if (q)
            for (size_t i = 0; i < v.dev.size(); ++i)
                if (v.dev[i] == q) { n = v.dev_bytes[i]; break; }
        std::fwrite(&n, 8, 1, f);
        if (n) std::fwrite(q, 1, n, f);
    }
    std::fclose(f);
    return TMPC_OK;
}

int tmpc_debug_dump_block_layout(const tmpc_handle *h, int variant, const char *path) {
    if (!h || !path || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    if (h->device >= 0) return TMPC_E_UNSUPPORTED;
    const Variant &v = h->v[variant];
    if (v.tiles == 0) return TMPC_E_UNSUPPORTED;
    FILE *f = std::fopen(path, "wb");
    if (!f) return TMPC_E_INVALID;
    const int32_t hd[2] = {v.tiles, tmpc::block_workspace_rows()};
    const uint64_t sz[2] = {sizeof(tmpc::DeviceQP), sizeof(tmpc::BlockQP)};
    const int32_t tag[2] = {tmpc::DUMP_TAG, tmpc::DUMP_FORMAT};
    std::fwrite(tag, 4, 2, f);
    std::fwrite(hd, 4, 2, f);
    std::fwrite(sz, 8, 2, f);
    std::fwrite(&v.db, sizeof(tmpc::DeviceQP), 1, f);
    std::fwrite(&v.bq, sizeof(tmpc::BlockQP), 1, f);
    const void *ptrs[] = {v.db.Hs, v.db.Hinv, v.db.F1s, v.db.F2s, v.db.gp0, v.db.Ep, v.db.Dv, v.db.Tzs, v.db.Txf, v.db.Mth, v.db.A, v.db.B,
                          v.bq.Grm, v.bq.Gcm, v.bq.GHrm, v.bq.g0, v.bq.Es, v.bq.ncols, v.bq.Gw == v.bq.Grm ? nullptr : v.bq.Gw, v.bq.ci};
    for (const void *q : ptrs) {
        uint64_t n = 0;
        if (q)
            for (size_t i = 0; i < v.dev.size(); ++i)
                if (v.dev[i] == q) { n = v.dev_bytes[i]; break; }
        std::fwrite(&n, 8, 1, f);
        if (n) std::fwrite(q, 1, n, f);
    }
    std::fclose(f);
    return TMPC_OK;
}

const char *tmpc_kernel_name(const tmpc_handle *h, int variant) {
    if (!h || variant < 0 || variant >= h->nvariants) return "";
    const Variant &v = h->v[variant];
    if (use_block(h, v)) {
        static const char *names[] = {"", "tmpc::solve_block_kernel<1>", "tmpc::solve_block_kernel<2>", "", "tmpc::solve_block_kernel<4>",
                                      "", "", "", "tmpc::solve_block_kernel<8>"};
        return (v.tiles >= 0 && v.tiles <= 8) ? names[v.tiles] : "";
    }
    return tmpc::kernel_name(v.shape);
}

int tmpc_mc_set_actuator(tmpc_handle *h, int kind) {
    if (!h) return TMPC_E_INVALID;
    if (kind != TMPC_ACTUATOR_CONSISTENT && kind != TMPC_ACTUATOR_SMART) { h->err = "tmpc_mc_set_actuator: unknown actuator"; return TMPC_E_INVALID; }
    h->actuator = kind;
    return TMPC_OK;
}

int tmpc_mc_set_plant(tmpc_handle *h, int kind, const double *par7, int substeps) {
    if (!h) return TMPC_E_INVALID;
    if (kind == TMPC_PLANT_LINEAR) { h->plant = kind; return TMPC_OK; }
    if (kind != TMPC_PLANT_CARTPOLE || !par7 || substeps < 1) { h->err = "tmpc_mc_set_plant: unknown plant or missing parameters"; return TMPC_E_INVALID; }
    if (h->nx != 4 || h->nu != 1) { h->err = "tmpc_mc_set_plant: the cart-pole plant needs nx = 4, nu = 1"; return TMPC_E_INVALID; }
    for (int i = 0; i < 7; ++i) h->plant_par[i] = par7[i];
    h->plant = kind;
    h->plant_substeps = substeps;
    return TMPC_OK;
}

int tmpc_mc_set_capture(tmpc_handle *h, int64_t index) {
    if (!h) return TMPC_E_INVALID;
    h->mc_capture = index < 0 ? -1 : index;
    return TMPC_OK;
}

int tmpc_mc_get_capture(tmpc_handle *h, int32_t T, double *x_traj, double *x_nom_traj, double *u_traj) {
    if (!h) return TMPC_E_INVALID;
    if (!h->mc_cap_dev || T != h->mc_cap_T) { h->err = "tmpc_mc_get_capture: no trajectory of this length was recorded by the last tmpc_mc_run"; return TMPC_E_INVALID; }
    const size_t nx = h->nx, nu = h->nu, w = 2 * nx + nu;
    std::vector<double> buf(static_cast<size_t>(T) * w);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(buf.data(), h->mc_cap_dev, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t) {
        for (size_t i = 0; i < nx; ++i) {
            if (x_traj) x_traj[t * nx + i] = buf[t * w + i];
            if (x_nom_traj) x_nom_traj[t * nx + i] = buf[t * w + nx + i];
        }
        for (size_t j = 0; j < nu; ++j) if (u_traj) u_traj[t * nu + j] = buf[t * w + 2 * nx + j];
    }
    return TMPC_OK;
}

int tmpc_set_solve_timing(tmpc_handle *h, int on) {
    if (!h) return TMPC_E_INVALID;
    h->want_ticks = on ? 1 : 0;
    if (!on) h->ticks_n = 0;
    return TMPC_OK;
}

int tmpc_get_solve_ticks(tmpc_handle *h, int64_t B, int64_t *ticks) {
    if (!h || !ticks) return TMPC_E_INVALID;
    if (!h->want_ticks || B != h->ticks_n || !h->d_ticks) { h->err = "tmpc_get_solve_ticks: no solve of this batch size was timed (tmpc_set_solve_timing)"; return TMPC_E_INVALID; }
    static_assert(sizeof(long long) == sizeof(int64_t), "tick counts are 64-bit");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(ticks, h->d_ticks, static_cast<size_t>(B) * sizeof(int64_t), hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_mc_get_solve_ticks(tmpc_handle *h, int64_t B, int64_t *ticks_sum, int64_t *ticks_max) {
    if (!h) return TMPC_E_INVALID;
    if (!h->mc_tick_sum || B != h->mc_tick_B) { h->err = "tmpc_mc_get_solve_ticks: the last tmpc_mc_run was not timed (tmpc_set_solve_timing) or had another batch size"; return TMPC_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (ticks_sum) HIP_TRY(h, hipMemcpy(ticks_sum, h->mc_tick_sum, static_cast<size_t>(B) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (ticks_max) HIP_TRY(h, hipMemcpy(ticks_max, h->mc_tick_max, static_cast<size_t>(B) * sizeof(int64_t), hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_mc_set_device_rng(tmpc_handle *h, int on, uint64_t seed, int64_t first_trajectory, const double *w_bound) {
    if (!h) return TMPC_E_INVALID;
    h->mc_rng_on = on ? 1 : 0;
    h->mc_rng_seed = seed;
    h->mc_rng_first = first_trajectory;
    h->mc_w_bound.assign(static_cast<size_t>(h->nx), 0.0);
    if (on && w_bound)
        for (int i = 0; i < h->nx; ++i) h->mc_w_bound[i] = w_bound[i];
    return TMPC_OK;
}

int tmpc_mc_get_physics_error(tmpc_handle *h, int64_t B, double *err2_phys) {
    if (!h || !err2_phys) return TMPC_E_INVALID;
    if (!h->mc_err2_phys || B != h->mc_phys_B) { h->err = "tmpc_mc_get_physics_error: the last tmpc_mc_run had the linear plant or another batch size"; return TMPC_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(err2_phys, h->mc_err2_phys, static_cast<size_t>(B) * sizeof(double), hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_mc_set_warm_start(tmpc_handle *h, int on) {
    if (!h) return TMPC_E_INVALID;
    h->mc_warm = on ? 1 : 0;
    return TMPC_OK;
}

int tmpc_mc_set_fused(tmpc_handle *h, int mode) {
    if (!h) return TMPC_E_INVALID;
    if (mode != TMPC_MC_FUSED_OFF && mode != TMPC_MC_FUSED_ON && mode != TMPC_MC_FUSED_AUTO) { h->err = "tmpc_mc_set_fused: mode is TMPC_MC_FUSED_OFF / _ON / _AUTO"; return TMPC_E_INVALID; }
    h->mc_fused = mode;
    return TMPC_OK;
}

int tmpc_mc_last_fused(const tmpc_handle *h) { return h ? h->mc_last_fused : 0; }

}  // extern "C"

namespace {
// packets injected by the caller (tmpc_mc_replay) and the per-step record that goes back
struct McReplay {
    const double *U, *xn0;      // host: [B][T][N+1][nu], [B][T][nx] (xn0 may be NULL unless extended)
    double *trace_f;            // host: [B][T][3 nx + nu]
    int32_t *trace_i;           // host: [B][T][3]
};

int mc_run_impl(tmpc_handle *h, int64_t B, int32_t T, int extended, const double *p_loss, const double *ref,
                const double *th_u, const double *ga_u, const double *w, const double *x0, const double *HZ, const double *hZ,
                int32_t rZ, double *err2, int32_t *tube_viol, int32_t *not_optimal, double *x_final, double *consistent,
                int32_t *iters_sum, const McReplay *rp) {
    if (!h) return TMPC_E_INVALID;
    const bool host_draws = rp != nullptr || !h->mc_rng_on;
    if (B < 0 || T < 0 || !p_loss || !ref || (host_draws && (!th_u || !ga_u || !w)) || (rZ > 0 && (!HZ || !hZ))) { h->err = "tmpc_mc_run: NULL argument"; return TMPC_E_INVALID; }
    if (h->device < 0) { h->err = "host-only handle (device < 0): nothing can be solved without the GPU"; return TMPC_E_DEVICE; }
    if (extended && h->nvariants < 2) { h->err = "tmpc_mc_run: extended loop needs a problem created with extended = 1"; return TMPC_E_INVALID; }
    if (h->hK.empty() || h->hKanc.empty()) { h->err = "tmpc_mc_run: the problem description carries no gains K / K_anc"; return TMPC_E_INVALID; }
    if (h->nu > 16) { h->err = "tmpc_mc_run: nu <= 16"; return TMPC_E_UNSUPPORTED; }
    if (B == 0 || T == 0) return TMPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_staging(h, B);
    if (rc) return rc;
    const size_t nx = h->nx, nu = h->nu, N = h->N, b = static_cast<size_t>(B), t_ = static_cast<size_t>(T);
    // upper bound of what the carve-outs below need (each rounded up to 256 B)
    const size_t need = 256 * 54 + sizeof(tmpc::McFused) + 8 * t_ + b + 8 * (4 * nx * nx + 4 * nu * nx + static_cast<size_t>(rZ) * (nx + 1) + b * (2 + (host_draws ? 2 * t_ + t_ * nx : 0)) + nx) +
                        8 * b * (6 * nx + (N + 1) * nu + nu + 5) + 4 * b * 8 + 2 * b + 2 * 4 * b * tmpc::WS_STRIDE + 8 * t_ * (2 * nx + nu) +
                        (rp ? b * t_ * (8 * ((N + 1) * nu + nx) + 8 * (3 * nx + nu) + 4 * 3) : 0);
    // what the previous run left in the arena is gone from here on, whether this run gets as far as replacing it or not
    // (tmpc_mc_get_capture / _solve_ticks / _physics_error must not read a freed or half-written arena)
    h->mc_cap_dev = nullptr; h->mc_cap_T = 0;
    h->mc_err2_phys = nullptr; h->mc_phys_B = 0;
    h->mc_tick_sum = h->mc_tick_max = nullptr; h->mc_tick_B = 0;
    if (need > h->mc_arena_bytes) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->mc_arena) (void)hipFree(h->mc_arena);
        h->mc_arena = nullptr;
        h->mc_arena_bytes = 0;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->mc_arena), need));
        h->mc_arena_bytes = need;
    }
    size_t arena_off = 0;
    auto dalloc = [&](size_t bytes, void **out) -> int {
        const size_t sz = ((bytes ? bytes : 8) + 255) / 256 * 256;
        if (arena_off + sz > h->mc_arena_bytes) { h->err = "tmpc_mc_run: internal arena too small"; return TMPC_E_NOMEM; }
        *out = h->mc_arena + arena_off;
        arena_off += sz;
        return TMPC_OK;
    };
    auto up = [&](const void *src, size_t bytes, const void **out) -> int {
        void *q2 = nullptr;
        int r2 = dalloc(bytes, &q2);
        if (r2) return r2;
        HIP_TRY(h, hipMemcpyAsync(q2, src, bytes, hipMemcpyHostToDevice, h->stream));
        *out = q2;
        return TMPC_OK;
    };
    auto run = [&]() -> int {
        tmpc::McModel m{};
        tmpc::McState st{};
        m.nx = h->nx; m.nu = h->nu; m.N = h->N; m.extended = extended ? 1 : 0; m.rZ = rZ;
        m.plant = h->plant; m.substeps = h->plant_substeps; m.smart = h->actuator == TMPC_ACTUATOR_SMART ? 1 : 0;
        for (int i = 0; i < 7; ++i) m.par[i] = h->plant_par[i];
        int r2;
        if ((r2 = up(h->hA.data(), nx * nx * 8, reinterpret_cast<const void **>(&m.A)))) return r2;
        if ((r2 = up(h->hB.data(), nx * nu * 8, reinterpret_cast<const void **>(&m.B)))) return r2;
        if ((r2 = up(h->hK.data(), nu * nx * 8, reinterpret_cast<const void **>(&m.K)))) return r2;
        if ((r2 = up(h->hKanc.data(), nu * nx * 8, reinterpret_cast<const void **>(&m.K_anc)))) return r2;
        if ((r2 = up(HZ, static_cast<size_t>(rZ) * nx * 8, reinterpret_cast<const void **>(&m.HZ)))) return r2;
        if ((r2 = up(hZ, static_cast<size_t>(rZ) * 8, reinterpret_cast<const void **>(&m.hZ)))) return r2;
        if ((r2 = up(p_loss, b * 8, reinterpret_cast<const void **>(&st.p_loss)))) return r2;
        if (host_draws) {
            if ((r2 = up(th_u, b * t_ * 8, reinterpret_cast<const void **>(&st.th_u)))) return r2;
            if ((r2 = up(ga_u, b * t_ * 8, reinterpret_cast<const void **>(&st.ga_u)))) return r2;
            if ((r2 = up(w, b * t_ * nx * 8, reinterpret_cast<const void **>(&st.w)))) return r2;
        } else {
            st.rng_on = 1;
            st.rng_seed = h->mc_rng_seed;
            st.rng_first = h->mc_rng_first;
            if ((r2 = up(h->mc_w_bound.data(), nx * 8, reinterpret_cast<const void **>(&st.w_bound)))) return r2;
        }
        struct { void **p; size_t bytes; int fill; } arrays[] = {
            {reinterpret_cast<void **>(&st.x), b * nx * 8, 0}, {reinterpret_cast<void **>(&st.x_hat), b * nx * 8, 0},
            {reinterpret_cast<void **>(&st.x_nom), b * nx * 8, 0}, {reinterpret_cast<void **>(&st.Ubuf), b * (N + 1) * nu * 8, 0},
            {reinterpret_cast<void **>(&st.u_latest0), b * nu * 8, 0}, {reinterpret_cast<void **>(&st.x_nom0_latest), b * nx * 8, 0},
            {reinterpret_cast<void **>(&st.ref_k), b * nx * 8, 0},
            {reinterpret_cast<void **>(&st.err2), b * 8, 0},
            {reinterpret_cast<void **>(&st.consistent), b * 8, 0}, {reinterpret_cast<void **>(&st.q_est), b * 4, 0},
            {reinterpret_cast<void **>(&st.q_act), b * 4, 0}, {reinterpret_cast<void **>(&st.s), b * 4, 0},
            {reinterpret_cast<void **>(&st.Theta), b * 4, 0}, {reinterpret_cast<void **>(&st.last_lost), b * 4, 0xFF},
            {reinterpret_cast<void **>(&st.tube_viol), b * 4, 0}, {reinterpret_cast<void **>(&st.not_optimal), b * 4, 0},
            {reinterpret_cast<void **>(&st.iters_sum), b * 4, 0},
            {reinterpret_cast<void **>(&st.gamma), b, 1}, {reinterpret_cast<void **>(&st.dead), b, 0}};
        for (auto &a : arrays) {
            if ((r2 = dalloc(a.bytes, a.p))) return r2;
            HIP_TRY(h, hipMemsetAsync(*a.p, a.fill, a.bytes, h->stream));        // 0xFF bytes = -1 for last_lost; gamma = 1
        }
        if (x0) {
            HIP_TRY(h, hipMemcpyAsync(st.x, x0, b * nx * 8, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(st.x_hat, x0, b * nx * 8, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(st.x_nom, x0, b * nx * 8, hipMemcpyHostToDevice, h->stream));
        }
        st.cap_index = -1;
        st.cap = nullptr;
        h->mc_cap_dev = nullptr;
        if (h->mc_capture >= 0 && h->mc_capture < B) {
            if ((r2 = dalloc(t_ * (2 * nx + nu) * 8, reinterpret_cast<void **>(&st.cap)))) return r2;
            HIP_TRY(h, hipMemsetAsync(st.cap, 0, t_ * (2 * nx + nu) * 8, h->stream));
            st.cap_index = h->mc_capture;
            h->mc_cap_dev = st.cap;
            h->mc_cap_T = T;
        }
        h->mc_err2_phys = nullptr;
        h->mc_phys_B = 0;
        if (m.plant != TMPC_PLANT_LINEAR) {
            if ((r2 = dalloc(b * 8, reinterpret_cast<void **>(&st.err2_phys)))) return r2;
            HIP_TRY(h, hipMemsetAsync(st.err2_phys, 0, b * 8, h->stream));
            h->mc_err2_phys = st.err2_phys; h->mc_phys_B = B;
        }
        h->mc_tick_sum = h->mc_tick_max = nullptr;
        h->mc_tick_B = 0;
        if (h->want_ticks) {
            if ((r2 = dalloc(b * 8, reinterpret_cast<void **>(&st.tick_sum)))) return r2;
            if ((r2 = dalloc(b * 8, reinterpret_cast<void **>(&st.tick_max)))) return r2;
            HIP_TRY(h, hipMemsetAsync(st.tick_sum, 0, b * 8, h->stream));
            HIP_TRY(h, hipMemsetAsync(st.tick_max, 0, b * 8, h->stream));
            h->mc_tick_sum = st.tick_sum; h->mc_tick_max = st.tick_max; h->mc_tick_B = B;
        }
        // warm start: one working-set record per trajectory and variant (row ids are per variant), updated in place by the
        // solve kernel; m = 0 (the memset) means "nothing to start from"
        int32_t *ws[2] = {nullptr, nullptr};
        if (h->mc_warm) {
            for (int k = 0; k < (extended ? 2 : 1); ++k) {
                if ((r2 = dalloc(b * tmpc::WS_STRIDE * 4, reinterpret_cast<void **>(&ws[k])))) return r2;
                HIP_TRY(h, hipMemsetAsync(ws[k], 0, b * tmpc::WS_STRIDE * 4, h->stream));
            }
        }
        if (rp) {
            if ((r2 = up(rp->U, b * t_ * (N + 1) * nu * 8, reinterpret_cast<const void **>(&st.rp_U)))) return r2;
            if (rp->xn0) { if ((r2 = up(rp->xn0, b * t_ * nx * 8, reinterpret_cast<const void **>(&st.rp_xn0)))) return r2; }
            else {
                // (plain controller: the packets carry no x_nom_0; the state machines then never read it -- zeros)
                void *z = nullptr;
                if ((r2 = dalloc(b * t_ * nx * 8, &z))) return r2;
                HIP_TRY(h, hipMemsetAsync(z, 0, b * t_ * nx * 8, h->stream));
                st.rp_xn0 = static_cast<const double *>(z);
            }
            if ((r2 = dalloc(b * t_ * (3 * nx + nu) * 8, reinterpret_cast<void **>(&st.trace_f)))) return r2;
            if ((r2 = dalloc(b * t_ * 3 * 4, reinterpret_cast<void **>(&st.trace_i)))) return r2;
            HIP_TRY(h, hipMemsetAsync(st.trace_f, 0, b * t_ * (3 * nx + nu) * 8, h->stream));
            HIP_TRY(h, hipMemsetAsync(st.trace_i, 0, b * t_ * 3 * 4, h->stream));
        }
        HIP_TRY(h, tmpc::launch_mc_pre(m, st, B, ref[0], h->stream));
        // ONE launch for the whole sweep where the controller has one problem and it runs on the wave kernel: a wave keeps its
        // trajectory for all T steps, solve and state machines alternating inside the kernel (tmpc_fused.hip).  The work item of
        // that launch is a trajectory, T solves long: with B a little above a multiple of the resident waves the last round
        // of trajectories would run on a nearly empty card, so TMPC_MC_FUSED_AUTO fuses when the rounds are at least 85 % full
        // (or there is a single round) and otherwise keeps the launch per time step, whose work item is one solve.
        h->mc_last_fused = 0;
        bool fuse = !rp && !extended && h->mc_fused != TMPC_MC_FUSED_OFF && !use_block(h, h->v[0]);
        if (fuse && h->mc_fused == TMPC_MC_FUSED_AUTO) {
            const int64_t slots = tmpc::resident_waves(h->v[0].shape, h->n_cu);
            const int64_t rounds = slots > 0 ? (B + slots - 1) / slots : 0;
            fuse = rounds == 1 || (rounds > 0 && static_cast<double>(B) >= 0.85 * static_cast<double>(rounds * slots));
        }
        // the extended controller (two problems, two kernel shapes): per time step ONE launch per problem with the state machines of
        // its trajectories inside (closed_loop_step_kernel) -- two launches per step where the plain per-step loop has three
        // (TMPC_MC_FUSED_AUTO: from one round of resident waves on -- below that a step is the latency of its launches, and the state
        // machines inside BOTH of them lengthen it: 200 trajectories at N = 20 0.0345 s with three launches per step, 0.0367 s with two)
        bool step_fuse = !rp && extended && h->mc_fused != TMPC_MC_FUSED_OFF && !use_block(h, h->v[0]) && !use_block(h, h->v[1]);
        if (step_fuse && h->mc_fused == TMPC_MC_FUSED_AUTO) step_fuse = B >= tmpc::resident_waves(h->v[1].shape, h->n_cu);
        // per-solve tick buffer and the hand-over save slots of the wave kernel, as enqueue() provides them per call
        auto prepare_wave = [&](int nvar) -> int {
            long long *ticks = nullptr;
            if (h->want_ticks) {
                if (B > h->ticks_cap) {
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                    if (h->d_ticks) (void)hipFree(h->d_ticks);
                    h->d_ticks = nullptr; h->ticks_cap = 0;
                    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->d_ticks), static_cast<size_t>(B) * sizeof(long long)));
                    h->ticks_cap = B;
                }
                ticks = h->d_ticks;
                h->ticks_n = B;
                HIP_TRY(h, hipMemsetAsync(ticks, 0, static_cast<size_t>(B) * sizeof(long long), h->stream));
            }
            st.ticks = ticks;
            // (a slice of the save buffer per problem)
            size_t need_save = 0, off_save[2] = {0, 0};
            for (int k = 0; k < nvar; ++k) {
                Variant &v = h->v[k];
                v.d.ticks = ticks;
                off_save[k] = need_save;
                if (!tmpc::parks_in_lds(v.shape)) {
                    const size_t rs = static_cast<size_t>(2 * v.shape.dp + v.shape.ds + 2 * v.shape.cp + v.shape.cs);
                    need_save += static_cast<size_t>(h->n_cu) * 8 * 2 * rs * 64 * sizeof(float);
                }
            }
            if (need_save > h->save_bytes) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                if (h->save_buf) (void)hipFree(h->save_buf);
                h->save_buf = nullptr; h->save_bytes = 0;
                HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->save_buf), need_save));
                h->save_bytes = need_save;
            }
            for (int k = 0; k < nvar; ++k)
                h->v[k].d.save = tmpc::parks_in_lds(h->v[k].shape) ? nullptr : h->save_buf + off_save[k] / sizeof(float);
            return TMPC_OK;
        };
        if (step_fuse) {
            tmpc::McFused mf{};
            if ((r2 = up(ref, t_ * 8, reinterpret_cast<const void **>(&mf.ref_seq)))) return r2;
            if ((r2 = prepare_wave(2))) return r2;
            uint8_t *gam[2] = {st.gamma, nullptr};          // selector read in a step / arrival flags written in it: swapped every step
            if ((r2 = dalloc(b, reinterpret_cast<void **>(&gam[1])))) return r2;
            HIP_TRY(h, hipMemsetAsync(gam[1], 1, b, h->stream));
            mf.m = m; mf.st = st; mf.T = T;
            const tmpc::McFused *d_mf = nullptr;
            if ((r2 = up(&mf, sizeof(mf), reinterpret_cast<const void **>(&d_mf)))) return r2;
            // (Measured and dropped: the two launches of a step on two streams, so that the second one's workgroups start on the CUs the
            // first one's tail leaves idle -- config 4 extended 0.27 -> 0.28 s: the fork / join events of every step cost more.)
            for (int t = 0; t < T; ++t) {
                if ((r2 = begin_timed_launch(h))) return r2;
                for (int k = 0; k < 2; ++k)
                    HIP_TRY(h, tmpc::launch_solve_mc_step(h->v[k].d, h->v[k].shape, k, B, gam[t & 1], h->d_u, h->d_x0, h->d_ss, h->d_st, h->d_it, ws[k],
                                                          d_mf, t, gam[(t + 1) & 1], &h->wc, h->n_cu, h->stream));
                HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
            }
            h->timed = true;
            h->mc_last_fused = 2;
            fuse = true;            // (the per-step loop below is skipped)
        } else if (fuse) {
            tmpc::McFused mf{};
            if ((r2 = up(ref, t_ * 8, reinterpret_cast<const void **>(&mf.ref_seq)))) return r2;
            Variant &v = h->v[0];
            if ((r2 = prepare_wave(1))) return r2;
            mf.m = m; mf.st = st; mf.T = T;
            const tmpc::McFused *d_mf = nullptr;            // the record itself lives in the arena: the kernel reads it field by field
            if ((r2 = up(&mf, sizeof(mf), reinterpret_cast<const void **>(&d_mf)))) return r2;
            if ((r2 = begin_timed_launch(h))) return r2;          // (the launch counts in tmpc_kernel_ms_total like any solve launch)
            HIP_TRY(h, tmpc::launch_solve_mc(v.d, v.shape, B, h->d_u, h->d_x0, h->d_ss, h->d_st, h->d_it, ws[0], d_mf, &h->wc, h->n_cu, h->stream));
            HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
            h->timed = true;
            h->mc_last_fused = 1;
        }
        // Otherwise, per time step: the solve launch(es) -- one per problem variant in use -- and ONE launch of the state machines
        // (round 3: mc_pre, the variant check, the solve, mc_post, mc_tube).  With injected packets nothing is solved.
        for (int t = 0; t < T && !fuse; ++t) {
            if (!rp) {
                int r3 = enqueue(h, B, st.x_hat, st.ref_k, extended ? st.gamma : nullptr, h->d_u, h->d_x0, h->d_ss, nullptr, h->d_st, h->d_it, ws, true);
                if (r3) return r3;
                st.ticks = h->want_ticks ? h->d_ticks : nullptr;       // (allocated by the first enqueue)
            }
            HIP_TRY(h, tmpc::launch_mc_step(m, st, t, T, B, ref[t], ref[t + 1 < T ? t + 1 : t], h->d_u, h->d_x0, h->d_ss, h->d_st, h->d_it, h->stream));
        }
        if (rp) {
            HIP_TRY(h, hipMemcpyAsync(rp->trace_f, st.trace_f, b * t_ * (3 * nx + nu) * 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(rp->trace_i, st.trace_i, b * t_ * 3 * 4, hipMemcpyDeviceToHost, h->stream));
        }
        if (err2) HIP_TRY(h, hipMemcpyAsync(err2, st.err2, b * 8, hipMemcpyDeviceToHost, h->stream));
        if (tube_viol) HIP_TRY(h, hipMemcpyAsync(tube_viol, st.tube_viol, b * 4, hipMemcpyDeviceToHost, h->stream));
        if (not_optimal) HIP_TRY(h, hipMemcpyAsync(not_optimal, st.not_optimal, b * 4, hipMemcpyDeviceToHost, h->stream));
        if (x_final) HIP_TRY(h, hipMemcpyAsync(x_final, st.x, b * nx * 8, hipMemcpyDeviceToHost, h->stream));
        if (consistent) HIP_TRY(h, hipMemcpyAsync(consistent, st.consistent, b * 8, hipMemcpyDeviceToHost, h->stream));
        if (iters_sum) HIP_TRY(h, hipMemcpyAsync(iters_sum, st.iters_sum, b * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return TMPC_OK;
    };
    rc = run();
    if (rc != TMPC_OK) (void)hipStreamSynchronize(h->stream);
    return rc;
}
}  // namespace

extern "C" {

int tmpc_mc_run(tmpc_handle *h, int64_t B, int32_t T, int extended, const double *p_loss, const double *ref,
                const double *th_u, const double *ga_u, const double *w, const double *x0, const double *HZ, const double *hZ,
                int32_t rZ, double *err2, int32_t *tube_viol, int32_t *not_optimal, double *x_final, double *consistent,
                int32_t *iters_sum) {
    return mc_run_impl(h, B, T, extended, p_loss, ref, th_u, ga_u, w, x0, HZ, hZ, rZ, err2, tube_viol, not_optimal, x_final, consistent,
                       iters_sum, nullptr);
}

int tmpc_mc_replay(tmpc_handle *h, int64_t B, int32_t T, int extended, const double *U_pkt, const double *xn0_pkt,
                   const uint8_t *theta, const uint8_t *gamma, const double *w, const double *x0, double *trace_f, int32_t *trace_i) {
    if (!h) return TMPC_E_INVALID;
    if (B < 0 || T < 0 || !U_pkt || !theta || !gamma || !w || !trace_f || !trace_i || (extended && !xn0_pkt)) { h->err = "tmpc_mc_replay: NULL argument"; return TMPC_E_INVALID; }
    // arrival flags as uniforms against a loss rate of one half: lost iff t > 0 and uniform < 1/2 (the draw rule of tmpc_mc_run)
    const size_t n = static_cast<size_t>(B) * static_cast<size_t>(T);
    std::vector<double> th(n), ga(n), pl(static_cast<size_t>(B), 0.5), ref(static_cast<size_t>(T), 0.0);
    for (size_t i = 0; i < n; ++i) { th[i] = theta[i] ? 1.0 : 0.0; ga[i] = gamma[i] ? 1.0 : 0.0; }
    McReplay rp{U_pkt, xn0_pkt, trace_f, trace_i};
    return mc_run_impl(h, B, T, extended, pl.data(), ref.data(), th.data(), ga.data(), w, x0, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, &rp);
}

int tmpc_synchronize(tmpc_handle *h) {
    if (!h) return TMPC_E_INVALID;
    if (h->device < 0) return TMPC_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_last_kernel_ms(tmpc_handle *h, float *ms) {
    if (!h || !ms) return TMPC_E_INVALID;
    if (!h->timed) { h->err = "tmpc_last_kernel_ms: no solve has been enqueued yet"; return TMPC_E_INVALID; }
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return TMPC_OK;
}

int tmpc_kernel_ms_total(tmpc_handle *h, float *total_ms, int32_t *launches, int reset) {
    if (!h) return TMPC_E_INVALID;
    if (h->device < 0) { h->err = "host-only handle"; return TMPC_E_DEVICE; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    float sum = 0.f;
    for (size_t i = 0; i < h->pool_used; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->pool[i].first, h->pool[i].second));
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = static_cast<int32_t>(h->pool_used);
    if (reset) h->pool_used = 0;
    return TMPC_OK;
}

#ifdef TMPC_STAMPS
int tmpc_debug_stamps(tmpc_handle *h, int variant, long long *out12 /* [16] */) {
    if (!h || !out12) return TMPC_E_INVALID;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const long long *src = use_block(h, h->v[variant]) ? h->v[variant].db.dbg : h->v[variant].d.dbg;
    HIP_TRY(h, hipMemcpy(out12, src, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return TMPC_OK;
}
#endif

int tmpc_get_dims(const tmpc_handle *h, int variant, int32_t *nv, int32_t *nc, int32_t *npar) {
    if (!h || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    const tmpc::Condensed &c = h->v[variant].c;
    if (nv) *nv = c.nv;
    if (nc) *nc = c.nc;
    if (npar) *npar = c.npar;
    return TMPC_OK;
}

int tmpc_get_factoring(const tmpc_handle *h, int variant, int32_t *nd, int32_t *ncc, int32_t *kc) {
    if (!h || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    const tmpc::Condensed &c = h->v[variant].c;
    if (nd) *nd = c.nd;
    if (ncc) *ncc = c.ncc;
    if (kc) *kc = c.kc;
    return TMPC_OK;
}

int tmpc_get_condensed(const tmpc_handle *h, int variant, double *H, double *F1, double *F2, double *G, double *g0, double *E) {
    if (!h || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    const tmpc::Condensed &c = h->v[variant].c;
    if (H) std::memcpy(H, c.H.a.data(), c.H.a.size() * sizeof(double));
    if (F1) std::memcpy(F1, c.F1.a.data(), c.F1.a.size() * sizeof(double));
    if (F2) std::memcpy(F2, c.F2.a.data(), c.F2.a.size() * sizeof(double));
    if (G) std::memcpy(G, c.G.a.data(), c.G.a.size() * sizeof(double));
    if (g0) std::memcpy(g0, c.g0.data(), c.g0.size() * sizeof(double));
    if (E) std::memcpy(E, c.E.a.data(), c.E.a.size() * sizeof(double));
    return TMPC_OK;
}

// ---- offline stage: batched support-function LPs (tmpc_lp.hip)

#define LP_TRY(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            g_create_error = std::string("tmpc_lp_batch: " #expr ": ") + hipGetErrorString(e_); \
            return TMPC_E_DEVICE;                                                          \
        }                                                                                  \
    } while (0)

namespace {
// the polytope in kernel units: rows to unit norm, h to max |h| = 1 (one scalar: x scales with it, the directions do not)
struct LpHost {
    int DP = 0, nrp = 0;
    double hm = 1.0;
    bool empty_set = false, no_normal = false;
    std::vector<double> Ht, hs, rs;
};

int lp_prepare(int32_t d, int32_t nr, const double *H, const double *hv, LpHost &o) {
    o.DP = tmpc::lp_padded_dim(d);
    if (d < 1 || o.DP < 0 || nr < 1) {
        g_create_error = "tmpc_lp_batch: need 1 <= d <= 32 and nr >= 1";
        return d > 32 ? TMPC_E_UNSUPPORTED : TMPC_E_INVALID;
    }
    const int nrp = o.nrp = (nr + 63) / 64 * 64;
    o.Ht.assign(static_cast<size_t>(o.DP) * nrp, 0.0);
    o.hs.assign(nrp, 1.0);
    o.rs.assign(nrp, 0.0);
    double hm = 0.0, nmax = 0.0;
    std::vector<double> nrm(nr, 0.0);
    for (int r = 0; r < nr; ++r) {
        double n2 = 0.0;
        for (int j = 0; j < d; ++j) {
            const double v = H[static_cast<size_t>(r) * d + j];
            if (!(v == v) || std::isinf(v)) { g_create_error = "tmpc_lp_batch: H is not finite"; return TMPC_E_INVALID; }
            n2 += v * v;
        }
        if (!(hv[r] == hv[r]) || std::isinf(hv[r])) { g_create_error = "tmpc_lp_batch: h is not finite"; return TMPC_E_INVALID; }
        nrm[r] = std::sqrt(n2);
        nmax = std::max(nmax, nrm[r]);
    }
    // A row whose normal vanishes against the others (round-off left by a product of matrices) says 0 <= h_r: it
    // constrains nothing, or everything.  Scaling it to unit norm would turn the round-off into a constraint.
    for (int r = 0; r < nr; ++r) {
        if (nrm[r] <= 1e-12 * nmax) {
            if (hv[r] < -1e-9 * (1.0 + std::fabs(hv[r]))) o.empty_set = true;
            continue;                                    // stays as the padding row 0 . x <= 1
        }
        o.rs[r] = 1.0 / nrm[r];
        for (int j = 0; j < d; ++j) o.Ht[static_cast<size_t>(j) * nrp + r] = H[static_cast<size_t>(r) * d + j] / nrm[r];
        o.hs[r] = hv[r] / nrm[r];
        hm = std::max(hm, std::fabs(o.hs[r]));
    }
    o.no_normal = !(nmax > 0.0);
    if (!(hm > 0.0)) hm = 1.0;
    o.hm = hm;
    for (int r = 0; r < nr; ++r) {
        if (o.rs[r] == 0.0) continue;                    // vanishing normal: keeps h = 1 in kernel units
        o.hs[r] /= hm; o.rs[r] /= hm;
    }
    return TMPC_OK;
}

constexpr int LP_MAX_ITER = 80;
constexpr double LP_TOL = 1e-8;
}  // namespace

int tmpc_lp_batch(int device, int32_t d, int32_t nr, const double *H, const double *hv, int64_t B, const double *C,
                  const int32_t *relax, double relax_by, double *val, double *x, int32_t *status, int32_t *iters) {
    if (!H || !hv || (B > 0 && (!C || !val || !status || !iters)) || B < 0) {
        g_create_error = "tmpc_lp_batch: NULL argument";
        return TMPC_E_INVALID;
    }
    if (d < 1 || tmpc::lp_padded_dim(d) < 0 || nr < 1) {
        g_create_error = "tmpc_lp_batch: need 1 <= d <= 32 and nr >= 1";
        return d > 32 ? TMPC_E_UNSUPPORTED : TMPC_E_INVALID;
    }
    if (relax)
        for (int64_t b = 0; b < B; ++b)
            if (relax[b] < -1 || relax[b] >= nr) { g_create_error = "tmpc_lp_batch: relax index out of range"; return TMPC_E_INVALID; }
    if (B == 0) return TMPC_OK;
    LpHost lh;
    if (const int rc = lp_prepare(d, nr, H, hv, lh); rc != TMPC_OK) return rc;
    const int nrp = lh.nrp;
    const std::vector<double> &Ht = lh.Ht, &hs = lh.hs, &rs = lh.rs;
    const double hm = lh.hm;
    if (lh.empty_set || lh.no_normal) {
        // 0 <= h_r < 0 for some r: no point satisfies the rows; no normal at all: every direction is unbounded
        for (int64_t b = 0; b < B; ++b) {
            val[b] = lh.empty_set ? std::nan("") : INFINITY;
            status[b] = lh.empty_set ? TMPC_STATUS_INFEASIBLE : TMPC_STATUS_UNBOUNDED;
            iters[b] = 0;
            if (x) for (int j = 0; j < d; ++j) x[b * d + j] = std::nan("");
        }
        return TMPC_OK;
    }

    LP_TRY(hipSetDevice(device));
    static int cu_count[64] = {};                       // hipGetDeviceProperties costs about a millisecond: once per device
    int n_cu = (device >= 0 && device < 64) ? cu_count[device] : 0;
    if (n_cu == 0) {
        LP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
        if (device >= 0 && device < 64) cu_count[device] = n_cu;
    }
    const int wpb = tmpc::lp_waves_per_block();
    const int64_t want = (B + wpb - 1) / wpb;
    const int nblocks = static_cast<int>(std::min<int64_t>(want, 2 * static_cast<int64_t>(n_cu)));
    const size_t b = static_cast<size_t>(B), dd = static_cast<size_t>(d);
    const size_t nws = static_cast<size_t>(nblocks) * wpb * tmpc::lp_workspace_arrays() * nrp;
    const size_t need = lp_round(Ht.size() * 8) + 2 * lp_round(static_cast<size_t>(nrp) * 8) + 2 * lp_round(b * dd * 8) + lp_round(nws * 8) +
                        lp_round(b * 8) + 3 * lp_round(b * 4) + 512;
    LpArena &ar = g_lp_arena;
    LP_TRY(ar.reserve(device, need));
    double *dHt = ar.take<double>(Ht.size()), *dh = ar.take<double>(nrp), *drs = ar.take<double>(nrp);
    double *dC = ar.take<double>(b * dd), *dws = ar.take<double>(nws), *dval = ar.take<double>(b);
    double *dx = x ? ar.take<double>(b * dd) : nullptr;
    int32_t *dst = ar.take<int32_t>(b), *dit = ar.take<int32_t>(b), *drel = relax ? ar.take<int32_t>(b) : nullptr;
    unsigned long long *dnext = ar.take<unsigned long long>(1);
    LP_TRY(hipMemset(dnext, 0, sizeof(unsigned long long)));
    if (relax) LP_TRY(hipMemcpy(drel, relax, b * sizeof(int32_t), hipMemcpyHostToDevice));
    LP_TRY(hipMemcpy(dHt, Ht.data(), Ht.size() * sizeof(double), hipMemcpyHostToDevice));
    LP_TRY(hipMemcpy(dh, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice));
    LP_TRY(hipMemcpy(drs, rs.data(), rs.size() * sizeof(double), hipMemcpyHostToDevice));
    LP_TRY(hipMemcpy(dC, C, b * dd * sizeof(double), hipMemcpyHostToDevice));
    tmpc::LpDevice lp{};
    lp.d = d; lp.nr = nr; lp.nrp = nrp; lp.max_iter = LP_MAX_ITER;
    lp.tol = LP_TOL; lp.relax_by = relax_by; lp.hm = hm;
    lp.Ht = dHt; lp.h = dh; lp.rscale = drs;
    lp.next_item = dnext;
    LP_TRY(tmpc::launch_lp(lp, B, nblocks, dC, drel, dws, dval, dx, dst, dit, nullptr));
    LP_TRY(hipDeviceSynchronize());
    LP_TRY(hipMemcpy(val, dval, b * sizeof(double), hipMemcpyDeviceToHost));
    LP_TRY(hipMemcpy(status, dst, b * sizeof(int32_t), hipMemcpyDeviceToHost));
    LP_TRY(hipMemcpy(iters, dit, b * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (x) LP_TRY(hipMemcpy(x, dx, b * dd * sizeof(double), hipMemcpyDeviceToHost));
    return TMPC_OK;
}

// Test support (tests/wavesim): the LP kernel's input in kernel units -- what tmpc_lp_batch uploads -- written to a file.
// No device is touched.  Format: int32 d, nr, nrp, DP, max_iter; double tol, relax_by, hm; Ht [DP][nrp], h [nrp], rscale [nrp].
int tmpc_debug_dump_lp_layout(int32_t d, int32_t nr, const double *H, const double *hv, double relax_by, const char *path) {
    if (!H || !hv || !path) { g_create_error = "tmpc_debug_dump_lp_layout: NULL argument"; return TMPC_E_INVALID; }
    LpHost lh;
    if (const int rc = lp_prepare(d, nr, H, hv, lh); rc != TMPC_OK) return rc;
    if (lh.empty_set || lh.no_normal) { g_create_error = "tmpc_debug_dump_lp_layout: the batch is decided on the host, no kernel input"; return TMPC_E_INVALID; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { g_create_error = "tmpc_debug_dump_lp_layout: cannot open the file"; return TMPC_E_INVALID; }
    const int32_t hd[5] = {d, nr, lh.nrp, lh.DP, LP_MAX_ITER};
    const double sc[3] = {LP_TOL, relax_by, lh.hm};
    bool ok = std::fwrite(hd, 4, 5, f) == 5 && std::fwrite(sc, 8, 3, f) == 3;
    ok = ok && std::fwrite(lh.Ht.data(), 8, lh.Ht.size(), f) == lh.Ht.size();
    ok = ok && std::fwrite(lh.hs.data(), 8, lh.hs.size(), f) == lh.hs.size();
    ok = ok && std::fwrite(lh.rs.data(), 8, lh.rs.size(), f) == lh.rs.size();
    std::fclose(f);
    if (!ok) { g_create_error = "tmpc_debug_dump_lp_layout: short write"; return TMPC_E_INVALID; }
    return TMPC_OK;
}

}  // extern "C"
