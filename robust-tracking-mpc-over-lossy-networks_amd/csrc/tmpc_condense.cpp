// See tmpc_condense.hpp.  Plain C++17, no dependencies.
#include "tmpc_condense.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace tmpc {
namespace {

Mat from_ptr(const double *p, int r, int c) {
    Mat m(r, c);
    if (p) std::memcpy(m.a.data(), p, sizeof(double) * static_cast<size_t>(r) * c);
    return m;
}
Mat mul(const Mat &A, const Mat &B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; ++i)
        for (int k = 0; k < A.c; ++k) {
            const double a = A(i, k);
            if (a == 0.0) continue;
            for (int j = 0; j < B.c; ++j) C(i, j) += a * B(k, j);
        }
    return C;
}
Mat tr(const Mat &A) {
    Mat T(A.c, A.r);
    for (int i = 0; i < A.r; ++i)
        for (int j = 0; j < A.c; ++j) T(j, i) = A(i, j);
    return T;
}
Mat add(const Mat &A, const Mat &B, double sb = 1.0) {
    Mat C = A;
    for (size_t i = 0; i < C.a.size(); ++i) C.a[i] += sb * B.a[i];
    return C;
}
void axpy(Mat &Y, const Mat &X, double s) {
    for (size_t i = 0; i < Y.a.size(); ++i) Y.a[i] += s * X.a[i];
}
Mat eye(int n) {
    Mat I(n, n);
    for (int i = 0; i < n; ++i) I(i, i) = 1.0;
    return I;
}

// A signal of the horizon as an affine function of (z, x_k, ref):  L z + Dx x_k + Dr ref
struct Aff {
    Mat L, Dx, Dr;
    Aff() = default;
    Aff(int dim, int nv, int nx) : L(dim, nv), Dx(dim, nx), Dr(dim, nx) {}
};
Aff sub(const Aff &a, const Aff &b) {
    Aff c;
    c.L = add(a.L, b.L, -1.0);
    c.Dx = add(a.Dx, b.Dx, -1.0);
    c.Dr = add(a.Dr, b.Dr, -1.0);
    return c;
}

// Orthonormal basis of null(S), S: r x c with full row rank (Householder QR of S').
bool null_space(const Mat &S, Mat &Nb) {
    const int m = S.c, n = S.r;   // factor S' (m x n)
    Mat M = tr(S), Q = eye(m);
    std::vector<double> v(m);
    for (int k = 0; k < n; ++k) {
        double nrm = 0;
        for (int i = k; i < m; ++i) nrm += M(i, k) * M(i, k);
        nrm = std::sqrt(nrm);
        if (nrm < 1e-12) return false;
        const double alpha = M(k, k) > 0 ? -nrm : nrm;
        std::fill(v.begin(), v.end(), 0.0);
        for (int i = k; i < m; ++i) v[i] = M(i, k);
        v[k] -= alpha;
        double vn = 0;
        for (int i = k; i < m; ++i) vn += v[i] * v[i];
        if (vn < 1e-300) continue;
        for (int j = k; j < n; ++j) {
            double d = 0;
            for (int i = k; i < m; ++i) d += v[i] * M(i, j);
            d = 2 * d / vn;
            for (int i = k; i < m; ++i) M(i, j) -= d * v[i];
        }
        for (int i = 0; i < m; ++i) {
            double d = 0;
            for (int j = k; j < m; ++j) d += Q(i, j) * v[j];
            d = 2 * d / vn;
            for (int j = k; j < m; ++j) Q(i, j) -= d * v[j];
        }
    }
    Nb = Mat(m, m - n);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m - n; ++j) Nb(i, j) = Q(i, n + j);
    return true;
}

// lower Cholesky in place; false if not positive definite
bool cholesky(Mat &M) {
    const int n = M.r;
    for (int j = 0; j < n; ++j) {
        double v = M(j, j);
        for (int k = 0; k < j; ++k) v -= M(j, k) * M(j, k);
        if (!(v > 0)) return false;
        v = std::sqrt(v);
        M(j, j) = v;
        for (int i = j + 1; i < n; ++i) {
            double t = M(i, j);
            for (int k = 0; k < j; ++k) t -= M(i, k) * M(j, k);
            M(i, j) = t / v;
        }
    }
    return true;
}

}  // namespace

std::string condense(const tmpc_problem &p, int variant, Condensed &out) {
    const int nx = p.nx, nu = p.nu, N = p.N;
    const bool received = variant == 1;
    if (received && !p.extended) return "variant 1 requested but problem is not extended";
    const bool fixed = !received && p.fixed_x0;
    // packet-received problem, literal terminal row (:293): auxiliaries eliminated when the caller supplies
    // the projection (HTP, hTP), kept with a vanishing weight otherwise (include/tmpc.h)
    const bool projected = received && p.literal_terminal_row && p.rTP > 0 && p.HTP && p.hTP;
    const bool aux = received && p.literal_terminal_row && !projected;
    const int rz = received ? p.rZW : (fixed ? 0 : p.rZ);
    const double *HZp = received ? p.HZW : p.HZ, *hZp = received ? p.hZW : p.hZ;
    if (!p.A || !p.B || !p.Q || !p.R || !p.P || !p.T) return "A, B, Q, R, P, T must be given";
    if (p.rx < 0 || p.ru < 0 || p.rT < 0) return "negative row count";
    if ((p.rx && (!p.Hx || !p.hx)) || (p.ru && (!p.Hu || !p.hu)) || (p.rT && (!p.HT || !p.hT)))
        return "constraint block declared but pointer is NULL";
    if (rz > 0 && (!HZp || !hZp)) return "initial-state set (HZ/HZW) missing";
    if (!fixed && rz == 0) return "free initial state needs the set Z (rZ > 0)";

    const Mat A = from_ptr(p.A, nx, nx), B = from_ptr(p.B, nx, nu), Q = from_ptr(p.Q, nx, nx),
              R = from_ptr(p.R, nu, nu), P = from_ptr(p.P, nx, nx), T = from_ptr(p.T, nx, nx);
    // steady-state parametrisation (TubeTrackingMPC.py:147)
    Mat S(nx, nx + nu);
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nx; ++j) S(i, j) = A(i, j) - (i == j ? 1.0 : 0.0);
        for (int j = 0; j < nu; ++j) S(i, nx + j) = B(i, j);
    }
    Mat Mth;
    if (!null_space(S, Mth)) return "[A-I, B] does not have full row rank; steady states are not parametrised by nu values";
    const int nth = Mth.c;

    Condensed c;
    c.nx = nx; c.nu = nu; c.N = N; c.nth = nth; c.Mth = Mth;
    c.off_theta = N * nu;
    int nvf = N * nu + nth;              // the parametrisation z_full = [u | theta | x_0 | aux] the outputs are read from
    if (!fixed) { c.off_x0 = nvf; nvf += nx; }
    if (aux) { c.off_aux = nvf; nvf += nx + nu; }
    const bool term_eq = !received && p.terminal_equality != 0;
    if (term_eq && p.rT > 0) return "terminal_equality and a terminal set (rT > 0) exclude each other";

    // z_full = Tz z + Tx x_k.  Identity unless an equality is eliminated below; the signals of the horizon are affine in
    // (z, x_k, ref) either way.
    Mat Tz = eye(nvf), Tx(nvf, nx);
    int nv = nvf;
    std::vector<Aff> x(N + 1), u(N);
    Aff xbar, ubar;
    auto selector = [&](int dim, int off) {
        Aff s(dim, nv, nx);
        for (int i = 0; i < dim; ++i) {
            for (int j = 0; j < nv; ++j) s.L(i, j) = Tz(off + i, j);
            for (int j = 0; j < nx; ++j) s.Dx(i, j) = Tx(off + i, j);
        }
        return s;
    };
    auto build_signals = [&]() {
        for (int i = 0; i < N; ++i) u[i] = selector(nu, i * nu);
        if (fixed) { x[0] = Aff(nx, nv, nx); x[0].Dx = eye(nx); }
        else x[0] = selector(nx, c.off_x0);
        for (int i = 0; i < N; ++i) {
            x[i + 1] = Aff(nx, nv, nx);
            x[i + 1].L = add(mul(A, x[i].L), mul(B, u[i].L));
            x[i + 1].Dx = add(mul(A, x[i].Dx), mul(B, u[i].Dx));
        }
        const Aff th = selector(nth, c.off_theta);
        xbar = Aff(nx, nv, nx);
        ubar = Aff(nu, nv, nx);
        Mat Mx(nx, nth), Mu(nu, nth);
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nth; ++j) Mx(i, j) = Mth(i, j);
        for (int i = 0; i < nu; ++i) for (int j = 0; j < nth; ++j) Mu(i, j) = Mth(nx + i, j);
        xbar.L = mul(Mx, th.L); xbar.Dx = mul(Mx, th.Dx);
        ubar.L = mul(Mu, th.L); ubar.Dx = mul(Mu, th.Dx);
    };
    build_signals();
    if (term_eq) {
        // TrackingMPC without a terminal set (TrackingMPC.py:105-107): x_N == x_bar, nx more equalities.  They are eliminated like
        // the others: C z_full + Cx x_k = 0 with C = L(x_N) - L(x_bar);  z_full = -C^+ Cx x_k + null(C) z.
        const Mat C = add(x[N].L, xbar.L, -1.0), Cx = add(x[N].Dx, xbar.Dx, -1.0);
        Mat Nc;
        if (!null_space(C, Nc)) return "terminal equality x_N == x_bar: the horizon is too short to reach a steady state (rank deficient)";
        Mat CCt = mul(C, tr(C));
        if (!cholesky(CCt)) return "terminal equality x_N == x_bar: rank deficient";
        // W = (C C')^-1 Cx by two triangular solves per column, then Tx = -C' W
        Mat W(nx, nx);
        for (int k = 0; k < nx; ++k) {
            std::vector<double> col(nx);
            for (int i = 0; i < nx; ++i) {
                double t = Cx(i, k);
                for (int j = 0; j < i; ++j) t -= CCt(i, j) * col[j];
                col[i] = t / CCt(i, i);
            }
            for (int i = nx - 1; i >= 0; --i) {
                double t = col[i];
                for (int j = i + 1; j < nx; ++j) t -= CCt(j, i) * col[j];
                col[i] = t / CCt(i, i);
            }
            for (int i = 0; i < nx; ++i) W(i, k) = col[i];
        }
        Tx = mul(tr(C), W);
        for (double &v : Tx.a) v = -v;
        Tz = Nc;
        nv = Nc.c;
        if (nv < 1) return "terminal equality x_N == x_bar leaves no degree of freedom";
        build_signals();
        c.Tz = Tz;
        c.Tx = Tx;
    }
    c.nv = nv;
    c.nvf = nvf;
    Aff rsig(nx, nv, nx);
    rsig.Dr = eye(nx);

    // ---- cost (TubeTrackingMPC.py:136,143,144):  sum ||e||^2_W  with e = L z + Dx x + Dr r
    c.H = Mat(nv, nv); c.F1 = Mat(nv, nx); c.F2 = Mat(nv, nx);
    auto add_cost = [&](const Aff &e, const Mat &W) {
        const Mat LtW = mul(tr(e.L), W);
        axpy(c.H, mul(LtW, e.L), 2.0);
        axpy(c.F1, mul(LtW, e.Dx), 2.0);
        axpy(c.F2, mul(LtW, e.Dr), 2.0);
    };
    for (int i = 0; i < N; ++i) {
        add_cost(sub(x[i], xbar), Q);
        add_cost(sub(u[i], ubar), R);
    }
    add_cost(sub(x[N], xbar), P);
    add_cost(sub(xbar, rsig), T);
    for (int i = 0; i < nv; ++i)
        for (int j = 0; j < i; ++j) { const double a = 0.5 * (c.H(i, j) + c.H(j, i)); c.H(i, j) = c.H(j, i) = a; }
    if (aux) {
        // The auxiliaries of TubeTrackingMPC.py:293 carry no cost, so the QP is not strictly
        // convex in them (any feasible value is optimal; the returned x, u, x_bar, u_bar do
        // not depend on the choice).  A vanishing weight eps*|aux|^2, eps = 2e-6 min(diag R),
        // selects one; it moves the reported minimiser by O(1e-11).
        double rmin = R(0, 0);
        for (int i = 1; i < nu; ++i) rmin = std::min(rmin, R(i, i));
        for (int i = c.off_aux; i < nv; ++i) c.H(i, i) += 2e-6 * rmin;
    }

    // ---- constraints, in the reference's order
    std::vector<std::vector<double>> Grow, Erow;
    std::vector<double> hrow, hcn;
    std::vector<int> rowblk, rowidx;      // block id (1 = terminal, 2 = initial state) and row number inside its block
    int cur_blk = 0;
    auto add_rows = [&](const Mat &Hc, const double *hc, const Aff &sig) {
        const Mat G = mul(Hc, sig.L), E = mul(Hc, sig.Dx);
        for (int r = 0; r < Hc.r; ++r) {
            Grow.emplace_back(G.a.begin() + static_cast<size_t>(r) * nv, G.a.begin() + static_cast<size_t>(r + 1) * nv);
            std::vector<double> e(nx);
            for (int j = 0; j < nx; ++j) e[j] = -E(r, j);
            Erow.push_back(e);
            hrow.push_back(hc[r]);
            double n2 = 0;
            for (int j = 0; j < Hc.c; ++j) n2 += Hc(r, j) * Hc(r, j);
            hcn.push_back(std::sqrt(n2));
            rowblk.push_back(cur_blk);
            rowidx.push_back(r);
        }
    };
    if (!fixed) {
        // Hz (x_k - x_0) <= hz  (TubeTrackingMPC.py:132 / :278)
        Aff xk(nx, nv, nx);
        xk.Dx = eye(nx);
        cur_blk = 2;
        add_rows(from_ptr(HZp, rz, nx), hZp, sub(xk, x[0]));
        cur_blk = 0;
    }
    const Mat Hx = from_ptr(p.Hx, p.rx, nx), Hu = from_ptr(p.Hu, p.ru, nu);
    for (int i = 0; i < N; ++i) {
        add_rows(Hx, p.hx, x[i]);        // :139
        add_rows(Hu, p.hu, u[i]);        // :140
    }
    if (projected) {
        // HTP [x_bar; u_bar] <= hTP: line :293 after eliminating the free (x_N', u_bar')
        Aff st(nx + nu, nv, nx);
        for (int j = 0; j < nv; ++j) {
            for (int i = 0; i < nx; ++i) st.L(i, j) = xbar.L(i, j);
            for (int i = 0; i < nu; ++i) st.L(nx + i, j) = ubar.L(i, j);
        }
        add_rows(from_ptr(p.HTP, p.rTP, nx + nu), p.hTP, st);
    } else {
        // HT [x_T; x_bar; u_T] <= hT   (:149; for variant 1 literally :293)
        const Aff xT = aux ? selector(nx, c.off_aux) : x[N];
        const Aff uT = aux ? selector(nu, c.off_aux + nx) : ubar;
        Aff st(2 * nx + nu, nv, nx);
        for (int i = 0; i < nx; ++i) {
            for (int j = 0; j < nv; ++j) { st.L(i, j) = xT.L(i, j); st.L(nx + i, j) = xbar.L(i, j); }
            for (int j = 0; j < nx; ++j) st.Dx(i, j) = xT.Dx(i, j);
        }
        for (int i = 0; i < nu; ++i) for (int j = 0; j < nv; ++j) st.L(2 * nx + i, j) = uT.L(i, j);
        cur_blk = 1;
        add_rows(from_ptr(p.HT, p.rT, 2 * nx + nu), p.hT, st);
        cur_blk = 0;
    }
    // factored form of the terminal block: rows = HcT * PsiT with
    //   PsiT = [x_T map (nx rows); theta selector (nth rows); u_aux selector (nu rows, variant 1 only)]
    const int kT = nx + nth + (aux ? nu : 0);
    Mat PsiT(kT, nv), HcT(p.rT, kT);
    {
        const Aff xT = aux ? selector(nx, c.off_aux) : x[N];
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nv; ++j) PsiT(i, j) = xT.L(i, j);
        if (!term_eq) {          // (with the terminal equality there is no terminal block, and z is not z_full)
            for (int i = 0; i < nth; ++i) PsiT(nx + i, c.off_theta + i) = 1.0;
            if (aux) for (int i = 0; i < nu; ++i) PsiT(nx + nth + i, c.off_aux + nx + i) = 1.0;
        }
        const Mat HT = from_ptr(p.HT, p.rT, 2 * nx + nu);
        for (int r = 0; r < p.rT; ++r) {
            for (int j = 0; j < nx; ++j) HcT(r, j) = HT(r, j);
            for (int j = 0; j < nth; ++j) {
                double v = 0;
                for (int i = 0; i < nx; ++i) v += HT(r, nx + i) * Mth(i, j);
                if (!aux) for (int i = 0; i < nu; ++i) v += HT(r, 2 * nx + i) * Mth(nx + i, j);
                HcT(r, nx + j) = v;
            }
            if (aux) for (int j = 0; j < nu; ++j) HcT(r, nx + nth + j) = HT(r, 2 * nx + j);
        }
    }

    // ---- classify rows: iterate / x_k-only / constant
    const int mi = static_cast<int>(Grow.size());
    std::vector<int> keep, par;
    for (int r = 0; r < mi; ++r) {
        double gn = 0, en = 0;
        for (double v : Grow[r]) gn += v * v;
        for (double v : Erow[r]) en += v * v;
        const double thr = 1e-11 * (1.0 + hcn[r]);
        if (std::sqrt(gn) > thr) keep.push_back(r);
        else if (std::sqrt(en) > thr) par.push_back(r);
        else if (hrow[r] < -1e-9 * (1.0 + std::fabs(hrow[r]))) c.always_infeasible = true;
    }
    // one block is kept in factored form when that pays (many rows, small rank): the terminal block (its rows then go
    // last) or, when it is the larger one, the initial-state block (its rows stay first).  Everything else, in the
    // reference's order, is "dense".
    int fblk = 0;          // 0: nothing factored
    {
        int nterm = 0, ninit = 0;
        for (int r : keep) { nterm += rowblk[r] == 1; ninit += rowblk[r] == 2; }
        const bool term_ok = nterm >= 64 && kT + 3 <= nv;
        const bool init_ok = !fixed && ninit >= 64 && nx + 3 <= nv;
        if (term_ok && (!init_ok || nterm >= ninit)) fblk = 1;
        else if (init_ok) fblk = 2;
        std::vector<int> dense_rows, fact_rows;
        for (int r : keep) ((fblk != 0 && rowblk[r] == fblk) ? fact_rows : dense_rows).push_back(r);
        c.nz = 0;
        if (!fixed)
            for (int r : keep) { if (rowblk[r] == 2) ++c.nz; else break; }      // they come first in the reference's order
        if (fblk == 2) {
            keep = fact_rows;
            keep.insert(keep.end(), dense_rows.begin(), dense_rows.end());
            c.fb0 = 0;
        } else {
            keep = dense_rows;
            keep.insert(keep.end(), fact_rows.begin(), fact_rows.end());
            c.fb0 = static_cast<int>(dense_rows.size());
        }
        c.nd = static_cast<int>(dense_rows.size());
        c.ncc = static_cast<int>(fact_rows.size());
        c.kc = fblk == 1 ? kT : (fblk == 2 ? nx : 0);
    }
    c.nc = static_cast<int>(keep.size());
    c.npar = static_cast<int>(par.size());
    c.G = Mat(c.nc, nv); c.E = Mat(c.nc, nx); c.g0.resize(c.nc);
    c.Ep = Mat(c.npar, nx); c.gp0.resize(c.npar);
    for (int a = 0; a < c.nc; ++a) {
        for (int j = 0; j < nv; ++j) c.G(a, j) = Grow[keep[a]][j];
        for (int j = 0; j < nx; ++j) c.E(a, j) = Erow[keep[a]][j];
        c.g0[a] = hrow[keep[a]];
    }
    for (int a = 0; a < c.npar; ++a) {
        for (int j = 0; j < nx; ++j) c.Ep(a, j) = Erow[par[a]][j];
        c.gp0[a] = hrow[par[a]];
    }

    // ---- scaling: Jacobi on z, unit rows
    c.Dv.resize(nv);
    for (int i = 0; i < nv; ++i) {
        if (!(c.H(i, i) > 0)) return "condensed Hessian has a non-positive diagonal entry";
        c.Dv[i] = 1.0 / std::sqrt(c.H(i, i));
    }
    c.Hs = Mat(nv, nv); c.F1s = Mat(nv, nx); c.F2s = Mat(nv, nx);
    for (int i = 0; i < nv; ++i) {
        for (int j = 0; j < nv; ++j) c.Hs(i, j) = c.H(i, j) * c.Dv[i] * c.Dv[j];
        for (int j = 0; j < nx; ++j) { c.F1s(i, j) = c.F1(i, j) * c.Dv[i]; c.F2s(i, j) = c.F2(i, j) * c.Dv[i]; }
    }
    c.Gs = Mat(c.nc, nv); c.Es = Mat(c.nc, nx); c.g0s.resize(c.nc);
    for (int r = 0; r < c.nc; ++r) {
        double n2 = 0;
        for (int j = 0; j < nv; ++j) { const double v = c.G(r, j) * c.Dv[j]; c.Gs(r, j) = v; n2 += v * v; }
        const double rn = std::sqrt(n2);
        for (int j = 0; j < nv; ++j) c.Gs(r, j) /= rn;
        for (int j = 0; j < nx; ++j) c.Es(r, j) = c.E(r, j) / rn;
        c.g0s[r] = c.g0[r] / rn;
    }
    if (c.ncc > 0) {
        c.Psi = Mat(c.kc, nv);
        c.Hc = Mat(c.ncc, c.kc);
        if (fblk == 1) {
            for (int a = 0; a < c.kc; ++a) for (int j = 0; j < nv; ++j) c.Psi(a, j) = PsiT(a, j) * c.Dv[j];
        } else {
            // Hz (x_k - x_0): the rows are -Hz on the x_0 block of z  (x_0 = rows off_x0.. of Tz z + Tx x_k)
            for (int a = 0; a < nx; ++a) for (int j = 0; j < nv; ++j) c.Psi(a, j) = Tz(c.off_x0 + a, j) * c.Dv[j];
        }
        double worst = 0;
        for (int r = 0; r < c.ncc; ++r) {
            const int src = rowidx[keep[c.fb0 + r]];
            double n2 = 0;
            for (int j = 0; j < nv; ++j) { const double v = c.G(c.fb0 + r, j) * c.Dv[j]; n2 += v * v; }
            const double rn = std::sqrt(n2);
            for (int a = 0; a < c.kc; ++a) c.Hc(r, a) = (fblk == 1 ? HcT(src, a) : -HZp[static_cast<size_t>(src) * nx + a]) / rn;
            for (int j = 0; j < nv; ++j) {           // consistency of the factorisation with the dense rows
                double v = 0;
                for (int a = 0; a < c.kc; ++a) v += c.Hc(r, a) * c.Psi(a, j);
                worst = std::max(worst, std::fabs(v - c.Gs(c.fb0 + r, j)));
            }
        }
        if (worst > 1e-10) return "internal error: factored block does not reproduce its rows";
    }
    // mirror rows (the two sides of box-type constraints), within the dense class and within the factored class
    c.mirror.assign(c.nc, -1);
    {
        auto in_fact = [&](int r) { return c.ncc > 0 && r >= c.fb0 && r < c.fb0 + c.ncc; };
        for (int r = 0; r < c.nc; ++r) {
            if (c.mirror[r] >= 0) continue;
            for (int q = r + 1; q < c.nc; ++q) {
                if (c.mirror[q] >= 0 || in_fact(q) != in_fact(r)) continue;
                double dmax = 0;
                for (int j = 0; j < nv && dmax <= 1e-13; ++j) dmax = std::max(dmax, std::fabs(c.Gs(r, j) + c.Gs(q, j)));
                if (dmax > 1e-13) continue;
                if (in_fact(r)) {      // the factored rows are applied through Hc: require the mirror there as well
                    double hmax = 0;
                    for (int a = 0; a < c.kc; ++a) hmax = std::max(hmax, std::fabs(c.Hc(r - c.fb0, a) + c.Hc(q - c.fb0, a)));
                    if (hmax > 1e-13) continue;
                }
                c.mirror[r] = q;
                c.mirror[q] = r;
                break;
            }
        }
    }
    // Hs^-1 (used for the unconstrained minimiser and by the active-set refinement)
    c.Hinv = Mat(nv, nv);
    {
        Mat L = c.Hs;
        if (!cholesky(L)) return "condensed Hessian is not positive definite";
        std::vector<double> col(nv);
        for (int k = 0; k < nv; ++k) {
            std::fill(col.begin(), col.end(), 0.0);
            col[k] = 1.0;
            for (int i = 0; i < nv; ++i) {
                double t = col[i];
                for (int j = 0; j < i; ++j) t -= L(i, j) * col[j];
                col[i] = t / L(i, i);
            }
            for (int i = nv - 1; i >= 0; --i) {
                double t = col[i];
                for (int j = i + 1; j < nv; ++j) t -= L(j, i) * col[j];
                col[i] = t / L(i, i);
            }
            for (int i = 0; i < nv; ++i) c.Hinv(i, k) = col[i];
        }
    }
    out = std::move(c);
    return "";
}

}  // namespace tmpc
