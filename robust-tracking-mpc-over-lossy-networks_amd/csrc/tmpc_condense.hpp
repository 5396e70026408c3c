// Host-side (C++) condensing of the tube-tracking QP into the dense form the HIP
// kernels iterate on.  Runs once per model inside tmpc_create().
//
// Reference formulation: TubeTrackingMPC.generate_optimization_problem
// (reference src/LinearMPCOverNetworks/TubeTrackingMPC.py:104-156) hands cvxpy the
// un-condensed problem in (x_mpc, u_mpc, x_bar, u_bar).  Here the equalities are
// eliminated analytically:
//
//   x_i        = A^i x_0 + sum_{j<i} A^(i-1-j) B u_j          (dynamics, :138)
//   [x_bar;u_bar] = Mth * theta,  Mth = null([A-I, B])         (steady state, :147)
//   x_0        = x_k (fixed_initial_state, :127)  or a decision variable (:132)
//
// leaving   z = [u_0 .. u_{N-1} | theta | x_0 (if free) | x_aux,u_aux (variant 1, :293)]
// and       min 1/2 z'Hz + (F1 x_k + F2 ref)'z   s.t.   G z <= g0 + E x_k .
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tmpc.h"

namespace tmpc {

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(int r_, int c_) : r(r_), c(c_), a(static_cast<size_t>(r_) * c_, 0.0) {}
    double &operator()(int i, int j) { return a[static_cast<size_t>(i) * c + j]; }
    double operator()(int i, int j) const { return a[static_cast<size_t>(i) * c + j]; }
};

// One condensed QP variant, unscaled and scaled.
struct Condensed {
    int nx = 0, nu = 0, N = 0;
    int nv = 0;        // decision variables
    int nvf = 0;       // length of z_full = [u | theta | x_0 | aux] = Tz z + Tx x_k, which the outputs are read from
    Mat Tz, Tx;        // nvf x nv, nvf x nx; empty = identity / zero (no equality eliminated beyond the standard ones)
    int nc = 0;        // inequality rows the solver iterates on
    int npar = 0;      // rows that depend on x_k only
    int nth = 0;       // dim(theta)
    int off_theta = 0, off_x0 = -1, off_aux = -1;   // offsets into z (-1: absent)
    bool always_infeasible = false;
    // unscaled:  min 1/2 z'Hz + (F1 x + F2 r)'z  s.t. G z <= g0 + E x ;  0 <= gp0 + Ep x
    Mat H, F1, F2, G, E, Ep, Mth;
    std::vector<double> g0, gp0;
    // scaled: z = Dv .* zs, rows divided by rn (unit row norm after column scaling)
    std::vector<double> Dv;
    Mat Hs, Hinv, Gs, Es, F1s, F2s;
    std::vector<double> g0s;
    // Row order.  One block of rows of small rank may be kept in FACTORED form  Gs[fb0 + r, :] = Hc(r, :) * Psi, r < ncc:
    //   * the terminal inequality (TubeTrackingMPC.py:149) acts on [x_N; x_bar; u_bar] only, i.e. on kc = nx + nth (+ nu)
    //     combinations of z; its rows then go LAST (fb0 = nd);
    //   * the initial-state rows Hz (x_k - x_0) <= hz of a free x_0 (:132, :278) act on x_0 only (kc = nx); they come FIRST
    //     in the reference's order and stay there (fb0 = 0) -- chosen when they outnumber the terminal rows (the
    //     packet-received problem: 854 rows of Z (-) W against the projected terminal rows).
    // The other nd = nc - ncc rows are "dense" (general rows of Gs), in the reference's order.  ncc == 0: nothing factored.
    int nd = 0, ncc = 0, kc = 0, fb0 = 0;
    int nz = 0;        // the first nz rows are the initial-state rows Hz (x_k - x_0) <= hz: they act on the x_0 block of z only
    Mat Psi;     // kc x nv, scaled with Dv
    Mat Hc;      // ncc x kc, rows scaled like Gs
    // mirror[r] = q when scaled row q is the exact mirror image of row r (Gs_q = -Gs_r: the two sides of a box-type
    // constraint) and both lie in the same class (dense / factored); -1 otherwise.
    std::vector<int> mirror;
};

// Builds variant 0 (base problem) or 1 (packet-received problem).  Returns "" on
// success, otherwise an error message.
std::string condense(const tmpc_problem &p, int variant, Condensed &out);

}  // namespace tmpc
