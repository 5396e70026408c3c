"""Solver-independent evidence on the HIP outputs, for every BASELINE configuration.

The oracle (oracle/tmpc_oracle.c) runs the same interior-point + refinement scheme as the
kernels, so agreement with it does not by itself show that either is right.  Here nothing of that
scheme is used: every instance the library returns as OPTIMAL must satisfy the KKT conditions of
the QP exactly as the reference states it in its own, un-condensed variables
(TubeTrackingMPC.py:104-156, :253-299; oracle/qp_sparse.py builds it line by line) -- primal
feasibility, stationarity with non-negative multipliers recovered by a least-squares fit -- and
every instance it returns as INFEASIBLE must be infeasible for HiGHS (scipy.optimize.linprog, the
LP solver the reference itself calls, utils_polytope.py:19).  The QPs are strictly convex
(SURVEY.md A.4), so a KKT point is THE minimiser, whichever solver produced it.

>= 256 instances per configuration: closed-loop states, random states, and states pushed onto the
boundary of the tightened state set (many active rows, degenerate vertices, infeasible cases).
"""
import os

import numpy as np
import pytest

import common
from oracle import qp_sparse

pytestmark = pytest.mark.gpu

S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
TOL_STAT, TOL_FEAS = 1e-7, 1e-9
# north_star's parity band is 1e-6 on u*_0; the residual-based certificate above scales with |q| (1e6 for the cart-pole)
# and lets a 1e-6 shift of u_0 through, so every answer is also measured against the exact minimiser on its certified
# active set (qp_sparse.minimiser_distance: a linear solve, no interior-point or refinement code involved)
TOL_U0 = 1e-8


def boundary_states(rng, hx_box, n, lo=0.9):
    """States with one coordinate at lo..1.0 of the tightened box, the others anywhere inside."""
    X = rng.uniform(-1, 1, (n, len(hx_box))) * hx_box
    k = rng.integers(0, len(hx_box), n)
    X[np.arange(n), k] = rng.choice([-1.0, 1.0], n) * rng.uniform(lo, 1.0, n) * hx_box[k]
    return X


def certify(mpc, X, R, variant=None, min_optimal=128, literal_check=None):
    """Solves the batch on the GPU and certifies every answer; returns (n_optimal, n_infeasible)."""
    p = mpc._problem_dict()
    out = mpc._solve(X, R, variant)
    var = np.zeros(len(X), np.uint8) if variant is None else np.broadcast_to(np.asarray(variant, np.uint8), (len(X),))
    tpl = {v: qp_sparse.SparseTemplate(p, int(v)) for v in np.unique(var)}
    st = out["status"]
    assert np.all((st == 0) | (st == 2)), np.bincount(st)
    worst = dict(r_stat=0.0, r_eq=0.0, r_ineq=0.0, du0=0.0)
    n_inf = 0
    for k in range(len(X)):
        qp = tpl[var[k]].instance(X[k], R[k])
        if st[k] == 2:
            tight = dict(qp)
            tight["h"] = qp["h"] - 1e-6 * np.maximum(1.0, np.abs(qp["h"]))       # borderline instances may go either way
            assert qp_sparse.lp_infeasible(tight), f"instance {k}: library says infeasible, HiGHS finds a strictly feasible point"
            assert np.all(np.isnan(out["u_nom"][k]))
            n_inf += 1
            continue
        v = qp_sparse.pack(qp, out["x_nom"][k], out["u_nom"][k], out["x_ss"][k], out["u_ss"][k])
        c = qp_sparse.kkt_certificate_fast(qp, v)
        lam_scale = max(1.0, float(np.abs(c["lam"]).max())) if len(c["lam"]) else 1.0
        assert c["r_eq"] < TOL_FEAS and c["r_ineq"] < TOL_FEAS and c["r_stat"] < TOL_STAT and c["min_lam"] >= -1e-9 * lam_scale, (k, c)
        for key in ("r_stat", "r_eq", "r_ineq"):
            worst[key] = max(worst[key], c[key])
        d = qp_sparse.minimiser_distance(qp, v, active=c["active"])
        assert d["certified"], (k, {a: d[a] for a in ("r_ineq", "min_mu", "r_stat", "resolution", "n_active")})
        assert d["du0"] <= TOL_U0, (k, d["du0"], c["n_active"])
        worst["du0"] = max(worst["du0"], d["du0"])
        if literal_check is not None and var[k] == 1:
            literal_check(out["x_ss"][k], out["u_ss"][k])
    n_opt = int((st == 0).sum())
    assert n_opt >= min_optimal, (n_opt, n_inf)
    print(f"certified {n_opt} optimal (worst {worst}), {n_inf} infeasible by LP")
    return n_opt, n_inf


@pytest.mark.parametrize("N", [5, 10])
def test_config1_double_integrator_free_x0(hip_lib, N):
    """BASELINE config 1 (Example_of_Tube_Tracking_MPC.py: free initial state, Rakovic sets)."""
    mpc, w = common.make_mpc("double_integrator", N, False, create=True)
    rng = np.random.default_rng(11)
    X = np.r_[rng.uniform(-1, 1, (160, 2)) * [7.5, 0.9], boundary_states(rng, np.array([8.0, 1.2]), 128, 0.85)]
    R = np.c_[rng.choice([5.0, -9.0, 9.0, 4.0, 0.0], len(X)) + rng.uniform(-1, 1, len(X)), np.zeros(len(X))]
    n_opt, n_inf = certify(mpc, X, R, min_optimal=150)
    assert n_inf > 0


@pytest.mark.parametrize("N", [10, 20])
def test_cartpole_base_problem(hip_lib, N):
    """BASELINE config 2 (N = 10) and the reference scripts' horizon (N = 20, results_linear_system.py:64)."""
    mpc, w = common.make_mpc("cartpole", N, True, create=True)
    rng = np.random.default_rng(12 + N)
    idx = rng.choice(len(S), 200, replace=False)
    Xb = boundary_states(rng, mpc._Xc.b[:4] * np.array([0.5, 0.4, 0.9, 0.5]), 160, 0.7)     # inside the region the terminal set reaches
    X = np.r_[S[idx, :4], Xb]
    R = np.r_[S[idx, 4:], np.c_[rng.uniform(-2, 2, len(Xb)), np.zeros((len(Xb), 3))]]
    n_opt, n_inf = certify(mpc, X, R, min_optimal=256)


@pytest.mark.parametrize("N,path", [(10, "auto"), (20, "auto"), (20, "block")])
def test_extended_packet_received_problem(hip_lib, N, path):
    """BASELINE config 3: the gamma = 1 problem (TubeTrackingMPC.py:253-299) in its projected form, mixed with
    gamma = 0 instances in one call; plus the literal line :293 -- for the returned x_bar there must exist free
    auxiliaries (x_aux, u_aux) with HT [x_aux; x_bar; u_aux] <= hT."""
    from scipy.optimize import linprog
    mpc, w = common.make_mpc("cartpole", N, True, extended=True, create=True)
    mpc.set_kernel_path(path)              # "block": the workgroup-per-QP kernel on the same instances (1056 rows, degenerate vertices)
    assert mpc.get_kernel_path(1) == ("wave" if path == "auto" else "block")
    p = mpc._problem_dict()
    HT, hT = np.asarray(p["HT"]), np.asarray(p["hT"])
    nx, nu = 4, 1
    Ha = np.c_[HT[:, :nx], HT[:, 2 * nx:]]
    checked = [0]

    def literal(x_ss, u_ss):
        if checked[0] >= 48:            # an LP each; a sample is enough
            return
        res = linprog(np.zeros(nx + nu), A_ub=Ha, b_ub=hT - HT[:, nx:2 * nx] @ x_ss + 1e-9, bounds=[(None, None)] * (nx + nu), method="highs")
        assert res.status == 0, "x_bar of the projected problem violates the literal terminal row for every choice of the auxiliaries"
        checked[0] += 1

    rng = np.random.default_rng(3 + N)
    idx = rng.choice(len(S), 192, replace=False)
    Xb = boundary_states(rng, mpc._Xc.b[:4] * np.array([0.5, 0.4, 0.9, 0.5]), 128, 0.7)
    X = np.r_[S[idx, :4], Xb]
    # x_k of a gamma = 1 instance is the plant state: off the nominal state by an element of Z (-) W
    X = X + rng.uniform(-1, 1, X.shape) * w["w_bound"] * 3.0
    R = np.r_[S[idx, 4:], np.c_[rng.uniform(-2, 2, len(Xb)), np.zeros((len(Xb), 3))]]
    gam = (rng.uniform(size=len(X)) < 0.75).astype(np.uint8)
    n_opt, n_inf = certify(mpc, X, R, gam, min_optimal=200, literal_check=literal)
    assert checked[0] >= 32


def test_config5_synthetic(hip_lib):
    """BASELINE config 5: n = 12, m = 4, N = 30 (nv = 124; block kernel, MFMA normal matrix)."""
    mpc, w = common.make_mpc("synthetic", 30, True, create=True)
    rng = np.random.default_rng(5)
    B = 256
    X = rng.uniform(-0.5, 0.5, (B, 12)) * mpc._Xc.b[:12]
    X[:96] *= 1.9
    R = np.zeros((B, 12))
    R[:, 0] = rng.uniform(-2, 2, B)
    certify(mpc, X, R, min_optimal=200)


@pytest.mark.parametrize("name", ["double_integrator", "cartpole"])
def test_tracking_mpc(hip_lib, name):
    """R-MPC comparator (TrackingMPC.py:62-115): un-tightened sets, x_0 fixed."""
    from LinearMPCOverNetworks import polytope_lite, workloads
    from LinearMPCOverNetworks.TrackingMPC import TrackingMPC
    w = workloads.double_integrator() if name == "double_integrator" else workloads.cartpole()
    N = 10
    mpc = TrackingMPC(w["A"], w["B"], w["Q"], w["R"], N)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    polytope_lite.set_lp_backend("hip")          # the cartpole's terminal set is thousands of LPs (0.2 s on the device, 20 s with HiGHS)
    try:
        mpc.setup_optimization()
    finally:
        polytope_lite.set_lp_backend("scipy")
    rng = np.random.default_rng(21)
    nx = w["A"].shape[0]
    if name == "double_integrator":
        X = np.r_[rng.uniform(-1, 1, (192, 2)) * [6.0, 1.0], boundary_states(rng, np.array([8.0, 1.5]), 96, 0.8)]
        R = np.c_[rng.uniform(-7, 7, len(X)), np.zeros(len(X))]
        min_opt = 150
    else:
        idx = rng.choice(len(S), 192, replace=False)
        Xb = boundary_states(rng, np.array([1.5, 1.0, 0.2, 0.6]), 96, 0.7)
        X = np.r_[S[idx, :4], Xb]
        R = np.r_[S[idx, 4:], np.c_[rng.uniform(-2, 2, len(Xb)), np.zeros((len(Xb), 3))]]
        min_opt = 200
    certify(mpc, X, R, min_optimal=min_opt)


def test_known_minimiser_on_device(hip_lib):
    """SURVEY Appendix D's solver-independent anchor (scipy trust-constr + exact active-set polish, sparse and condensed
    form): double integrator, N = 5, free x_0, Darup sets, x_k = [1, 2], r = [5, 0]."""
    mpc, _ = common.make_mpc("double_integrator_darup", 5, False, create=True)
    x_nom, u_nom, x_ss, u_ss = mpc.solve_optimization_problem(np.array([1.0, 2.0]), np.array([5.0, 0.0]))
    assert abs(u_nom[0, 0] - (-0.737182900857)) < 1e-9
    np.testing.assert_allclose(x_nom[:, 0], [1.534713867, 1.745580949], atol=1e-8)
    np.testing.assert_allclose(x_ss, [4.93424, 0.0], atol=1e-5)
