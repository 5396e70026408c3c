"""Pins the CPU oracle (oracle/tmpc_oracle.c).  CPU only.

The reference has no golden vectors for the QP solution (PARITY UNPINNED, see
oracle/qp_sparse.py).  What stands in: (1) an independent KKT certificate, computed in
numpy from the QP written exactly as the reference writes it for cvxpy; (2) the numpy
interior-point restatement; (3) scipy's SLSQP at a small size; (4) the committed
fixture the GPU parity tests compare against.
"""
import os

import numpy as np
import pytest
from scipy.optimize import minimize

import common
from oracle import ipm_numpy, qp_sparse
from oracle.oracle import Oracle

S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))


@pytest.fixture(scope="module")
def cartpole(oracle_lib):
    mpc, w = common.make_mpc("cartpole", 10, True)
    p = mpc._problem_dict()
    return p, Oracle(p), mpc, w


def _certify(p, x, r, sol, i, variant=0, tol_stat=1e-7):
    qp = qp_sparse.build_sparse_qp(p, x, r, variant)
    v = qp_sparse.pack(qp, sol["x_nom"][i], sol["u_nom"][i], sol["x_ss"][i], sol["u_ss"][i])
    if variant == 1:
        return qp, v       # auxiliaries are not returned by the solver; certified elsewhere
    c = qp_sparse.kkt_certificate(qp, v)
    assert c["r_eq"] < 1e-9, c
    assert c["r_ineq"] < 1e-9, c
    assert c["r_stat"] < tol_stat, c
    assert c["min_lam"] >= 0.0, c
    return qp, v


def test_dims(cartpole):
    _, orc, _, _ = cartpole
    assert orc.dims() == (11, 504, 8)        # 11 = 10 inputs + 1 steady-state parameter


def test_kkt_certificate_on_closed_loop_states(cartpole):
    p, orc, _, _ = cartpole
    idx = np.arange(0, len(S), 7)
    sol = orc.solve(S[idx, :4], S[idx, 4:])
    assert np.all(sol["status"] == 0)
    for k, i in enumerate(idx):
        _certify(p, S[i, :4], S[i, 4:], sol, k)


def test_matches_numpy_interior_point(cartpole):
    """Same minimiser from the un-reduced numpy IPM (different linear algebra: full KKT
    solves on the sparse form, no null-space reduction, no refinement).  That method stops
    at a duality-gap tolerance, and for this cost (weights 1e-1 .. 5e6) a gap of 1e-10
    relative still leaves the flat input directions uncertain, hence the 2e-3 band on u
    and the tight band on the objective."""
    p, orc, _, _ = cartpole
    for i in (0, 40, 123, 301, 450):
        qp = qp_sparse.build_sparse_qp(p, S[i, :4], S[i, 4:])
        ref = ipm_numpy.solve_qp(qp["P"], qp["q"], qp["A"], qp["b"], qp["G"], qp["h"], tol=1e-10, max_iter=200)
        sol = orc.solve(S[i:i + 1, :4], S[i:i + 1, 4:])
        v = qp_sparse.pack(qp, sol["x_nom"][0], sol["u_nom"][0], sol["x_ss"][0], sol["u_ss"][0])
        f_ref, f_orc = qp_sparse.objective(qp, ref["v"]), qp_sparse.objective(qp, v)
        assert f_orc <= f_ref + 1e-9 * max(1.0, abs(f_ref))
        x_r, u_r, xs_r, _ = qp_sparse.unpack(qp, ref["v"])
        np.testing.assert_allclose(sol["x_ss"][0], xs_r, atol=1e-6)
        np.testing.assert_allclose(sol["u_nom"][0], u_r, atol=2e-3)


def test_fixture_regression(cartpole):
    """The committed golden outputs are what the oracle computes today."""
    _, orc, _, _ = cartpole
    gold = np.load(os.path.join(common.GOLDEN, "cartpole_N10_oracle.npz"))
    sol = orc.solve(S[:, :4], S[:, 4:])
    assert np.array_equal(sol["status"], gold["status"])
    np.testing.assert_allclose(sol["u_nom"], gold["u_nom"], atol=1e-9)
    np.testing.assert_allclose(sol["xu_ss"], gold["xu_ss"], atol=1e-9)


def test_thread_count_does_not_change_results(cartpole):
    _, orc, _, _ = cartpole
    a = orc.solve(S[:64, :4], S[:64, 4:], nthreads=1)
    b = orc.solve(S[:64, :4], S[:64, 4:], nthreads=4)
    assert np.array_equal(a["u_nom"], b["u_nom"])


def test_infeasible_and_unconstrained_cases(cartpole):
    p, orc, mpc, _ = cartpole
    # angle beyond the tightened bound (hx[2] ~ 0.0759): stage-0 state row violated, x_0 is fixed
    x_bad = np.array([[0.0, 0.0, 0.2, 0.0]])
    sol = orc.solve(x_bad, np.array([[0.5, 0, 0, 0]]))
    assert sol["status"][0] == 2 and np.all(np.isnan(sol["u_nom"]))
    # at the reference steady state the unconstrained minimiser is feasible: zero iterations, u = 0
    sol = orc.solve(np.array([[0.5, 0, 0, 0.0]]), np.array([[0.5, 0, 0, 0]]))
    assert sol["status"][0] == 0 and sol["iters"][0] == 0
    np.testing.assert_allclose(sol["u_nom"], 0, atol=1e-9)
    np.testing.assert_allclose(sol["x_ss"][0], [0.5, 0, 0, 0], atol=1e-9)
    # far-away initial position: stays feasible only if inside the N-step controllable set
    sol = orc.solve(np.array([[4.4, 0.0, 0.0, 0.0]]), np.array([[0.0, 0, 0, 0]]))
    assert sol["status"][0] in (0, 2)


def test_double_integrator_free_initial_state_vs_scipy(oracle_lib):
    """Config 1 (Example_of_Tube_Tracking_MPC.py:19-53, N=5): initial state is a decision
    variable constrained to x_k (+) Z.  Independent check with scipy SLSQP on the sparse form."""
    mpc, w = common.make_mpc("double_integrator", 5, False)
    p = mpc._problem_dict()
    orc = Oracle(p)
    assert orc.dims()[0] == 8        # 5 inputs + theta + x_0 (2)
    for x, r in ((np.array([1.0, 2.0]), np.array([5.0, 0.0])), (np.array([-3.0, 0.5]), np.array([-9.0, 0.0])),
                 (np.array([4.0, 0.0]), np.array([4.0, 0.0]))):
        sol = orc.solve(x[None], r[None])
        assert sol["status"][0] == 0
        qp, v = _certify(p, x, r, sol, 0)
        cons = [{"type": "eq", "fun": lambda z, qp=qp: qp["A"] @ z - qp["b"], "jac": lambda z, qp=qp: qp["A"]},
                {"type": "ineq", "fun": lambda z, qp=qp: qp["h"] - qp["G"] @ z, "jac": lambda z, qp=qp: -qp["G"]}]
        res = minimize(lambda z: 0.5 * z @ qp["P"] @ z + qp["q"] @ z, np.zeros(v.size), jac=lambda z: qp["P"] @ z + qp["q"],
                       constraints=cons, method="SLSQP", options={"ftol": 1e-13, "maxiter": 1000})
        assert res.success or res.status == 8      # 8: SLSQP stalls at its own line-search precision
        assert np.max(qp["G"] @ res.x - qp["h"]) < 1e-7 and np.max(np.abs(qp["A"] @ res.x - qp["b"])) < 1e-7
        f_orc = 0.5 * v @ qp["P"] @ v + qp["q"] @ v
        assert f_orc <= res.fun + 1e-7 * max(1.0, abs(res.fun))
        _, u_s, _, _ = qp_sparse.unpack(qp, res.x)
        np.testing.assert_allclose(sol["u_nom"][0], u_s, atol=1e-4)
    # SURVEY appendix D anchor (Darup sets there; Rakovic sets here give a nearby value)
    sol = orc.solve(np.array([[1.0, 2.0]]), np.array([[5.0, 0.0]]))
    assert abs(sol["u_nom"][0, 0, 0] - (-0.7372)) < 5e-3


def test_extended_variant_builds_and_solves(oracle_lib):
    """Packet-received problem (TubeTrackingMPC.py:253-299) with the literal terminal row (:293) kept on free
    auxiliaries (the fall-back when no projection is supplied)."""
    mpc, w = common.make_mpc("double_integrator", 5, True, extended=True)
    mpc._eliminate_auxiliaries = False
    p = mpc._problem_dict()
    assert p["extended"] == 1 and "HTP" not in p
    orc = Oracle(p)
    nv0, nc0, _ = orc.dims(0)
    nv1, nc1, _ = orc.dims(1)
    assert nv0 == 6 and nv1 == 5 + 1 + 2 + 3      # u, theta, x_0, (x_aux, u_aux)
    x = np.array([[0.5, 0.2], [0.5, 0.2]])
    r = np.array([[3.0, 0.0], [3.0, 0.0]])
    sol = orc.solve(x, r, variant=np.array([0, 1], dtype=np.uint8))
    assert np.all(sol["status"] == 0)
    # variant 0 has x_0 = x_k; variant 1 may move x_0 inside x_k (+) (Z (-) W)
    np.testing.assert_allclose(sol["x_nom0"][0], x[0], atol=1e-12)
    ZmW = mpc._ZmW
    assert ZmW.contains(x[1] - sol["x_nom0"][1], 1e-9)
    qp = qp_sparse.build_sparse_qp(p, x[1], r[1], 1)
    # feasibility of the returned part w.r.t. every row that does not involve the auxiliaries
    v = qp_sparse.pack(qp, sol["x_nom"][1], sol["u_nom"][1], sol["x_ss"][1], sol["u_ss"][1])
    L = qp["layout"]
    rows = np.flatnonzero(np.abs(qp["G"][:, L.oxa:]).sum(axis=1) == 0)
    assert np.max(qp["G"][rows] @ v - qp["h"][rows]) < 1e-9
    assert np.max(np.abs(qp["A"] @ v - qp["b"])) < 1e-9


def test_extended_variant_with_auxiliaries_eliminated(oracle_lib):
    """Default form of the packet-received problem: line :293's free auxiliaries projected out at set-up
    (include/tmpc.h: HTP).  Same minimiser as the literal form, certified on the un-condensed QP."""
    from scipy.optimize import linprog
    mpc, w = common.make_mpc("double_integrator", 5, True, extended=True)
    p = mpc._problem_dict()
    assert p["HTP"].shape[1] == 3
    orc = Oracle(p)
    assert orc.dims(1)[0] == 5 + 1 + 2            # u, theta, x_0 -- no auxiliaries
    mpc2, _ = common.make_mpc("double_integrator", 5, True, extended=True)
    mpc2._eliminate_auxiliaries = False
    p_lit = mpc2._problem_dict()
    orc_lit = Oracle(p_lit)
    rng = np.random.default_rng(3)
    X = rng.uniform(-1.5, 1.5, (24, 2))
    R = np.c_[rng.uniform(-6, 6, 24), np.zeros(24)]
    one = np.ones(24, dtype=np.uint8)
    a, b = orc.solve(X, R, one), orc_lit.solve(X, R, one)
    ok = (a["status"] == 0) & (b["status"] == 0)
    assert ok.sum() >= 12 and np.array_equal(a["status"] >= 2, b["status"] >= 2)
    np.testing.assert_allclose(a["u_nom"][ok], b["u_nom"][ok], atol=2e-5, rtol=0)     # the literal form is only this accurate
    HT, hT = p["HT"], p["hT"]
    for i in np.flatnonzero(ok)[:8]:
        qp = qp_sparse.build_sparse_qp(p, X[i], R[i], 1)
        v = qp_sparse.pack(qp, a["x_nom"][i], a["u_nom"][i], a["x_ss"][i], a["u_ss"][i])
        c = qp_sparse.kkt_certificate(qp, v)
        assert c["r_eq"] < 1e-9 and c["r_ineq"] < 1e-9 and c["r_stat"] < 1e-8 and c["min_lam"] >= 0, c
        # the literal rows are satisfiable for this x_bar: an auxiliary pair exists (LP feasibility)
        res = linprog(np.zeros(3), A_ub=np.c_[HT[:, :2], HT[:, 4:]], b_ub=hT - HT[:, 2:4] @ a["x_ss"][i] + 1e-9,
                      bounds=(None, None), method="highs")
        assert res.status == 0


def test_minimiser_distance_measures_u0_shift():
    """qp_sparse.minimiser_distance is the entry-wise distance to the exact minimiser on the certified active set.  The
    residual-based KKT certificate normalises by |q| (1e6 for the cart-pole, T = 10 P) and lets a 1e-6 shift of u_0 through
    (instance 123 of the fixture: r_stat 5.7e-9 < 1e-7); the distance reads 1.0e-6 for that very point."""
    import os
    import common
    from oracle import qp_sparse
    from oracle.oracle import Oracle
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
    mpc, _ = common.make_mpc("cartpole", 10, True, create=False)
    p = mpc._problem_dict()
    idx = np.r_[123, np.arange(0, 600, 12)]
    out = Oracle(p).solve(S[idx, :4], S[idx, 4:])
    tpl = qp_sparse.SparseTemplate(p, 0)
    A, B = np.asarray(p["A"]), np.asarray(p["B"])
    worst = 0.0
    for j, k in enumerate(idx):
        assert out["status"][j] == 0
        qp = tpl.instance(S[k, :4], S[k, 4:])
        v = qp_sparse.pack(qp, out["x_nom"][j], out["u_nom"][j], out["x_ss"][j], out["u_ss"][j])
        d = qp_sparse.minimiser_distance(qp, v)
        assert d["certified"] and d["du0"] < 1e-9, (k, d["du0"], d["certified"])
        worst = max(worst, d["du0"])
    # the probe: u_0 moved by 1e-6, trajectory re-simulated (equalities hold), everything else as returned
    qp = tpl.instance(S[123, :4], S[123, 4:])
    u = out["u_nom"][0].copy()
    u[0] += 1e-6
    x = [S[123, :4]]
    for i in range(10):
        x.append(A @ x[-1] + B @ u[i])
    v2 = qp_sparse.pack(qp, np.array(x), u, out["x_ss"][0], out["u_ss"][0])
    c2 = qp_sparse.kkt_certificate_fast(qp, v2)
    d2 = qp_sparse.minimiser_distance(qp, v2)
    assert c2["r_stat"] < 1e-7                    # the residual certificate does not see it
    assert d2["certified"] and abs(d2["du0"] - 1e-6) < 1e-8, d2["du0"]      # the distance does


def test_hard_packet_received_states_certify(oracle_lib):
    """tests/golden/cartpole_N20_extended_hard_states.npy: the eleven instances (x_hat, ref, gamma) of a 65536-instance
    extended closed-loop batch at N = 20 (tests/test_full_size.py wrote them out) that round 2's refinement left uncertified
    (TMPC_STATUS_MAX_ITER with the exact minimiser in hand): degenerate vertices of the packet-received problem with
    nearly parallel active facets of the 854-row initial-state set.  With the repeated Newton steps on an unchanged working
    set and the refinement at the stalled-gap exit they certify; the exact distance certificate agrees."""
    import os
    import common
    from LinearMPCOverNetworks import polytope_lite
    from oracle import qp_sparse
    from oracle.oracle import Oracle
    D = np.load(os.path.join(common.GOLDEN, "cartpole_N20_extended_hard_states.npy"))
    X, R, G = D[:, :4], D[:, 4:8], D[:, 8].astype(np.uint8)
    mpc, _ = common.make_mpc("cartpole", 20, True, extended=True, create=False)
    polytope_lite.set_lp_backend("scipy")
    p = mpc._problem_dict()
    o = Oracle(p).solve(X, R, G)
    assert np.all(o["status"] == 0), o["status"]
    tpl = {v: qp_sparse.SparseTemplate(p, v) for v in (0, 1)}
    for k in range(len(X)):
        qp = tpl[int(G[k])].instance(X[k], R[k])
        v = qp_sparse.pack(qp, o["x_nom"][k], o["u_nom"][k], o["x_ss"][k], o["u_ss"][k])
        d = qp_sparse.minimiser_distance(qp, v)
        assert d["certified"] and d["du0"] < 1e-9, (k, d["du0"])
