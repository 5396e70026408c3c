"""R-MPC comparator (reference TrackingMPC.py) on the device kernels."""
import os

import numpy as np
import pytest

import common
from LinearMPCOverNetworks import workloads
from LinearMPCOverNetworks.TrackingMPC import TrackingMPC
from oracle.oracle import Oracle


def _make(create):
    w = workloads.double_integrator()
    mpc = TrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    mpc._Xc, mpc._Uc = mpc._X, mpc._U
    mpc.determine_Xf(verbose=False)
    mpc._fixed_initial_state = True
    if create:
        mpc.generate_optimization_problem()
    return mpc, w


def test_problem_is_the_untightened_fixed_x0_qp(oracle_lib):
    mpc, w = _make(False)
    p = mpc._problem_dict()
    assert p["fixed_x0"] == 1 and "HZ" not in p
    assert np.array_equal(p["Hx"], w["X"].A) and np.array_equal(p["hu"], w["U"].b)       # TrackingMPC.py:94-97
    orc = Oracle(p)
    sol = orc.solve(np.array([[1.0, 2.0], [7.9, 0.9]]), np.array([[5.0, 0.0], [5.0, 0.0]]))
    assert sol["status"][0] == 0
    # x = (7.9, 0.9): the next state 8.8 leaves X whatever the bounded input does -> infeasible (U_t = None in the reference)
    assert sol["status"][1] == 2


def _bare(name, N):
    """TrackingMPC before setup_optimization(): no terminal set, x_N == x_bar instead (TrackingMPC.py:105-107)."""
    w = workloads.double_integrator() if name == "double_integrator" else workloads.cartpole()
    mpc = TrackingMPC(w["A"], w["B"], w["Q"], w["R"], N)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    mpc._Xc, mpc._Uc = mpc._X, mpc._U
    mpc._fixed_initial_state = True
    return mpc, w


def test_terminal_equality_in_the_oracle(oracle_lib):
    """The oracle's restatement of TrackingMPC.py:105-107 (x_N == x_bar, eliminated numerically with the other equalities):
    KKT-certified on the un-condensed QP, and the equality holds in the returned trajectory."""
    from oracle import qp_sparse
    mpc, w = _bare("double_integrator", 10)
    p = mpc._problem_dict()
    assert p["terminal_equality"] == 1 and "HT" not in p
    orc = Oracle(p)
    rng = np.random.default_rng(2)
    X = rng.uniform(-1, 1, (48, 2)) * [5.0, 0.8]
    R = np.c_[rng.uniform(-6, 6, 48), np.zeros(48)]
    sol = orc.solve(X, R)
    ok = sol["status"] == 0
    assert ok.sum() > 20
    np.testing.assert_allclose(sol["x_nom"][ok][:, -1], sol["x_ss"][ok], atol=1e-9)
    for k in np.flatnonzero(ok)[:24]:
        qp = qp_sparse.build_sparse_qp(p, X[k], R[k])
        v = qp_sparse.pack(qp, sol["x_nom"][k], sol["u_nom"][k], sol["x_ss"][k], sol["u_ss"][k])
        c = qp_sparse.kkt_certificate(qp, v)
        assert c["r_eq"] < 1e-9 and c["r_ineq"] < 1e-9 and c["r_stat"] < 1e-7 and c["min_lam"] > -1e-9, c


@pytest.mark.gpu
@pytest.mark.parametrize("name,N,path", [("double_integrator", 10, "wave"), ("double_integrator", 10, "block"),
                                         ("cartpole", 20, "auto")])
def test_terminal_equality_on_the_device(hip_lib, oracle_lib, name, N, path):
    """TrackingMPC without a terminal set on both kernels: statuses and minimisers of the oracle, x_N == x_bar, and the
    solver-independent KKT certificate on the un-condensed QP with the equality rows."""
    from oracle import qp_sparse
    mpc, w = _bare(name, N)
    mpc.generate_optimization_problem()
    mpc.set_kernel_path(path)
    p = mpc._problem_dict()
    orc = Oracle(p)
    rng = np.random.default_rng(4)
    nx = w["A"].shape[0]
    if name == "double_integrator":
        X = rng.uniform(-1, 1, (96, 2)) * [5.0, 0.8]
        R = np.c_[rng.uniform(-6, 6, 96), np.zeros(96)]
    else:
        S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
        idx = rng.choice(len(S), 96, replace=False)
        X, R = S[idx, :4] * 0.5, S[idx, 4:] * 0.3
    ref = orc.solve(X, R)
    x_mpc, u_mpc, x_bar, u_bar = mpc.solve_optimization_problem(X, R)
    assert np.array_equal(mpc.last_status, ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() > 20
    np.testing.assert_allclose(u_mpc[ok], ref["u_nom"][ok], atol=1e-8, rtol=0)
    np.testing.assert_allclose(x_bar[ok], ref["x_ss"][ok], atol=1e-9, rtol=0)
    np.testing.assert_allclose(x_mpc[ok][:, -1], x_bar[ok], atol=1e-9, rtol=0)             # x_N == x_bar
    for k in np.flatnonzero(ok)[:32]:
        qp = qp_sparse.build_sparse_qp(p, X[k], R[k])
        v = qp_sparse.pack(qp, x_mpc[k], u_mpc[k], x_bar[k], u_bar[k])
        c = qp_sparse.kkt_certificate_fast(qp, v)
        assert c["r_eq"] < 1e-9 and c["r_ineq"] < 1e-9 and c["r_stat"] < 1e-7 and c["min_lam"] > -1e-9, c
    pkt = mpc.determine_packet(X[np.flatnonzero(ok)[0]].copy(), R[np.flatnonzero(ok)[0]].copy(), 2)
    assert pkt["U_t"].shape == (w["B"].shape[1], N + 1)


@pytest.mark.gpu
def test_tracking_mpc_parity_and_packets(hip_lib, oracle_lib):
    mpc, w = _make(True)
    orc = Oracle(mpc._problem_dict())
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (128, 2)) * [6.0, 1.0]
    X[-4:] = [[7.9, 0.9], [-7.9, -0.9], [7.5, 1.0], [-7.99, -0.5]]      # the next state leaves X whatever u does
    R = np.c_[rng.uniform(-7, 7, 128), np.zeros(128)]
    ref = orc.solve(X, R)
    x_mpc, u_mpc, x_bar, u_bar = mpc.solve_optimization_problem(X, R)
    assert np.array_equal(mpc.last_status, ref["status"])
    ok = ref["status"] == 0
    assert 40 < ok.sum() and (~ok).sum() > 0
    np.testing.assert_allclose(u_mpc[ok], ref["u_nom"][ok], atol=1e-8, rtol=0)
    np.testing.assert_allclose(x_bar[ok], ref["x_ss"][ok], atol=1e-9, rtol=0)
    i, j = int(np.flatnonzero(ok)[0]), int(np.flatnonzero(~ok)[0])
    pkt = mpc.determine_packet(X[i].copy(), R[i].copy(), 3)
    assert pkt["U_t"].shape == (1, 11) and pkt["q_t"] == 3
    np.testing.assert_allclose(pkt["U_t"][0, :10], ref["u_nom"][i, :, 0], atol=1e-8)
    np.testing.assert_allclose(pkt["U_t"][0, 10], ref["u_ss"][i, 0] + (mpc._K @ ref["x_ss"][i])[0], atol=1e-8)
    assert mpc.determine_packet(X[j].copy(), R[j].copy(), 4)["U_t"] is None                 # infeasible -> None (results_linear_system.py:268)


@pytest.mark.gpu
def test_rmpc_closed_loop_device_equals_host(hip_lib):
    """The comparator's closed loop (TrackingMPC + Estimator + plain SmartActuator, results_linear_system.py:198-205,
    262-287) on the device against the numpy state machines; trajectories that become infeasible stop and report NaN."""
    from LinearMPCOverNetworks import montecarlo
    mpc, w = _make(True)
    nb, T = 96, 60
    rng = np.random.default_rng(11)
    x0 = rng.uniform(-1, 1, (nb, 2)) * [7.6, 0.6]
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, 3.0 * w["w_bound"], seed=5)       # a rough ride: some leave the feasible set
    ref = np.where(np.arange(T) < 30, 6.0, -6.0)
    host = montecarlo.run_remote_tracking_mpc(mpc.determine_packets, w["A"], w["B"], mpc.get_steady_state_controller_gain(), 10,
                                              p_loss, ref, th, ga, dist, x0=x0)
    dev = mpc.run_closed_loop(p_loss, ref, th, ga, dist, x0=x0)
    dead = np.isnan(dev["tracking_error"])
    assert np.array_equal(dead, host["infeasible"]) and 0 < dead.sum() < nb
    assert np.array_equal(dev["not_optimal"], host["not_optimal"])
    np.testing.assert_allclose(dev["tracking_error"][~dead], host["tracking_error"][~dead], atol=1e-10, rtol=0)
    np.testing.assert_allclose(dev["x_final"], host["x_final"], atol=1e-8, rtol=0)


@pytest.mark.gpu
def test_rmpc_cartpole_closed_loop_device_equals_host(hip_lib):
    """The R-MPC leg of the reference's cartpole experiment (results_linear_system.py:132-140, 198-205, 262-287): TrackingMPC with
    the UN-tightened cartpole sets, N = 20, over the lossy network with disturbances; device-resident loop against the numpy state
    machines around the same device solver, incl. the trajectories that become infeasible (track_feasible = False, :268-270)."""
    from LinearMPCOverNetworks import montecarlo
    mpc, w = workloads.make_controller("cartpole", 20, tracking=True)
    assert mpc._Xc is mpc._X and mpc.get_kernel_path() == "wave"
    nb, T = 120, 120
    p_loss = np.repeat(np.arange(10) / 10.0, nb // 10)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=17)
    ref = np.where(np.arange(T) < 60, 0.5, -0.5)
    host = montecarlo.run_remote_tracking_mpc(mpc.determine_packets, w["A"], w["B"], mpc.get_steady_state_controller_gain(), 20,
                                              p_loss, ref, th, ga, dist)
    dev = mpc.run_closed_loop(p_loss, ref, th, ga, dist)
    dead = np.isnan(dev["tracking_error"])
    assert np.array_equal(dead, host["infeasible"])
    assert np.array_equal(dev["not_optimal"], host["not_optimal"])
    np.testing.assert_allclose(dev["tracking_error"][~dead], host["tracking_error"][~dead], atol=1e-10, rtol=0)
    np.testing.assert_allclose(dev["x_final"][~dead], host["x_final"][~dead], atol=1e-8, rtol=0)
    assert (~dead).sum() >= nb // 2
    lossless = (p_loss == 0.0) & ~dead
    assert lossless.any() and np.all(np.abs(dev["x_final"][lossless][:, 0] + 0.5) < 0.3)               # loss-free runs are on their way to -0.5
