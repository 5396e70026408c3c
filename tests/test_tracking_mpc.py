"""R-MPC comparator (reference TrackingMPC.py) on the device kernels."""
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
    # terminal-set requirement of this implementation
    bare = TrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
    with pytest.raises(NotImplementedError):
        bare.generate_optimization_problem()


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
