"""Closed loop over a lossy network (results_linear_system.py:209-291) driven by the batched state
machines.  The CPU test uses the oracle as the solver (tests may); the GPU test uses the product
path and compares trajectories with the oracle-driven loop."""
import os

import numpy as np
import pytest

import common
from LinearMPCOverNetworks import montecarlo
from oracle.oracle import Oracle


def _oracle_packets(mpc, orc):
    def fn(x_hat, r, gamma=None):
        sol = orc.solve(x_hat, r, gamma)
        u_ss = sol["u_ss"] + sol["x_ss"] @ mpc._K.T                        # TubeTrackingMPC.py:217
        U = np.concatenate([sol["u_nom"], u_ss[:, None, :]], axis=1).transpose(0, 2, 1)
        return np.ascontiguousarray(U), sol["x_nom0"], sol["status"]
    return fn


def _setup(nb, T, seed=7, extended=False):
    mpc, w = common.make_mpc("cartpole", 10, True, extended=extended)
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=seed)
    ref = 0.5 * np.ones(T)
    return mpc, w, p_loss, th, ga, dist, ref


def test_closed_loop_invariants_cpu(oracle_lib):
    nb, T = 8, 40
    mpc, w, p_loss, th, ga, dist, ref = _setup(nb, T)
    orc = Oracle(mpc._problem_dict())
    K = mpc.get_steady_state_controller_gain()
    out = montecarlo.run_remote_tube_mpc(_oracle_packets(mpc, orc), w["A"], w["B"], K, mpc.get_ancillary_controller_gain(),
                                         10, mpc._Z, p_loss, ref, th, ga, dist)
    assert np.all(out["not_optimal"] == 0)
    assert np.all(out["tube_violations"] == 0)               # x_t - x_nom_t in Z: the paper's tube guarantee (:258)
    assert out["consistent_estimate_error"] < 1e-9           # Proposition 1
    assert np.all(np.isfinite(out["tracking_error"])) and out["tracking_error"].max() < 0.2
    # realisations do not depend on the sharding: trajectory 5 drawn alone equals row 5 of the batch
    th1, ga1, d1 = montecarlo.draw_realisations(1, T, w["w_bound"], seed=7, first=5)
    assert np.array_equal(th1[0], th[5]) and np.array_equal(d1[0], dist[5])


def test_extended_closed_loop_invariants_cpu(oracle_lib):
    """results_linear_system_with_extendedMPC.py:247-378: ExtendedTubeTrackingMPC + RobustEstimator +
    ConsistentActuator(is_extended_MPC_used=True); the controller is told gamma_{t-1}."""
    nb, T = 8, 40
    mpc, w, p_loss, th, ga, dist, ref = _setup(nb, T, extended=True)
    orc = Oracle(mpc._problem_dict())
    K = mpc.get_steady_state_controller_gain()
    seen = []

    def fn(x_hat, r, gamma):
        seen.append(gamma.copy())
        return _oracle_packets(mpc, orc)(x_hat, r, gamma)
    out = montecarlo.run_remote_tube_mpc(fn, w["A"], w["B"], K, mpc.get_ancillary_controller_gain(), 10, mpc._Z, p_loss, ref,
                                         th, ga, dist, extended=True)
    assert np.all(out["not_optimal"] == 0)
    assert np.all(out["tube_violations"] == 0)
    assert np.all(np.isfinite(out["tracking_error"])) and out["tracking_error"].max() < 0.2
    # gamma handed to the controller at step t is the plant-packet arrival of step t-1 (first step: 1)
    assert np.all(seen[0] == 1)
    g1 = np.where(ga[:, 1] < p_loss, 0, 1)
    assert np.array_equal(seen[2], g1.astype(np.uint8)) and np.all(seen[1] == 1)
    # loss-free trajectories: both problems are used and the packet-received one dominates
    assert np.all(np.array(seen)[:, p_loss == 0.0] == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("extended", [False, True])
def test_closed_loop_gpu_matches_oracle_loop(hip_lib, oracle_lib, extended):
    nb, T = 64, 60
    mpc, w, p_loss, th, ga, dist, ref = _setup(nb, T, seed=11, extended=extended)
    mpc_gpu, _ = common.make_mpc("cartpole", 10, True, extended=extended, create=True)
    orc = Oracle(mpc._problem_dict())
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    a = montecarlo.run_remote_tube_mpc(mpc_gpu.determine_packets, w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist,
                                       extended=extended)
    b = montecarlo.run_remote_tube_mpc(_oracle_packets(mpc, orc), w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist,
                                       extended=extended)
    assert np.all(a["not_optimal"] == 0) and np.all(a["tube_violations"] == 0)
    np.testing.assert_allclose(a["x_final"], b["x_final"], atol=1e-7, rtol=0)
    np.testing.assert_allclose(a["tracking_error"], b["tracking_error"], atol=1e-9, rtol=0)
    if not extended:           # Proposition 1 is about the plain remote tube MPC; the robust estimator tracks the plant state
        assert a["consistent_estimate_error"] < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("extended,N", [(False, 10), (True, 10), (False, 20), (True, 20)])
def test_device_loop_at_p_loss_09_plays_the_buffered_tail_like_the_oracle_loop(hip_lib, oracle_lib, extended, N):
    """Nine packets in ten are lost in both directions: the actuator plays u_1 .. u_{N-1} of its buffered sequence and the
    terminal law past its end (SmartActuator.py:100-103), so the LATER inputs of every solve reach the plant.  The
    device-resident loop (device solver + device state machines) against the numpy loop driven by the CPU oracle, same
    realisations: final states to 1e-7, and the buffer really is played to its end."""
    nb, T = 48, 120
    mpc, w = common.make_mpc("cartpole", N, True, extended=extended)
    mpc_gpu, _ = common.make_mpc("cartpole", N, True, extended=extended, create=True)
    p_loss = np.full(nb, 0.9)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=90 + N)
    ref = np.where(np.arange(T) < T // 2, 0.5, -0.3)
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    orc = Oracle(mpc._problem_dict())
    seen_d = []
    host = montecarlo.run_remote_tube_mpc(_oracle_packets(mpc, orc), w["A"], w["B"], K, Kp, N, mpc._Z, p_loss, ref, th, ga, dist,
                                          extended=extended, observer=lambda t, st: seen_d.append(t - st["s"]))
    dev = mpc_gpu.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended)
    d = np.array(seen_d)
    assert d.max() >= N and (d >= 1).mean() > 0.7          # most steps play a later input, some run past the sequence
    assert np.all(dev["not_optimal"] == 0) and np.all(host["not_optimal"] == 0)
    assert np.array_equal(dev["tube_violations"], host["tube_violations"]) and np.all(dev["tube_violations"] == 0)
    err = float(np.max(np.abs(dev["x_final"] - host["x_final"])))
    print(f"p_loss 0.9, N = {N}, extended = {extended}: max |x_final(device loop) - x_final(oracle loop)| = {err:.2e}, "
          f"longest run on one buffered sequence {int(d.max())} steps")
    assert err <= 1e-7, err
    np.testing.assert_allclose(dev["tracking_error"], host["tracking_error"], atol=1e-9, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("extended", [False, True])
def test_device_resident_loop_equals_host_loop(hip_lib, extended):
    """tmpc_mc_run (state machines in HIP between the solve launches) against the numpy state machines driving
    the same GPU solver: same statistics, same final states."""
    nb, T = 96, 80
    mpc, w, p_loss, th, ga, dist, ref = _setup(nb, T, seed=23, extended=extended)
    mpc_gpu, _ = common.make_mpc("cartpole", 10, True, extended=extended, create=True)
    ref = np.where(np.arange(T) < T // 2, 0.5, -0.3)                       # a reference step on the way
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    host = montecarlo.run_remote_tube_mpc(mpc_gpu.determine_packets, w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist,
                                          extended=extended)
    dev = mpc_gpu.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended)
    assert np.array_equal(dev["not_optimal"], host["not_optimal"]) and np.all(dev["not_optimal"] == 0)
    assert np.array_equal(dev["tube_violations"], host["tube_violations"]) and np.all(dev["tube_violations"] == 0)
    np.testing.assert_allclose(dev["x_final"], host["x_final"], atol=1e-8, rtol=0)
    np.testing.assert_allclose(dev["tracking_error"], host["tracking_error"], atol=1e-10, rtol=0)
    if not extended:
        assert dev["consistent_estimate_error"] < 1e-9


def test_nonlinear_cartpole_linearises_to_the_reference_model():
    """The cart-pole ODE used as the nonlinear plant, held over one sampling period (RK4, 500 Hz), has exactly the
    (A, B) of results_linear_system.py:35-61 as its Jacobian at the upright equilibrium."""
    from LinearMPCOverNetworks import workloads
    w = workloads.cartpole()
    eps = 1e-6
    A_fd = np.array([(workloads.cartpole_step(np.eye(4)[i] * eps, np.zeros(())) - workloads.cartpole_step(-np.eye(4)[i] * eps, np.zeros(())))
                     / (2 * eps) for i in range(4)]).T
    B_fd = (workloads.cartpole_step(np.zeros(4), np.array(eps)) - workloads.cartpole_step(np.zeros(4), np.array(-eps))) / (2 * eps)
    np.testing.assert_allclose(A_fd, w["A"], atol=1e-9)
    np.testing.assert_allclose(B_fd, w["B"][:, 0], atol=1e-9)
    # energy-like sanity: without force and friction the upright equilibrium is a fixed point
    np.testing.assert_allclose(workloads.cartpole_step(np.zeros(4), np.zeros(())), 0.0, atol=0)


@pytest.mark.gpu
def test_device_loop_with_nonlinear_plant_equals_host_loop(hip_lib):
    """Remote tube MPC (linear model inside) driving the NONLINEAR cart-pole, device loop vs numpy loop."""
    nb, T = 64, 100
    mpc, w, p_loss, th, ga, dist, ref = _setup(nb, T, seed=31)
    mpc_gpu, _ = common.make_mpc("cartpole", 10, True, create=True)
    dist = 0.0 * dist                                      # the mismatch between model and plant is the disturbance
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    host = montecarlo.run_remote_tube_mpc(mpc_gpu.determine_packets, w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist,
                                          plant=montecarlo.plant_callable("cartpole"))
    dev = mpc_gpu.run_closed_loop(p_loss, ref, th, ga, dist, plant="cartpole")
    lin = mpc_gpu.run_closed_loop(p_loss, ref, th, ga, dist)             # back to the linear plant: the setting is per call
    assert np.all(dev["not_optimal"] == 0)
    np.testing.assert_allclose(dev["x_final"], host["x_final"], atol=1e-8, rtol=0)
    np.testing.assert_allclose(dev["tracking_error"], host["tracking_error"], atol=1e-10, rtol=0)
    # tracking error over the physics-rate trajectory (results_nonlinear_system.py:361: x_traj[:, 0:-1] at 500 Hz)
    np.testing.assert_allclose(dev["tracking_error_physics"], host["tracking_error_physics"], atol=1e-10, rtol=0)
    assert "tracking_error_physics" not in lin
    assert np.all(np.abs(dev["tracking_error_physics"] * np.sqrt(10.0) - dev["tracking_error"]) < 0.2 * dev["tracking_error"])
    assert np.array_equal(dev["tube_violations"], host["tube_violations"])
    # the nonlinear plant really is a different plant, and the loop still tracks the reference
    assert np.max(np.abs(dev["x_final"] - lin["x_final"])) > 1e-6
    assert np.all(dev["x_final"][:, 0] > 0.2) and np.max(np.abs(dev["x_final"][:, 2])) < 0.05     # 2 s in: on its way, pole upright


def test_reference_stream_order():
    """draw_realisations_reference_order consumes the three generators exactly like the reference's loops
    (results_linear_system.py:21-23, 209-233): scalar draws in (i, l_mc, t) order."""
    from LinearMPCOverNetworks import montecarlo
    p_list, n_mc, T = [0.0, 0.3, 0.6], 2, 5
    wb = np.array([0.1, 0.2, 0.3, 0.4])
    p, th, ga, w = montecarlo.draw_realisations_reference_order(p_list, n_mc, T, wb)
    rng_w, rng_gamma, rng_theta = np.random.default_rng(679), np.random.default_rng(347), np.random.default_rng(124)
    k = 0
    for i in range(len(p_list)):
        for l_mc in range(n_mc):
            assert p[k] == p_list[i]
            for t in range(T):
                if t == 0:
                    assert th[k, 0] == 1.0 and ga[k, 0] == 1.0
                else:
                    assert th[k, t] == rng_theta.uniform() and ga[k, t] == rng_gamma.uniform()
                wt = np.r_[rng_w.uniform(-wb[0], wb[0]), rng_w.uniform(-wb[1], wb[1]),
                           rng_w.uniform(-wb[2], wb[2]), rng_w.uniform(-wb[3], wb[3])]
                assert np.array_equal(w[k, t], wt)
            k += 1


@pytest.mark.gpu
@pytest.mark.parametrize("extended", [False, True])
def test_config4_sweep_on_device(hip_lib, extended):
    """BASELINE config 4 at reduced N_MC: the sweep of results_linear_system.py:147-301 -- p_loss = 0, 0.1 ... 0.9,
    T = 250 steps, N = 20, reference 0.5 -- through montecarlo.mc_sweep with the state machines on the device
    (tmpc_mc_run), against the same sweep with the numpy state machines around the same GPU solver.  Same table, row for
    row; no tube violation; every solve optimal (what the reference's experiment reports for the tube MPC)."""
    mpc, w = common.make_mpc("cartpole", 20, True, extended=extended, create=True)
    p_loss = np.arange(10) / 10.0                                          # results_linear_system.py:149
    n_mc, T = 6, 250
    dev, pi = montecarlo.mc_sweep(mpc, w, p_loss, n_mc, T, 0.5, extended=extended, on_device=True)
    host, pi2 = montecarlo.mc_sweep(mpc, w, p_loss, n_mc, T, 0.5, extended=extended, on_device=False)
    assert dev.shape == (60, 3) and np.array_equal(pi, pi2) and sorted(set(pi)) == list(range(10))
    assert np.all(dev[:, 1] == 0) and np.all(dev[:, 2] == 0)
    np.testing.assert_allclose(dev[:, 0], host[:, 0], atol=1e-10, rtol=0)
    assert np.array_equal(dev[:, 1:], host[:, 1:])
    # the loss-free trajectories track best, and losing 90 % of the packets still keeps the loop inside the tube
    assert dev[pi == 0, 0].mean() <= dev[pi == 9, 0].mean() + 1e-12
    # a sharded sweep (two halves, as two ranks would run it) gives the same rows
    lo, hi = montecarlo.shard_bounds(60, 1, 2)
    th, ga, wd = montecarlo.draw_realisations(hi - lo, T, w["w_bound"], first=lo)
    half = mpc.run_closed_loop(p_loss[pi[lo:hi]], np.full(T, 0.5), th, ga, wd, extended=extended)
    np.testing.assert_allclose(half["tracking_error"], dev[lo:hi, 0], atol=1e-12, rtol=0)


@pytest.mark.gpu
def test_reference_streams_replay_on_device(hip_lib):
    """The reference's own realisations -- generators 679 / 347 / 124 consumed in its loop order
    (results_linear_system.py:21-23, 209-233) -- for 10 loss rates x 4 runs x 250 steps, N = 20: device-resident loop
    against the host loop."""
    mpc, w = common.make_mpc("cartpole", 20, True, create=True)
    p_loss = np.arange(10) / 10.0
    n_mc, T = 4, 250
    pl, th, ga, wd = montecarlo.draw_realisations_reference_order(p_loss, n_mc, T, w["w_bound"])
    ref = np.full(T, 0.5)
    dev = mpc.run_closed_loop(pl, ref, th, ga, wd)
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    host = montecarlo.run_remote_tube_mpc(mpc.determine_packets, w["A"], w["B"], K, Kp, 20, mpc._Z, pl, ref, th, ga, wd)
    assert np.all(dev["not_optimal"] == 0) and np.all(dev["tube_violations"] == 0)
    assert np.array_equal(dev["tube_violations"], host["tube_violations"])
    np.testing.assert_allclose(dev["tracking_error"], host["tracking_error"], atol=1e-10, rtol=0)
    np.testing.assert_allclose(dev["x_final"], host["x_final"], atol=1e-8, rtol=0)
    assert dev["consistent_estimate_error"] < 1e-9                                  # Proposition 1 of the paper
    # p_loss = 0 never drops a packet (strict <, :218): every loss-free run of the replay tracks identically up to the disturbance
    assert np.ptp(dev["tracking_error"][:n_mc]) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("extended,N", [(False, 10), (False, 20), (True, 10)])
def test_warm_started_loop_equals_cold_loop(hip_lib, extended, N):
    """tmpc_mc_set_warm_start: every solve first tries the working set of the trajectory's previous solve in the exact
    refinement.  Accepted points are exact minimisers, rejected ones fall back to the cold start -- so the closed loop is the
    same to 1e-8, while the interior-point iterations per solve drop."""
    nb, T = 256, 120
    mpc, w = common.make_mpc("cartpole", N, True, extended=extended, create=True)
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=41)
    ref = np.where(np.arange(T) < T // 2, 0.5, -0.3)
    cold = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended)
    warm = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended, warm_start=True)
    assert np.all(cold["not_optimal"] == 0) and np.all(warm["not_optimal"] == 0)
    assert np.array_equal(cold["tube_violations"], warm["tube_violations"])
    np.testing.assert_allclose(warm["x_final"], cold["x_final"], atol=1e-8, rtol=0)
    np.testing.assert_allclose(warm["tracking_error"], cold["tracking_error"], atol=1e-10, rtol=0)
    print(f"mean interior-point iterations per solve: cold {cold['iters_mean']:.2f}, warm {warm['iters_mean']:.2f}")
    assert warm["iters_mean"] < 0.8 * cold["iters_mean"]        # (non-extended: about 0.25; the extended loop alternates between its two problems)
    # the setting is per call: the next cold call is cold again
    again = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended)
    assert np.array_equal(again["iters_sum"], cold["iters_sum"])


def test_nonlinear_plant_against_an_independent_integrator():
    """The cart-pole ODE of the product (workloads.cartpole_rhs / cartpole_step, mirrored on the device) against equations of
    motion written down here from the Lagrangian of a cart (mass M, friction b) with a pendulum (mass m, inertia I about its
    centre, centre at distance l, angle th from the upright position) and integrated by scipy's adaptive solver:
        L = 1/2 (M + m) p'^2 + m l p' th' cos th + 1/2 (I + m l^2) th'^2 - m g l cos th
    => (M + m) p'' + m l th'' cos th - m l th'^2 sin th = F - b p',   (I + m l^2) th'' + m l p'' cos th = m g l sin th."""
    from scipy.integrate import solve_ivp
    from LinearMPCOverNetworks import workloads
    par = workloads.CARTPOLE_PARAMS
    M, m, b, I, g, l = (par[k] for k in ("M", "m", "b", "I", "g", "l"))

    def rhs(t, y, F):
        p, v, th, om = y
        Mm = np.array([[M + m, m * l * np.cos(th)], [m * l * np.cos(th), I + m * l * l]])
        f = np.array([F - b * v + m * l * om * om * np.sin(th), m * g * l * np.sin(th)])
        acc = np.linalg.solve(Mm, f)
        return [v, acc[0], om, acc[1]]

    rng = np.random.default_rng(3)
    for _ in range(20):
        x0 = rng.uniform(-1, 1, 4) * [1.0, 1.0, 0.3, 1.0]
        F = rng.uniform(-10, 10)
        ref = solve_ivp(rhs, (0.0, 0.02), x0, args=(F,), rtol=1e-12, atol=1e-14).y[:, -1]
        got = workloads.cartpole_step(x0, np.array(F))                  # RK4, 10 substeps of 2 ms (the reference's physics rate)
        np.testing.assert_allclose(got, ref, atol=2e-9, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("extended", [False, True])
def test_sample_trajectory_capture(hip_lib, extended):
    """tmpc_mc_set_capture / tmpc_mc_get_capture: the sample run the scripts keep for their plots (x_traj, x_nom_traj,
    results_linear_system.py:298-301) from the device loop = the one recorded by the host loop; the recorded pair is what the
    tube statistic tests."""
    nb, T = 16, 90
    mpc, w = common.make_mpc("cartpole", 10, True, extended=extended, create=True)
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=3)
    ref = np.where(np.arange(T) < 45, 0.5, -0.2)
    k = min(5, nb - 1)                                                   # :298 l_mc == min(5, N_MC - 1)
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    host = montecarlo.run_remote_tube_mpc(mpc.determine_packets, w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist,
                                          extended=extended, capture=k)
    dev = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended, capture=k)
    for key in ("x_traj", "x_nom_traj", "u_traj"):
        np.testing.assert_allclose(dev[key], host[key], atol=1e-8, rtol=0, err_msg=key)
    assert np.all(mpc._Z.contains((dev["x_traj"] - dev["x_nom_traj"]).T, 1e-7))
    np.testing.assert_allclose(dev["x_traj"][0], 0.0, atol=0)
    again = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended)          # the setting is per call
    assert "x_traj" not in again


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["wave", "block"])
def test_per_solve_device_times(hip_lib, path):
    """tmpc_set_solve_timing: the computational times the reference's controllers keep per solve (TubeTrackingMPC.py:205,
    242-243) and its scripts summarise (results_linear_system.py:305-315), taken per instance on the device.  Timing must not
    change any result; an instance that takes interior-point iterations takes longer than one the unconstrained minimiser
    solves; a trajectory's maximum is at least its mean."""
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))[:256]
    mpc, w = common.make_mpc("cartpole", 10, True, create=True)
    mpc.set_kernel_path(path)
    X = np.r_[S[:, :4], np.zeros((8, 4))]                       # eight instances at the origin tracking 0: no active row
    R = np.r_[S[:, 4:], np.zeros((8, 4))]
    plain = mpc._solve(X, R)
    timed = mpc._solve(X, R, timing=True)
    for key in ("u_nom", "x_nom0", "status", "iters"):
        np.testing.assert_array_equal(timed[key], plain[key], err_msg=key)
    t = timed["solve_time"]
    assert t.shape == (len(X),) and np.all(t > 0) and np.all(t < 1.0)
    assert np.all(timed["iters"][-8:] == 0)
    assert np.median(t[:-8][timed["iters"][:-8] > 0]) > 2 * np.median(t[-8:])
    assert "solve_time" not in mpc._solve(X, R)                  # the setting is per call
    if path == "wave":
        nb, T = 32, 40
        p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
        th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=5)
        ref = np.full(T, 0.4)
        a = mpc.run_closed_loop(p_loss, ref, th, ga, dist)
        b = mpc.run_closed_loop(p_loss, ref, th, ga, dist, timing=True)
        for key in ("err2", "tube_violations", "not_optimal", "x_final", "iters_sum"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)
        assert np.all(b["solve_time_mean"] > 0) and np.all(b["solve_time_max"] >= b["solve_time_mean"])
        assert len(mpc.get_computational_times()) >= nb
        table, _ = montecarlo.mc_sweep(mpc, w, np.array([0.0, 0.5]), 4, 30, 0.3, on_device=True, timing=True)
        assert table.shape == (8, 5) and np.all(table[:, 3] > 0) and np.all(table[:, 4] >= table[:, 3])


def test_nonlinear_scenario_stream_order():
    """scripts/mc_nonlinear_system.py draws its loss realisations with draw_realisations_reference_order(..., seeds=(1, 3467,
    124)): the reference's nonlinear experiment has its own generators -- gamma 3467, theta 124
    (results_nonlinear_system.py:25-26) -- consumed once per CONTROL step t >= 1, theta first (:264-273), over (loss rate,
    run) in loop order; no disturbance is injected (the linearisation error is the disturbance)."""
    from LinearMPCOverNetworks import montecarlo
    p_list, n_mc, T = [0.0, 0.2, 0.9], 3, 7
    p, th, ga, w = montecarlo.draw_realisations_reference_order(p_list, n_mc, T, np.zeros(4), seeds=(1, 3467, 124))
    rng_gamma, rng_theta = np.random.default_rng(3467), np.random.default_rng(124)
    k = 0
    for i in range(len(p_list)):
        for l_mc in range(n_mc):
            assert p[k] == p_list[i]
            assert th[k, 0] == 1.0 and ga[k, 0] == 1.0                 # first transmission always succeeds (:258-261)
            for t in range(1, T):
                assert th[k, t] == rng_theta.uniform() and ga[k, t] == rng_gamma.uniform()
            k += 1
    assert np.all(w == 0.0)


@pytest.mark.gpu
def test_nonlinear_script_smoke(hip_lib):
    """scripts/mc_nonlinear_system.py end to end at N_MC = 2: linear controllers on the nonlinear cart-pole (RK4 at 500 Hz on
    the device), tube MPC and tracking MPC on the same realisations; the table it prints must show the tube MPC inside its
    tube with every solve optimal, and a finite physics-rate tracking error for every loss rate."""
    import subprocess
    import sys
    root = os.path.dirname(common.PKG)
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "mc_nonlinear_system.py"), "--n-mc", "2"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    rows = [l.replace("|", " ").split() for l in res.stdout.splitlines() if l[:1].isspace() and l.strip()[:1].isdigit()]
    assert len(rows) == 10, res.stdout
    for r in rows:
        e500, e50, outside, nonopt = float(r[1]), float(r[2]), int(r[3]), int(r[4])
        assert np.isfinite(e500) and 0.0 < e500 < 0.1 and 0.0 < e50 < 0.2
        assert nonopt == 0
    assert sum(int(r[3]) for r in rows) <= 2          # the linearisation error is not bounded by the design's W: rare excursions are reported, not hidden
