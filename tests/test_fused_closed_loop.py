"""The fused closed loop (include/tmpc.h: tmpc_mc_set_fused; csrc/tmpc_fused.hip: closed_loop_kernel) against the launch pair per time
step.  A trajectory's arithmetic does not depend on which wavefront runs it or when, so ONE launch in which a wavefront keeps its
trajectory for all T steps -- the body of the reference's loop, results_linear_system.py:209-259, with nothing between two of its
iterations -- must give the numbers of T solve launches + T state-machine launches BIT FOR BIT: statistics, final states, iteration
sums, the captured sample run, dead R-MPC trajectories (:262-287) and the physics-rate error of the nonlinear plant."""
import numpy as np
import pytest

import common
from LinearMPCOverNetworks import montecarlo, workloads
from LinearMPCOverNetworks.TrackingMPC import TrackingMPC

pytestmark = pytest.mark.gpu

KEYS = ("err2", "tube_violations", "not_optimal", "x_final", "consistent", "iters_sum")


def _same(a, b, keys=KEYS):
    for k in keys:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), k


@pytest.mark.parametrize("name,N,fixed,warm", [("cartpole", 10, True, False), ("cartpole", 10, True, True), ("cartpole", 20, True, False),
                                               ("double_integrator", 10, False, False), ("double_integrator", 5, False, True)])
def test_fused_equals_per_step_bit_for_bit(hip_lib, name, N, fixed, warm):
    nb, T = 200, 60                                            # (the reference's own experiment size: 10 x 20 trajectories)
    mpc, w = common.make_mpc(name, N, fixed, create=True)
    p_loss = np.tile(np.arange(10) / 10.0, nb // 10)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=41)
    amp = 0.5 if name == "cartpole" else 4.0
    ref = np.where(np.arange(T) < T // 2, amp, -0.6 * amp)
    off = mpc.run_closed_loop(p_loss, ref, th, ga, dist, warm_start=warm, capture=7, fused="off")
    on = mpc.run_closed_loop(p_loss, ref, th, ga, dist, warm_start=warm, capture=7, fused="on")
    assert on["fused"] and not off["fused"]
    _same(on, off, KEYS + ("x_traj", "x_nom_traj", "u_traj"))
    assert np.all(on["not_optimal"] == 0) and on["iters_mean"] > 0.5
    # the realisations drawn on the device (Philox per trajectory and step) and per-solve timing switched on: same again
    rng = (5, 1000, w["w_bound"])
    off = mpc.run_closed_loop(p_loss, ref, device_rng=rng, warm_start=warm, timing=True, fused="off")
    on = mpc.run_closed_loop(p_loss, ref, device_rng=rng, warm_start=warm, timing=True, fused="on")
    assert on["fused"] and not off["fused"]
    _same(on, off)
    assert np.all(on["solve_time_mean"] > 0) and np.all(on["solve_time_max"] >= on["solve_time_mean"])


def test_fused_rmpc_loop_with_dead_trajectories(hip_lib):
    """TrackingMPC + plain SmartActuator: a trajectory whose solve is infeasible stops (results_linear_system.py:268-270) -- inside
    the fused kernel its wavefront leaves the time loop and draws the next trajectory."""
    w = workloads.double_integrator()
    mpc = TrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    mpc._Xc, mpc._Uc = mpc._X, mpc._U
    mpc.determine_Xf(verbose=False)
    mpc._fixed_initial_state = True
    mpc.generate_optimization_problem()
    nb, T = 96, 60
    rng = np.random.default_rng(11)
    x0 = rng.uniform(-1, 1, (nb, 2)) * [7.6, 0.6]
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, 3.0 * w["w_bound"], seed=5)
    ref = np.where(np.arange(T) < 30, 6.0, -6.0)
    off = mpc.run_closed_loop(p_loss, ref, th, ga, dist, x0=x0, fused="off")
    on = mpc.run_closed_loop(p_loss, ref, th, ga, dist, x0=x0, fused="on")
    assert on["fused"] and not off["fused"]
    dead = np.isnan(on["tracking_error"])
    assert 0 < dead.sum() < nb
    _same(on, off)


def test_fused_loop_with_the_nonlinear_plant(hip_lib):
    nb, T = 64, 50
    mpc, w = common.make_mpc("cartpole", 10, True, create=True)
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=31)
    ref = 0.5 * np.ones(T)
    off = mpc.run_closed_loop(p_loss, ref, th, ga, 0.0 * dist, plant="cartpole", fused="off")
    on = mpc.run_closed_loop(p_loss, ref, th, ga, 0.0 * dist, plant="cartpole", fused="on")
    assert on["fused"] and not off["fused"]
    _same(on, off, KEYS + ("err2_physics",))


def test_automatic_choice(hip_lib):
    """TMPC_MC_FUSED_AUTO: one round of trajectories, or rounds filled to 85 %, run fused; a batch a little above a multiple of the
    resident wavefronts, the extended controller (two problems), the workgroup-per-QP kernel and explicit "off" step per launch."""
    T = 3
    mpc, w = common.make_mpc("cartpole", 10, True, create=True)
    ref = 0.2 * np.ones(T)

    def run(B, **kw):
        return mpc.run_closed_loop(np.full(B, 0.3), ref, device_rng=(1, 0, w["w_bound"]), **kw)["fused"]
    assert run(64) and run(2048) and run(4096)
    assert not run(2048 + 256)                    # 256 CUs x 8 wavefronts resident: the second round would be an eighth full
    assert not run(64, fused="off") and run(2048 + 256, fused="on")
    mpc.set_kernel_path("block")
    assert not run(64) and not run(64, fused="on")
    mpc.set_kernel_path("wave")
    assert run(64)
    ext, _ = common.make_mpc("cartpole", 10, True, extended=True, create=True)
    e = ext.run_closed_loop(np.full(64, 0.3), ref, device_rng=(1, 0, w["w_bound"]), extended=True, fused="on")
    assert not e["fused"] and e["loop_mode"] == 2       # two problems: a launch per problem and step, the state machines inside
    assert ext.run_closed_loop(np.full(64, 0.3), ref, device_rng=(1, 0, w["w_bound"]), extended=True, fused="off")["loop_mode"] == 0
    # automatic: from one round of resident wavefronts on (below that the launches' latency decides, and three short ones win)
    assert ext.run_closed_loop(np.full(64, 0.3), ref, device_rng=(1, 0, w["w_bound"]), extended=True)["loop_mode"] == 0
    assert ext.run_closed_loop(np.full(2048, 0.3), ref, device_rng=(1, 0, w["w_bound"]), extended=True)["loop_mode"] == 2


@pytest.mark.parametrize("N,warm", [(10, False), (10, True), (20, True)])
def test_extended_loop_with_the_state_machines_inside_the_solve_launches(hip_lib, N, warm):
    """The extended controller changes its QP from step to step with the arrival flag (results_linear_system_with_extendedMPC.py:
    267-279): closed_loop_step_kernel<shape of the problem> solves the trajectories whose flag selects its problem and runs their
    state machines; the flags a step writes are the selector of the NEXT step (two buffers).  Same numbers as two solve launches
    + one state-machine launch per step, bit for bit."""
    nb, T = 200, 60
    mpc, w = common.make_mpc("cartpole", N, True, extended=True, create=True)
    p_loss = np.tile(np.arange(10) / 10.0, nb // 10)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=43)
    ref = np.where(np.arange(T) < T // 2, 0.5, -0.3)
    off = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=True, warm_start=warm, capture=3, timing=True, fused="off")
    on = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=True, warm_start=warm, capture=3, timing=True, fused="on")
    assert off["loop_mode"] == 0 and on["loop_mode"] == 2
    _same(on, off, KEYS + ("x_traj", "x_nom_traj", "u_traj"))
    assert np.all(on["not_optimal"] == 0) and np.all(on["tube_violations"] == 0)
    assert np.all(on["solve_time_mean"] > 0)
