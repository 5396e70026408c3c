"""Realisations drawn on the device (tmpc_mc_set_device_rng, SURVEY.md 8(d): "Philox on device for throughput runs").
The generator is pinned on the CPU against numpy's own Philox; on the GPU the closed loop fed by the device generator must
equal, bit for bit, the closed loop fed with the host twin's arrays."""
import numpy as np
import pytest

import common
from LinearMPCOverNetworks import montecarlo


def test_philox_twin_against_numpy():
    rng = np.random.default_rng(0)
    for _ in range(20):
        key = [int(v) for v in rng.integers(0, 2 ** 63, 2, dtype=np.uint64) * 2 + rng.integers(0, 2, 2, dtype=np.uint64)]
        c0, c1 = int(rng.integers(1, 2 ** 62)), int(rng.integers(0, 8))
        # numpy increments the counter before it generates: counter c0 - 1 yields the block of c0
        want = np.random.Philox(key=np.array(key, dtype=np.uint64), counter=[c0 - 1, c1, 0, 0]).random_raw(4)   # (a list of large ints would pass through float64)
        got = montecarlo.philox4x64(c0, c1, key[0], key[1]).reshape(4)
        assert np.array_equal(want, got)
    # vectorised call = element-wise calls
    t = np.arange(5, dtype=np.uint64)[None, :]
    g = np.arange(3, dtype=np.uint64)[:, None] + np.uint64(40)
    blk = montecarlo.philox4x64(t, np.uint64(1), np.uint64(9), g)
    assert blk.shape == (4, 3, 5)
    assert np.array_equal(blk[:, 2, 4], montecarlo.philox4x64(4, 1, 9, 42).reshape(4))


def test_philox_realisations_layout_and_sharding():
    wb = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7])              # nx = 7: three blocks per step
    th, ga, w = montecarlo.draw_realisations_philox(6, 9, wb, seed=77, first=0)
    assert th.shape == (6, 9) and ga.shape == (6, 9) and w.shape == (6, 9, 7)
    assert np.all((th >= 0) & (th < 1)) and np.all((ga >= 0) & (ga < 1)) and np.all(np.abs(w) <= wb)
    # numpy's generator over the same key / counter: trajectory 4, step 3, block 0 then block 1
    u0 = np.random.Generator(np.random.Philox(key=[77, 4], counter=[2, 0, 0, 0])).random(4)
    u1 = np.random.Generator(np.random.Philox(key=[77, 4], counter=[2, 1, 0, 0])).random(4)
    assert th[4, 3] == u0[0] and ga[4, 3] == u0[1]
    np.testing.assert_allclose(w[4, 3, :6], wb[:6] * (2 * np.r_[u0[2:], u1] - 1), rtol=0, atol=0)
    # a trajectory's stream depends on its global index only
    th2, ga2, w2 = montecarlo.draw_realisations_philox(3, 9, wb, seed=77, first=3)
    assert np.array_equal(th2, th[3:]) and np.array_equal(ga2, ga[3:]) and np.array_equal(w2, w[3:])
    # and looks uniform
    thL, _, wL = montecarlo.draw_realisations_philox(400, 250, wb[:4], seed=1)
    assert abs(thL.mean() - 0.5) < 5e-3 and abs(np.mean(thL < 0.3) - 0.3) < 5e-3
    assert np.all(np.abs(wL.mean(axis=(0, 1))) < 0.01 * wb[:4])


@pytest.mark.gpu
@pytest.mark.parametrize("extended", [False, True])
def test_device_generator_equals_host_twin(hip_lib, extended):
    nb, T, seed, first = 96, 60, 4242, 1000
    mpc, w = common.make_mpc("cartpole", 10, True, extended=extended, create=True)
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    ref = np.where(np.arange(T) < 30, 0.5, -0.3)
    th, ga, dist = montecarlo.draw_realisations_philox(nb, T, w["w_bound"], seed=seed, first=first)
    host_arrays = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=extended)
    on_device = mpc.run_closed_loop(p_loss, ref, extended=extended, device_rng=(seed, first, w["w_bound"]))
    for key in ("err2", "tube_violations", "not_optimal", "x_final", "iters_sum"):
        np.testing.assert_array_equal(on_device[key], host_arrays[key], err_msg=key)
    assert np.all(on_device["tube_violations"] == 0) and np.all(on_device["not_optimal"] == 0)
    # the setting is per call: the next call with arrays uses the arrays
    again = mpc.run_closed_loop(p_loss, ref, th * 0 + 1.0, ga * 0 + 1.0, dist * 0, extended=extended)
    assert not np.array_equal(again["x_final"], on_device["x_final"])
    # a shard of the same sweep: trajectories first + 32 .. first + 64
    part = mpc.run_closed_loop(p_loss[32:64], ref, extended=extended, device_rng=(seed, first + 32, w["w_bound"]))
    np.testing.assert_allclose(part["err2"], on_device["err2"][32:64], rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_sweep_with_device_generator(hip_lib):
    """mc_sweep(device_rng=True): device loop with device draws = host loop with the twin's draws."""
    mpc, w = common.make_mpc("cartpole", 10, True, create=True)
    p_loss = np.array([0.0, 0.4, 0.8])
    dev, pi = montecarlo.mc_sweep(mpc, w, p_loss, 6, 80, 0.5, seed=99, on_device=True, device_rng=True)
    host, _ = montecarlo.mc_sweep(mpc, w, p_loss, 6, 80, 0.5, seed=99, on_device=False, device_rng=True)
    np.testing.assert_allclose(dev[:, 0], host[:, 0], rtol=0, atol=1e-9)
    assert np.array_equal(dev[:, 1:], host[:, 1:]) and np.all(dev[:, 1] == 0)


@pytest.mark.gpu
def test_device_generator_other_state_dimension(hip_lib):
    """nx = 2 (double integrator, free initial state): one Philox block per step carries theta, gamma and both disturbances."""
    nb, T, seed = 64, 40, 7
    mpc, w = common.make_mpc("double_integrator", 5, False, create=True)
    p_loss = np.tile([0.0, 0.5], nb // 2)
    ref = np.full(T, 2.0)
    th, ga, dist = montecarlo.draw_realisations_philox(nb, T, w["w_bound"], seed=seed, first=0)
    a = mpc.run_closed_loop(p_loss, ref, th, ga, dist)
    b = mpc.run_closed_loop(p_loss, ref, device_rng=(seed, 0, w["w_bound"]))
    for key in ("err2", "tube_violations", "not_optimal", "x_final"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert np.all(b["tube_violations"] == 0)
