"""The DEVICE state machines (csrc/tmpc_mc.hip: mc_step_kernel, what tmpc_mc_run launches after every solve) against the
trajectories recorded from the REFERENCE's own numpy classes: Estimator / RobustEstimator (Estimator.py:43-161) and
SmartActuator / ConsistentActuator (SmartActuator.py:57-231), tests/golden/make_glue_golden.py.

tmpc_mc_replay feeds the recorded controller packets, arrival flags and disturbances to the kernel (no QP is solved) and
returns every step of every trajectory.  Integers (s_t, Theta_t, q_t) must agree exactly, floats to 1e-12 relative to the
trajectory's scale: the device forms the same sums as numpy's matrix products in a different order, and the recorded packets
are random sequences that do not stabilise the cart-pole (|x| grows to 1e2 .. 1e4 over the 60 steps)."""
import os

import numpy as np
import pytest

import common
from LinearMPCOverNetworks import _native

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(common.GOLDEN, "glue_golden.npz"))
GS = np.load(os.path.join(common.GOLDEN, "glue_smart_golden.npz"))
FLOAT_TOL = 1e-12


def _controller(extended, K_anc=None):
    mpc, w = common.make_mpc("cartpole", 10, True, extended=extended)
    if K_anc is not None:
        mpc.set_ancillary_controller_gain(K_anc)
    mpc.generate_optimization_problem(True)
    return mpc


def _stack(Gz, names, key):
    return np.stack([Gz[f"{n}/{key}"] for n in names], axis=0)          # (B, T, ...)


def _check(name, got, want, scale):
    err = float(np.max(np.abs(got - want) / scale))
    print(f"   {name:6s} max deviation / trajectory scale = {err:.2e}")
    assert err <= FLOAT_TOL, (name, err)
    return err


@pytest.mark.parametrize("ext", [0, 1])
def test_device_state_machines_replay_the_reference_recordings(ext):
    """12 cases per controller kind (4 loss rates x 3 seeds) as one batch: ConsistentActuator + Estimator (ext = 0),
    ConsistentActuator(is_extended_MPC_used) + RobustEstimator (ext = 1); ancillary gain 0.8 K as recorded."""
    names = [str(c) for c in G["cases"] if str(c).startswith(f"e{ext}")]
    assert len(names) == 12
    mpc = _controller(bool(ext), K_anc=G["Kp"])
    np.testing.assert_allclose(mpc.get_steady_state_controller_gain(), G["K"], rtol=1e-12)
    U = _stack(G, names, "U").transpose(0, 1, 3, 2)                     # (B, T, nu, N+1) -> (B, T, N+1, nu)
    out = _native.mc_replay(mpc._handle, U, _stack(G, names, "theta"), _stack(G, names, "gamma"), _stack(G, names, "wv"),
                            xn0=_stack(G, names, "xn0") if ext else None, extended=bool(ext))
    assert np.array_equal(out["s"], _stack(G, names, "s"))
    assert np.array_equal(out["Theta"], _stack(G, names, "Theta").astype(np.int32))
    assert np.array_equal(out["q"], _stack(G, names, "q"))
    X = _stack(G, names, "x")
    scale = np.maximum(np.abs(X).max(axis=(1, 2), keepdims=True), 1.0)  # per trajectory
    _check("x", out["x"], X, scale)
    _check("x_hat", out["x_hat"], _stack(G, names, "xhat"), scale)
    _check("x_nom", out["x_nom"], _stack(G, names, "xnom"), scale)
    _check("u", out["u"], _stack(G, names, "u"), np.maximum(np.abs(_stack(G, names, "u")).max(axis=(1, 2), keepdims=True), 1.0))
    # the loss patterns really exercise the buffer: at p = 0.9 the actuator plays past the end of its sequence
    d = np.arange(out["s"].shape[1])[None, :] - out["s"]
    assert d.max() >= 10 and (out["Theta"] == 0).any() and (out["Theta"] == 1).any()


def test_device_smart_actuator_replays_the_reference_recordings():
    """The R-MPC comparator's pair: plain SmartActuator (SmartActuator.py:11-123) + Estimator, 12 cases as one batch."""
    names = [str(c) for c in GS["cases"]]
    assert len(names) == 12
    mpc = _controller(False)
    np.testing.assert_allclose(mpc.get_steady_state_controller_gain(), GS["K"], rtol=1e-12)
    U = _stack(GS, names, "U").transpose(0, 1, 3, 2)
    out = _native.mc_replay(mpc._handle, U, _stack(GS, names, "theta"), _stack(GS, names, "gamma"), _stack(GS, names, "wv"), smart=True)
    assert np.array_equal(out["s"], _stack(GS, names, "s"))
    assert np.array_equal(out["Theta"], _stack(GS, names, "Theta").astype(np.int32))
    assert np.array_equal(out["q"], _stack(GS, names, "q"))
    X = _stack(GS, names, "x")
    scale = np.maximum(np.abs(X).max(axis=(1, 2), keepdims=True), 1.0)
    _check("x", out["x"], X, scale)
    _check("x_hat", out["x_hat"], _stack(GS, names, "xhat"), scale)
    _check("pkt_x", out["x_nom"], _stack(GS, names, "pkt_x"), scale)   # the smart actuator's packet carries the measured state
    _check("u", out["u"], _stack(GS, names, "u"), np.maximum(np.abs(_stack(GS, names, "u")).max(axis=(1, 2), keepdims=True), 1.0))


def test_replay_rejects_missing_arguments():
    mpc = _controller(True)
    h = _native.lib()
    tf = np.zeros((1, 1, 13)); ti = np.zeros((1, 1, 3), np.int32)
    U = np.zeros((1, 1, 11, 1)); fl = np.ones((1, 1), np.uint8); w = np.zeros((1, 1, 4))
    rc = h.tmpc_mc_replay(mpc._handle.ptr, 1, 1, 1, U.ctypes.data, None, fl.ctypes.data, fl.ctypes.data, w.ctypes.data, None,
                          tf.ctypes.data, ti.ctypes.data)
    assert rc != 0 and "NULL" in mpc._handle.error()
