"""The batched estimator / actuator state machines against golden trajectories recorded from
the REFERENCE's own numpy classes (tests/golden/make_glue_golden.py).  CPU only.  This pins
SURVEY rows a10/a11 (the callers on either side of the QP solve)."""
import os

import numpy as np
import pytest

import common
from LinearMPCOverNetworks.Estimator import BatchedEstimator, Estimator, RobustEstimator
from LinearMPCOverNetworks.SmartActuator import BatchedConsistentActuator, ConsistentActuator

G = np.load(os.path.join(common.GOLDEN, "glue_golden.npz"))
CASES = [str(c) for c in G["cases"]]
TOL = dict(rtol=1e-11, atol=1e-11)    # same sums in a different order ((B,n)@A.T vs A@(n,1)), fed back through gains of ~1e2 over 60 steps


def _replay_single(name):
    ext = name.startswith("e1")
    A, B, K, Kp, N = G["A"], G["B"], G["K"], G["Kp"], int(G["N"])
    g = {k: G[f"{name}/{k}"] for k in ("U", "xn0", "theta", "gamma", "wv", "q", "u", "x", "xhat", "xnom", "s", "Theta")}
    x0 = np.zeros((4, 1))
    est = RobustEstimator(A, B, K, Kp, x0, N) if ext else Estimator(A, B, K, x0, N)
    act = ConsistentActuator(A, B, K, Kp, x0, is_extended_MPC_used=ext)
    x = x0.copy()
    for t in range(len(g["theta"])):
        q = est.get_qt()
        assert q == g["q"][t]
        pkt = {"U_t": g["U"][t], "q_t": q}
        est.store_sent_control_sequence(g["U"][t])
        if ext:
            pkt["x_nom_0"] = g["xn0"][t].reshape(4, 1)
            est.store_current_optimal_inital_nominal_plant_states(g["xn0"][t].reshape(4, 1))
        u, ppkt = act.process_packet(pkt, x, int(g["theta"][t]))
        assert act.get_s_t() == g["s"][t] and act.get_Theta_t() == g["Theta"][t]
        np.testing.assert_allclose(u.reshape(-1), g["u"][t], **TOL)
        x = g["x"][t].reshape(4, 1).copy()          # follow the recorded plant so that errors cannot accumulate
        est.update_estimate(ppkt, int(g["gamma"][t]))
        np.testing.assert_allclose(est.get_estimate().reshape(-1), g["xhat"][t], **TOL)


@pytest.mark.parametrize("name", CASES)
def test_single_trajectory_views_reproduce_reference(name):
    _replay_single(name)


@pytest.mark.parametrize("ext", [0, 1])
def test_batched_state_machines_reproduce_reference(ext):
    """All twelve cases of one kind advance as ONE batch."""
    names = [c for c in CASES if c.startswith(f"e{ext}")]
    A, B, K, Kp, N = G["A"], G["B"], G["K"], G["Kp"], int(G["N"])
    g = {k: np.stack([G[f"{n}/{k}"] for n in names], axis=1) for k in
         ("U", "xn0", "theta", "gamma", "wv", "q", "u", "x", "xhat", "xnom", "s", "Theta")}     # (T, B, ...)
    nb = len(names)
    est = BatchedEstimator(A, B, K, np.zeros((nb, 4)), N, Kp, robust=bool(ext))
    act = BatchedConsistentActuator(A, B, K, Kp, np.zeros((nb, 4)), bool(ext))
    x = np.zeros((nb, 4))
    for t in range(g["theta"].shape[0]):
        q = est.get_qt()
        assert np.array_equal(q, g["q"][t])
        est.store(g["U"][t])
        if ext:
            est.store_x_nom_0(g["xn0"][t])
        u, pk = act.process(g["U"][t], q, x, g["theta"][t], g["xn0"][t] if ext else None)
        assert np.array_equal(act.s, g["s"][t]) and np.array_equal(act.Theta, g["Theta"][t])
        np.testing.assert_allclose(u, g["u"][t], **TOL)
        np.testing.assert_allclose(pk["x_nom_t"] if ext else pk["x_t"], g["xnom"][t], **TOL)
        x = g["x"][t].copy()
        est.update(pk, g["gamma"][t])
        np.testing.assert_allclose(est.get_estimate(), g["xhat"][t], **TOL)


# ---- the plain SmartActuator of the non-robust comparator (reference SmartActuator.py:11-123 with Estimator.py:9-93,
# driven as results_linear_system.py:198-205, :262-287 drives them), recorded by tests/golden/make_glue_golden.py
GS = np.load(os.path.join(common.GOLDEN, "glue_smart_golden.npz"))
SCASES = [str(c) for c in GS["cases"]]
SKEYS = ("U", "theta", "gamma", "wv", "q", "u", "x", "xhat", "s", "Theta", "pkt_x")


@pytest.mark.parametrize("name", SCASES)
def test_plain_smart_actuator_view_reproduces_reference(name):
    from LinearMPCOverNetworks.SmartActuator import SmartActuator
    A, B, K, N = GS["A"], GS["B"], GS["K"], int(GS["N"])
    g = {k: GS[f"{name}/{k}"] for k in SKEYS}
    x0 = np.zeros((4, 1))
    est = Estimator(A, B, K, x0, N)
    act = SmartActuator(K)
    x = x0.copy()
    for t in range(len(g["theta"])):
        q = est.get_qt()
        assert q == g["q"][t]
        est.store_sent_control_sequence(g["U"][t])
        u, ppkt = act.process_packet({"U_t": g["U"][t], "q_t": q}, x, int(g["theta"][t]))
        assert act.get_s_t() == g["s"][t] and act.get_Theta_t() == g["Theta"][t]
        np.testing.assert_allclose(u.reshape(-1), g["u"][t], **TOL)
        np.testing.assert_allclose(np.asarray(ppkt["x_t"]).reshape(-1), g["pkt_x"][t], **TOL)
        x = g["x"][t].reshape(4, 1).copy()
        est.update_estimate(ppkt, int(g["gamma"][t]))
        np.testing.assert_allclose(est.get_estimate().reshape(-1), g["xhat"][t], **TOL)


def test_remote_tracking_loop_reproduces_reference_smart_actuator():
    """montecarlo.run_remote_tracking_mpc -- the host loop the device's TMPC_ACTUATOR_SMART loop is tested against
    (tests/test_tracking_mpc.py) -- builds the plain smart actuator out of BatchedConsistentActuator with x_nom := x and
    a zero ancillary gain.  Fed with the recorded packets, loss patterns and disturbances of all twelve reference runs as
    one batch it must land on the reference's final states and tracking errors."""
    from LinearMPCOverNetworks import montecarlo
    A, B, K, N = GS["A"], GS["B"], GS["K"], int(GS["N"])
    g = {k: np.stack([GS[f"{n}/{k}"] for n in SCASES], axis=0) for k in SKEYS}      # (B, T, ...)
    nb, T = g["theta"].shape
    # loss draws that reproduce the recorded arrivals under `u < p_loss` (results_linear_system.py:218-226)
    p_loss = np.full(nb, 0.5)
    th_u = np.where(g["theta"] == 1, 0.75, 0.25)
    ga_u = np.where(g["gamma"] == 1, 0.75, 0.25)
    step = [0]

    def packets(x_hat, r_t):
        t = step[0]
        np.testing.assert_allclose(x_hat, g["xhat"][:, t - 1] if t else np.zeros((nb, 4)), **TOL)   # the estimate handed to the MPC
        step[0] += 1
        return g["U"][:, t], None, np.zeros(nb, dtype=np.int32)

    ref = np.zeros(T)
    out = montecarlo.run_remote_tracking_mpc(packets, A, B, K, N, p_loss, ref, th_u, ga_u, g["wv"])
    assert step[0] == T
    np.testing.assert_allclose(out["x_final"], g["x"][:, -1], rtol=1e-9, atol=1e-9)
    # results_linear_system.py:297 on the recorded trajectories: x_0 = 0, then the states before each update
    xs = np.concatenate([np.zeros((nb, 1, 4)), g["x"][:, :-1]], axis=1)
    want = np.sqrt(np.sum(xs ** 2, axis=(1, 2))) / T
    np.testing.assert_allclose(out["tracking_error"], want, rtol=1e-9)
