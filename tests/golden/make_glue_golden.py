"""Generates tests/golden/glue_golden.npz by RUNNING THE REFERENCE's numpy-only classes
(`Estimator`, `RobustEstimator`, `ConsistentActuator`, `SmartActuator` from /root/reference/src, importable in the
build container; they need neither cvxpy nor polytope) on scripted packet-loss patterns with
synthetic controller packets.  Only inputs and the recorded outputs are stored; no reference code
travels.  The batched state machines in LinearMPCOverNetworks/{Estimator,SmartActuator}.py must
reproduce these trajectories exactly (tests/test_glue_golden.py).

    python tests/golden/make_glue_golden.py        (needs /root/reference)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src/LinearMPCOverNetworks")
sys.path.insert(0, os.path.join(HERE, ".."))
import Estimator as RefEst            # noqa: E402  (reference module, numpy only)
import SmartActuator as RefAct        # noqa: E402
import common                         # noqa: E402
from LinearMPCOverNetworks.control_lite import dlqr   # noqa: E402

w = common.workload("cartpole")
A, B = w["A"], w["B"]
K, _, _ = dlqr(A, B, w["Q"], w["R"])
Kp = 0.8 * K                      # a different ancillary gain exercises K vs K_plant
N, T, nx, nu = 10, 60, 4, 1


def run(extended, p_loss, seed):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((nx, 1))
    est = (RefEst.RobustEstimator(A, B, K, Kp, x0.copy(), N) if extended else RefEst.Estimator(A, B, K, x0.copy(), N))
    act = RefAct.ConsistentActuator(A, B, K, Kp, x0.copy(), is_extended_MPC_used=extended)
    x = x0.copy()
    rec = {k: [] for k in ("U", "xn0", "theta", "gamma", "wv", "q", "u", "x", "xhat", "xnom", "s", "Theta")}
    for t in range(T):
        theta = 1 if t == 0 else int(rng.uniform() >= p_loss)
        gamma = 1 if t == 0 else int(rng.uniform() >= p_loss)
        wv = rng.uniform(-1, 1, (nx, 1)) * w["w_bound"].reshape(nx, 1)
        q = est.get_qt()
        U = rng.normal(size=(nu, N + 1))
        xn0 = (x + 0.01 * rng.normal(size=(nx, 1))).copy()
        pkt = {"U_t": U.copy(), "q_t": q}
        est.store_sent_control_sequence(U.copy())
        if extended:
            pkt["x_nom_0"] = xn0.copy()
            est.store_current_optimal_inital_nominal_plant_states(xn0.copy())
        u, ppkt = act.process_packet(pkt, x.copy(), theta)
        rec["xnom"].append(np.array(ppkt["x_nom_t"] if extended else ppkt["x_t"]).reshape(nx).copy())
        x = A @ x + B @ u + wv
        est.update_estimate({k: (np.array(v).copy() if isinstance(v, np.ndarray) else v) for k, v in ppkt.items()}, gamma)
        for k, v in (("U", U), ("xn0", xn0.reshape(nx)), ("theta", theta), ("gamma", gamma), ("wv", wv.reshape(nx)), ("q", q),
                     ("u", np.array(u).reshape(nu)), ("x", x.reshape(nx)), ("xhat", np.array(est.get_estimate()).reshape(nx)),
                     ("s", act.get_s_t()), ("Theta", act.get_Theta_t())):
            rec[k].append(np.array(v).copy())
    return {k: np.array(v) for k, v in rec.items()}


def run_smart(p_loss, seed):
    """The comparator's loop (results_linear_system.py:198-205, :262-287): the reference's plain `SmartActuator`
    (SmartActuator.py:11-123) with its `Estimator`; the MPC's packets are synthetic, everything else is the reference's code."""
    rng = np.random.default_rng(1000 + seed)
    x0 = np.zeros((nx, 1))
    est = RefEst.Estimator(A, B, K, x0.copy(), N)
    act = RefAct.SmartActuator(K)
    x = x0.copy()
    rec = {k: [] for k in ("U", "theta", "gamma", "wv", "q", "u", "x", "xhat", "s", "Theta", "pkt_x")}
    for t in range(T):
        theta = 1 if t == 0 else int(rng.uniform() >= p_loss)
        gamma = 1 if t == 0 else int(rng.uniform() >= p_loss)
        wv = rng.uniform(-1, 1, (nx, 1)) * w["w_bound"].reshape(nx, 1)
        q = est.get_qt()
        U = 0.3 * rng.normal(size=(nu, N + 1))
        est.store_sent_control_sequence(U.copy())
        u, ppkt = act.process_packet({"U_t": U.copy(), "q_t": q}, x.copy(), theta)
        rec["pkt_x"].append(np.array(ppkt["x_t"]).reshape(nx).copy())
        x = A @ x + B @ u + wv
        est.update_estimate({k: (np.array(v).copy() if isinstance(v, np.ndarray) else v) for k, v in ppkt.items()}, gamma)
        for k, v in (("U", U), ("theta", theta), ("gamma", gamma), ("wv", wv.reshape(nx)), ("q", q), ("u", np.array(u).reshape(nu)),
                     ("x", x.reshape(nx)), ("xhat", np.array(est.get_estimate()).reshape(nx)), ("s", act.get_s_t()),
                     ("Theta", act.get_Theta_t())):
            rec[k].append(np.array(v).copy())
    return {k: np.array(v) for k, v in rec.items()}


if __name__ == "__main__":
    out = {"A": A, "B": B, "K": K, "Kp": Kp, "N": N}
    cases = []
    for extended in (0, 1):
        for p in (0.0, 0.3, 0.7, 0.9):
            for seed in (1, 2, 3):
                name = f"e{extended}_p{int(p * 10)}_s{seed}"
                cases.append(name)
                for k, v in run(bool(extended), p, seed).items():
                    out[f"{name}/{k}"] = v
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "glue_golden.npz"), **out)
    print(len(cases), "cases written")
    # the plain smart actuator of the non-robust comparator, its own file (glue_golden.npz stays as recorded in round 1)
    smart = {"A": A, "B": B, "K": K, "N": N, "w_bound": w["w_bound"]}
    scases = []
    for p in (0.0, 0.3, 0.6, 0.9):
        for seed in (1, 2, 3):
            name = f"sa_p{int(p * 10)}_s{seed}"
            scases.append(name)
            for k, v in run_smart(p, seed).items():
                smart[f"{name}/{k}"] = v
    smart["cases"] = np.array(scases)
    np.savez_compressed(os.path.join(HERE, "glue_smart_golden.npz"), **smart)
    print(len(scases), "smart-actuator cases written")
