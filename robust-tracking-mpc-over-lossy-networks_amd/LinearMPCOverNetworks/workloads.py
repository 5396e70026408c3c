"""Model definitions of the workloads named in BASELINE.json.

The numbers are the reference's experiment constants: the linearised cartpole
of `Results/results_linear_system.py:26-110` (masses, ZOH at 20 ms, Q/R, the
disturbance box estimated by `estimate_W_for_Cartpole.py`, state/input boxes)
and the double integrator of
`Examples of Model Predictive Controllers/Example_of_Tube_Tracking_MPC.py:19-44`.
"""
from __future__ import annotations

import numpy as np

from .control_lite import c2d
from .polytope_lite import Polytope, box2poly


def cartpole(Th: float = 0.02):
    """Returns dict(A,B,Q,R,X,U,W) for the linearised cartpole (n=4, m=1)."""
    M, m, b, I, g, l = 1.0, 0.1, 0.0, 0.001, 9.8, 0.5
    p = I * (M + m) + M * m * l ** 2
    Ac = np.array([[0, 1, 0, 0],
                   [0, -(I + m * l ** 2) * b / p, -(m ** 2 * g * l ** 2) / p, 0],
                   [0, 0, 0, 1],
                   [0, -(m * l * b) / p, m * g * l * (M + m) / p, 0]], dtype=np.float64)
    Bc = np.array([[0], [(I + m * l ** 2) / p], [0], [-m * l / p]], dtype=np.float64)
    A, B = c2d(Ac, Bc, Th)
    Q = np.diag([100.0, 10.0, 100.0, 10.0])
    R = 0.1 * np.eye(1)
    w = np.array([0.0001, 0.0027, 0.0003, 0.043])
    W = box2poly(np.c_[-w, w])
    xb = np.array([5.0, 5.0, 0.3, 2.0])
    X = box2poly(np.c_[-xb, xb])
    U = box2poly([[-10.0, 10.0]])
    return dict(A=A, B=B, Q=Q, R=R, X=X, U=U, W=W, w_bound=w, name="cartpole")


def double_integrator():
    """Returns dict(A,B,Q,R,X,U,W) for the 2-state double integrator (n=2, m=1)."""
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([[0.0], [1.0]])
    X = box2poly([[-8.0, 8.0]] * 2)
    U = box2poly([[-1.0, 1.0]])
    W = box2poly([[-0.1, 0.1]] * 2)
    return dict(A=A, B=B, Q=np.eye(2), R=np.eye(1), X=X, U=U, W=W,
                w_bound=np.array([0.1, 0.1]), name="double_integrator")


def synthetic(n: int = 12, m: int = 4, seed: int = 5):
    """Random stable model of BASELINE.json config 5 (not in the reference):
    A = 0.95 A0 / rho(A0), A0 ~ N(0,1)^(n x n), B ~ N(0,1)^(n x m); Q = I, R = 0.1 I;
    X = +-10, U = +-1, W = +-0.01 boxes."""
    rng = np.random.default_rng(seed)
    A0 = rng.standard_normal((n, n))
    A = 0.95 * A0 / np.max(np.abs(np.linalg.eigvals(A0)))
    B = rng.standard_normal((n, m))
    w = 0.01 * np.ones(n)
    return dict(A=A, B=B, Q=np.eye(n), R=0.1 * np.eye(m), X=box2poly([[-10.0, 10.0]] * n), U=box2poly([[-1.0, 1.0]] * m),
                W=box2poly(np.c_[-w, w]), w_bound=w, name="synthetic")


CARTPOLE_PARAMS = dict(M=1.0, m=0.1, b=0.0, I=0.001, g=9.8, l=0.5)       # results_linear_system.py:26-33


def cartpole_rhs(x, F, par=CARTPOLE_PARAMS):
    """Continuous-time cart-pole dynamics about the upright position, x = [pos, vel, angle, angular velocity]
    (batched over the leading axis), whose linearisation at the origin is the (Ac, Bc) of
    results_linear_system.py:35-47:
        (M + m) p'' + b p' + m l th'' cos(th) - m l th'^2 sin(th) = F
        (I + m l^2) th'' + m l p'' cos(th) - m g l sin(th) = 0
    The reference integrates the same mechanism with PyBullet at 500 Hz (Results/Cartpole/cartpole.py); this is the
    closed-form counterpart used by the device-resident closed loop (plant = 'cartpole')."""
    x = np.asarray(x, dtype=np.float64)
    F = np.asarray(F, dtype=np.float64).reshape(x.shape[:-1])
    M, m, b, I, g, l = (par[k] for k in ("M", "m", "b", "I", "g", "l"))
    vel, th, om = x[..., 1], x[..., 2], x[..., 3]
    s, c = np.sin(th), np.cos(th)
    a11, a12, a22 = M + m, m * l * c, I + m * l * l
    r1 = F - b * vel + m * l * om * om * s
    r2 = m * g * l * s
    det = a11 * a22 - a12 * a12
    acc = (r1 * a22 - a12 * r2) / det
    alp = (a11 * r2 - a12 * r1) / det
    return np.stack([vel, acc, om, alp], axis=-1)


def cartpole_trace(x, F, Th: float = 0.02, substeps: int = 10, par=CARTPOLE_PARAMS):
    """Zero-order hold of the force over one sampling period Th, classical RK4 with `substeps` steps (500 Hz for
    Th = 20 ms, the reference's physics rate, results_nonlinear_system.py:30-37).  Returns the states at the physics
    steps, (substeps + 1, ...): [0] = x, [-1] = the state one sampling period later."""
    x = np.array(x, dtype=np.float64)
    dt = Th / substeps
    xs = [x]
    for _ in range(substeps):
        k1 = cartpole_rhs(x, F, par)
        k2 = cartpole_rhs(x + 0.5 * dt * k1, F, par)
        k3 = cartpole_rhs(x + 0.5 * dt * k2, F, par)
        k4 = cartpole_rhs(x + dt * k3, F, par)
        x = x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        xs.append(x)
    return np.stack(xs)


def cartpole_step(x, F, Th: float = 0.02, substeps: int = 10, par=CARTPOLE_PARAMS):
    """The state one sampling period later (cartpole_trace)."""
    return cartpole_trace(x, F, Th, substeps, par)[-1]


# --------------------------------------------------------------------------- controllers for the named workloads
def make_controller(name: str = "cartpole", N: int = 10, fixed_initial_state: bool = True, extended: bool = False,
                    device: int = 0, tracking: bool = False, verbose: bool = False):
    """Controller of a BASELINE workload, set up the way the reference's scripts do it
    (results_linear_system.py:113-128, Example_of_Tube_Tracking_MPC.py:46-53): constraints, then
    `setup_optimization(W, fixed_initial_state, rpi_method)` -- mRPI, tightening, terminal set (and Z (-) W for the
    extended controller) through the batched LP kernel on `device` (0.2 s for the cartpole), then the device-resident QP.
    Returns (mpc, model dict).  tracking=True: the non-robust comparator TrackingMPC (results_linear_system.py:132-140)."""
    import contextlib
    import io
    from . import polytope_lite
    from .TrackingMPC import TrackingMPC
    from .TubeTrackingMPC import ExtendedTubeTrackingMPC, TubeTrackingMPC
    model = {"cartpole": cartpole, "double_integrator": double_integrator, "synthetic": synthetic}[name]()
    cls = TrackingMPC if tracking else (ExtendedTubeTrackingMPC if extended else TubeTrackingMPC)
    mpc = cls(model["A"], model["B"], model["Q"], model["R"], N)
    mpc.set_input_constraints(model["U"])
    mpc.set_state_constraints(model["X"])
    mpc.set_device(device)
    old = polytope_lite.set_lp_backend("hip", device)
    try:
        with (contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO())):     # the reference's progress prints
            if tracking:
                mpc.setup_optimization()
            else:
                # the double integrator example uses the default (Rakovic) mRPI, the result scripts pass rpi_method = 1 (Darup)
                mpc.setup_optimization(model["W"], fixed_initial_state=fixed_initial_state,
                                       rpi_method=0 if name == "double_integrator" else 1)
    finally:
        polytope_lite.set_lp_backend(old)
    return mpc, model


def harvest_closed_loop_states(mpc, model: dict, n_traj: int, T: int, seed: int = 0, extended: bool = False,
                               p_loss: float = 0.3, ref_hold: int = 25, ref_range: float = 2.5):
    """(x_hat_k, ref_k[, gamma_k]) triples the remote controller is actually asked to solve: `n_traj` closed loops over
    the lossy network (montecarlo.run_remote_tube_mpc around this controller's own device solver), every trajectory with its
    own piecewise-constant position reference (a new level every `ref_hold` steps, uniform in +-ref_range) and its own loss /
    disturbance realisation.  Returns X (n_traj*T, nx), R (n_traj*T, nx), gamma (n_traj*T,) uint8 -- all distinct states,
    transients and steady phases alike.  Synthetic input generator of bench.py and the scripts."""
    from . import montecarlo
    rng = np.random.default_rng(seed)
    nx = model["A"].shape[0]
    levels = rng.uniform(-ref_range, ref_range, (n_traj, (T + ref_hold - 1) // ref_hold))
    refs = np.repeat(levels, ref_hold, axis=1)[:, :T]                       # (n_traj, T)
    th, ga, w = montecarlo.draw_realisations(n_traj, T, model["w_bound"], seed=seed + 17)
    Xs, Rs, Gs = [], [], []

    def packets(x_hat, r_t, gamma=None):
        Xs.append(np.array(x_hat, dtype=np.float64))
        Rs.append(np.array(r_t, dtype=np.float64))
        Gs.append(np.ones(len(x_hat), np.uint8) if gamma is None else np.array(gamma, dtype=np.uint8))
        return mpc.determine_packets(x_hat, r_t, gamma) if gamma is not None else mpc.determine_packets(x_hat, r_t)

    montecarlo.run_remote_tube_mpc(packets, model["A"], model["B"], mpc.get_steady_state_controller_gain(),
                                   mpc.get_ancillary_controller_gain(), mpc._N, mpc._Z, np.full(n_traj, p_loss), refs, th, ga, w,
                                   extended=extended)
    return np.concatenate(Xs), np.concatenate(Rs), np.concatenate(Gs)
