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


def cartpole_step(x, F, Th: float = 0.02, substeps: int = 10, par=CARTPOLE_PARAMS):
    """Zero-order hold of the force over one sampling period Th, classical RK4 with `substeps` steps (500 Hz for
    Th = 20 ms, the reference's physics rate, results_nonlinear_system.py:30-37)."""
    x = np.array(x, dtype=np.float64)
    dt = Th / substeps
    for _ in range(substeps):
        k1 = cartpole_rhs(x, F, par)
        k2 = cartpole_rhs(x + 0.5 * dt * k1, F, par)
        k3 = cartpole_rhs(x + 0.5 * dt * k2, F, par)
        k4 = cartpole_rhs(x + dt * k3, F, par)
        x = x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x
