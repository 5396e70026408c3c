"""Discrete LQR / Lyapunov / ZOH helpers (scipy only).

Stand-ins for the three python-control calls the reference makes while building
a model: `ct.dlqr` (TubeRegulatorMPC.py:19), `ct.dlyap` (TubeRegulatorMPC.py:23)
and `ct.c2d` (results_linear_system.py:59-61).  python-control is not part of
this image.  Conventions follow python-control's documentation:

* dlqr(A,B,Q,R) -> (K, S, E) with u = -K x, S the DARE solution;
* dlyap(A,Q) solves  A X A^T - X + Q = 0;
* c2d(..., Ts) is the zero-order-hold discretisation.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm, solve_discrete_are, solve_discrete_lyapunov


def dlqr(A, B, Q, R):
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    Q = np.asarray(Q, dtype=np.float64)
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))
    S = solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(R + B.T @ S @ B, B.T @ S @ A)
    E = np.linalg.eigvals(A - B @ K)
    return K, S, E


def dlyap(A, Q):
    return solve_discrete_lyapunov(np.asarray(A, dtype=np.float64), np.asarray(Q, dtype=np.float64))


def c2d(Ac, Bc, Ts: float):
    """Zero-order hold: [[Ad, Bd], [0, I]] = expm([[Ac, Bc], [0, 0]] Ts)."""
    Ac = np.asarray(Ac, dtype=np.float64)
    Bc = np.asarray(Bc, dtype=np.float64)
    n, m = Ac.shape[0], Bc.shape[1]
    Mx = np.zeros((n + m, n + m))
    Mx[:n, :n] = Ac
    Mx[:n, n:] = Bc
    E = expm(Mx * Ts)
    return E[:n, :n], E[:n, n:]
