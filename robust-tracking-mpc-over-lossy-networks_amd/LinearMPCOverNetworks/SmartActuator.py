"""Plant-side packet logic, batched over B independent trajectories.

Counterpart of the reference's `SmartActuator` / `ConsistentActuator`
(reference SmartActuator.py:11-231), written as an O(1)-state machine per trajectory so
that a whole Monte-Carlo batch advances with a few vector operations per time step:

* the reference appends every theta_t to a vector and evaluates
  Theta_t = prod(theta[q_t+1:]) (SmartActuator.py:62-67); only "the last step whose packet
  was lost" matters for that product, so it is kept as one integer per trajectory;
* s_t = Theta_t t + (1 - Theta_t) s_t (SmartActuator.py:77), the buffered sequence is replaced
  iff Theta_t = 1 (SmartActuator.py:219-222), the applied nominal input is U[:, t - s_t] inside
  the horizon and U[:, -1] - K x_nom beyond it (SmartActuator.py:100-103), the ancillary law is
  u = u_nom - K_plant (x - x_nom) (SmartActuator.py:171) and the nominal model advances with
  u_nom (SmartActuator.py:152).

`ConsistentActuator` / `SmartActuator` below are single-trajectory views with the reference's
method names and packet dictionaries.
"""
from __future__ import annotations

import numpy as np


class BatchedConsistentActuator:
    def __init__(self, A, B, K, K_plant, x0, is_extended_MPC_used: bool = False):
        self.A = np.asarray(A, dtype=np.float64)
        self.B = np.asarray(B, dtype=np.float64)
        self.K = np.atleast_2d(np.asarray(K, dtype=np.float64))
        self.K_plant = np.atleast_2d(np.asarray(K_plant, dtype=np.float64))
        self.x_nom = np.array(x0, dtype=np.float64).reshape(-1, self.A.shape[0]).copy()
        nb = self.x_nom.shape[0]
        self.extended = bool(is_extended_MPC_used)
        self.t = 0
        self.q = np.zeros(nb, dtype=np.int64)
        self.s = np.zeros(nb, dtype=np.int64)
        self.Theta = np.zeros(nb, dtype=np.int64)
        self.last_lost = np.full(nb, -1, dtype=np.int64)      # last step with theta == 0
        self.U = None                                          # (B, nu, N+1) buffered sequences

    def process(self, U_t, q_t, x, theta, x_nom_0=None):
        """One step for the whole batch.
        U_t (B, nu, N+1), q_t (B,), x (B, nx) plant states, theta (B,) in {0,1},
        x_nom_0 (B, nx) or None.  Returns u (B, nu) and the plant packet
        {'x_t', 's_t'[, 'x_nom_t']} (arrays over the batch)."""
        theta = np.asarray(theta).astype(np.int64).reshape(-1)
        x = np.asarray(x, dtype=np.float64).reshape(self.x_nom.shape)
        t = self.t
        self.last_lost = np.where(theta == 0, t, self.last_lost)
        recv = theta == 1
        self.q = np.where(recv, np.asarray(q_t, dtype=np.int64), self.q)
        self.Theta = (recv & (self.last_lost <= self.q)).astype(np.int64)
        self.s = np.where(self.Theta == 1, t, self.s)
        acc = self.Theta == 1
        U_t = np.asarray(U_t, dtype=np.float64)
        if self.U is None:
            self.U = np.zeros_like(U_t)
        self.U[acc] = U_t[acc]
        if x_nom_0 is not None:
            self.x_nom[acc] = np.asarray(x_nom_0, dtype=np.float64).reshape(self.x_nom.shape)[acc]
        x_nom_t = self.x_nom.copy()
        N = self.U.shape[2] - 1
        d = t - self.s
        inside = d < N
        idx = np.where(inside, d, N)
        u_nom = np.take_along_axis(self.U, idx[:, None, None], axis=2)[:, :, 0]
        u_nom = np.where(inside[:, None], u_nom, u_nom - x_nom_t @ self.K.T)
        u = u_nom - (x - x_nom_t) @ self.K_plant.T
        if self.extended:
            packet = {"x_t": x.copy(), "s_t": self.s.copy(), "x_nom_t": x_nom_t}
        else:
            packet = {"x_t": x_nom_t, "s_t": self.s.copy()}
        self.x_nom = x_nom_t @ self.A.T + u_nom @ self.B.T
        self.t += 1
        return u, packet


class ConsistentActuator:
    """Single-trajectory view with the reference's interface (SmartActuator.py:125-231)."""

    def __init__(self, A, B, K, K_plant, x0, is_extended_MPC_used: bool = False):
        self._nx = np.asarray(A).shape[1]
        self._nu = np.asarray(B).shape[1]
        self._core = BatchedConsistentActuator(A, B, K, K_plant, np.asarray(x0, dtype=np.float64).reshape(1, -1),
                                               is_extended_MPC_used)

    def process_packet(self, packet: dict, x_t, theta_t):
        U = np.asarray(packet["U_t"], dtype=np.float64)[None]
        xn0 = packet.get("x_nom_0")
        u, pk = self._core.process(U, np.array([packet["q_t"]]), np.asarray(x_t, dtype=np.float64).reshape(1, -1),
                                   np.array([theta_t]), None if xn0 is None else np.asarray(xn0).reshape(1, -1))
        out = {"x_t": pk["x_t"][0].reshape(self._nx, 1), "s_t": int(pk["s_t"][0])}
        if "x_nom_t" in pk:
            out["x_nom_t"] = pk["x_nom_t"][0].reshape(self._nx, 1)
        return u[0].reshape(self._nu, 1), out

    def get_x_nom(self):
        return self._core.x_nom[0].reshape(self._nx, 1)

    def get_s_t(self):
        return int(self._core.s[0])

    def get_Theta_t(self):
        return int(self._core.Theta[0])


class SmartActuator(ConsistentActuator):
    """The reference's plain smart actuator (SmartActuator.py:11-123) is the consistent actuator
    without a nominal model: no ancillary feedback and the terminal law acts on the measured
    state.  It is only used by the non-robust comparator (results_linear_system.py:205)."""

    def __init__(self, K):
        K = np.atleast_2d(np.asarray(K, dtype=np.float64))
        nx, nu = K.shape[1], K.shape[0]
        self._nx, self._nu = nx, nu
        self._core = BatchedConsistentActuator(np.zeros((nx, nx)), np.zeros((nx, nu)), K, np.zeros((nu, nx)), np.zeros((1, nx)))

    def process_packet(self, packet: dict, x_t, theta_t):
        # the "nominal state" of this degenerate case is the measured state itself
        self._core.x_nom = np.asarray(x_t, dtype=np.float64).reshape(1, -1).copy()
        u, pk = super().process_packet(packet, x_t, theta_t)
        return u, {"x_t": np.asarray(x_t, dtype=np.float64).reshape(self._nx, 1), "s_t": pk["s_t"]}
