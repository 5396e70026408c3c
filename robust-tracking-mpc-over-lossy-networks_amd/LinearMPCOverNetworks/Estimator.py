"""Controller-side state estimator, batched over B independent trajectories.

Counterpart of the reference's `Estimator` / `RobustEstimator` (reference Estimator.py:9-161):
received packet (gamma = 1): x_hat = A x_pkt + B u_hat with u_hat read from the sequence the plant
reports to be using, `U_{s_t}[:, t - s_t]` inside the horizon and `U_{s_t}[:, -1] - K x_pkt` beyond
it (Estimator.py:50-65); lost packet: x_hat = A x_hat + B U_latest[:, 0] (Estimator.py:68-74);
q_t = gamma t + (1 - gamma) q_t (Estimator.py:92).  The robust variant adds the ancillary term
(Estimator.py:131-140) and restarts from the MPC's own x_nom_0 when the packet is lost
(Estimator.py:145-152).

The reference keeps every sequence ever sent in a Python list; here they live in one
(T, B, nu, N+1) array that is indexed by s_t.
"""
from __future__ import annotations

import numpy as np


class BatchedEstimator:
    def __init__(self, A, B, K, x0, N: int, K_plant=None, robust: bool = False):
        self.A = np.asarray(A, dtype=np.float64)
        self.B = np.asarray(B, dtype=np.float64)
        self.K = np.atleast_2d(np.asarray(K, dtype=np.float64))
        self.K_plant = None if K_plant is None else np.atleast_2d(np.asarray(K_plant, dtype=np.float64))
        self.robust = bool(robust)
        self.x_hat = np.array(x0, dtype=np.float64).reshape(-1, self.A.shape[0]).copy()
        nb = self.x_hat.shape[0]
        self.N = int(N)
        self.t = 0
        self.q = np.zeros(nb, dtype=np.int64)
        self.hist = None
        self.n_sent = 0
        self.x_nom_0 = None

    def get_qt(self):
        return self.q.copy()

    def get_estimate(self):
        return self.x_hat.copy()

    def store(self, U_t):
        """Remember the sequence just sent (Estimator.py:34-41)."""
        U_t = np.asarray(U_t, dtype=np.float64)
        if self.hist is None:
            self.hist = np.zeros((64,) + U_t.shape)
        if self.n_sent == self.hist.shape[0]:
            self.hist = np.concatenate([self.hist, np.zeros_like(self.hist)], axis=0)
        self.hist[self.n_sent] = U_t
        self.n_sent += 1

    def store_x_nom_0(self, x_nom_0):
        self.x_nom_0 = np.asarray(x_nom_0, dtype=np.float64).reshape(self.x_hat.shape).copy()

    def update(self, packet: dict, gamma):
        gamma = np.asarray(gamma).astype(np.int64).reshape(-1)
        nb = self.x_hat.shape[0]
        ar = np.arange(nb)
        recv = gamma == 1
        s_t = np.asarray(packet["s_t"], dtype=np.int64).reshape(-1)
        x_pkt = np.asarray(packet["x_t"], dtype=np.float64).reshape(self.x_hat.shape)
        U_s = self.hist[np.clip(s_t, 0, self.n_sent - 1), ar]            # (B, nu, N+1)
        d = self.t - s_t
        inside = d < self.N
        u = np.take_along_axis(U_s, np.where(inside, d, self.N)[:, None, None], axis=2)[:, :, 0]
        if self.robust:
            x_nom = np.asarray(packet["x_nom_t"], dtype=np.float64).reshape(self.x_hat.shape)
            u = np.where(inside[:, None], u, u - x_nom @ self.K.T)
            u = u - (x_pkt - x_nom) @ self.K_plant.T
        else:
            u = np.where(inside[:, None], u, u - x_pkt @ self.K.T)
        x_recv = x_pkt @ self.A.T + u @ self.B.T
        u0 = self.hist[self.n_sent - 1][:, :, 0]
        base = self.x_nom_0 if self.robust else self.x_hat
        x_lost = base @ self.A.T + u0 @ self.B.T
        self.x_hat = np.where(recv[:, None], x_recv, x_lost)
        self.q = np.where(recv, self.t, self.q)
        self.t += 1


class Estimator:
    """Single-trajectory view with the reference's interface (Estimator.py:9-98)."""
    _robust = False

    def __init__(self, A, B, K, x0, N: int, K_plant=None):
        self._nx = np.asarray(A).shape[1]
        self._core = BatchedEstimator(A, B, K, np.asarray(x0, dtype=np.float64).reshape(1, -1), N, K_plant, self._robust)

    def store_sent_control_sequence(self, Ut):
        self._core.store(np.asarray(Ut, dtype=np.float64)[None])

    def update_estimate(self, packet: dict, gamma_t: int):
        pk = {"x_t": np.asarray(packet["x_t"], dtype=np.float64).reshape(1, -1), "s_t": np.array([packet["s_t"]])}
        if "x_nom_t" in packet:
            pk["x_nom_t"] = np.asarray(packet["x_nom_t"], dtype=np.float64).reshape(1, -1)
        self._core.update(pk, np.array([gamma_t]))

    def get_estimate(self):
        return self._core.x_hat[0].reshape(self._nx, 1)

    def get_qt(self):
        return int(self._core.q[0])


class RobustEstimator(Estimator):
    """Estimator.py:101-161."""
    _robust = True

    def __init__(self, A, B, K, K_plant, x0, N: int):
        super().__init__(A, B, K, x0, N, K_plant)

    def store_current_optimal_inital_nominal_plant_states(self, x_nom_0):
        self._core.store_x_nom_0(np.asarray(x_nom_0, dtype=np.float64).reshape(1, -1))
