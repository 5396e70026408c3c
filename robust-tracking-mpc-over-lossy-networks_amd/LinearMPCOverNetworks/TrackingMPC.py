"""Non-robust tracking MPC -- the R-MPC comparator of the reference's result scripts -- on the same
device kernels.

Drop-in for the reference class `TrackingMPC` (TrackingMPC.py:20-198; Limon 2008 + the remote
packetisation of Pezzutto 2022): same constructor, `setup_optimization()`, `solve_optimization_problem`,
`determine_packet`, `encapsulate(u_mpc, x_bar, u_bar, q_t)`, gain / timing accessors.  Its QP
(TrackingMPC.py:62-115) is the tube-tracking QP with the initial state fixed to the estimate
(:87), the UN-tightened sets X and U (:94-97) and the terminal set computed from them
(:155-185): the same condensed form, the same kernels, different (H, h).

Without a terminal set (no `setup_optimization()` / `determine_Xf()` yet) the reference constrains x_N == x_bar
(TrackingMPC.py:105-107); so does this class (`terminal_equality` of include/tmpc.h: the nx equalities are eliminated at
set-up like the dynamics).
"""
from __future__ import annotations

import time

import numpy as np

from .TubeTrackingMPC import TubeTrackingMPC


class TrackingMPC(TubeTrackingMPC):

    _smart_actuator = True             # results_linear_system.py:198-205: R-MPC runs with Estimator + SmartActuator

    def setup_optimization(self):
        """TrackingMPC.py:187-192."""
        self._Xc, self._Uc = self._X, self._U            # no tightening: the nominal prediction IS the prediction
        self.determine_Xf()
        self.generate_optimization_problem()

    def setup_from_sets(self, sets: dict, create: bool = True):
        from .polytope_lite import Polytope
        self._Xc, self._Uc = self._X, self._U
        self._Xf = Polytope(sets["Xf_A"], sets["Xf_b"])
        self._fixed_initial_state = True
        if create:
            self.generate_optimization_problem()

    def generate_optimization_problem(self, fixed_initial_state: bool = True):
        """TrackingMPC.py:62-115.  Before `setup_optimization()` / `determine_Xf()` there is no terminal set and the reference
        constrains x_N == x_bar instead (:105-107); the library eliminates those nx equalities at set-up (tmpc.h:
        terminal_equality)."""
        if self._Xc is None:
            self._Xc, self._Uc = self._X, self._U
        super().generate_optimization_problem(True)

    def _problem_dict(self) -> dict:
        if self._Xf is not None:
            return super()._problem_dict()
        K_anc = self._K if self._K_ancillary is None else self._K_ancillary
        return dict(nx=self._nx, nu=self._nu, N=self._N, A=self._A, B=self._B, Q=self._Q, R=self._R, P=self._P, T=self._Tout,
                    K=self._K, K_anc=K_anc, Hx=self._Xc.A, hx=self._Xc.b, Hu=self._Uc.A, hu=self._Uc.b,
                    fixed_x0=1, extended=0, terminal_equality=1, tol=self._tol, max_iter=self._max_iter)

    def solve_optimization_problem(self, x_init, ref, verbose_MPC: bool = False):
        """TrackingMPC.py:117-135 (returns x_mpc, u_mpc, x_bar, u_bar), batched over a leading axis when given one."""
        batched = self._is_batched(x_init)
        return self._unpack(self._solve(x_init, ref), batched, "tracking MPC")

    def determine_packet(self, x_hat, ref, q_t):
        """TrackingMPC.py:47-60."""
        start = time.time()
        _, u_mpc, x_ss, u_ss = self.solve_optimization_problem(np.asarray(x_hat).reshape(-1), ref)
        self._computational_times.append(time.time() - start)
        return self.encapsulate(u_mpc, x_ss, u_ss, q_t)

    def encapsulate(self, u_mpc, x_bar, u_bar, q_t):
        """U_t = [u_mpc | u_bar + K x_bar]  (TrackingMPC.py:141-153; note the argument order)."""
        if x_bar is not None:
            U_t = np.hstack((u_mpc, (u_bar + self._K @ x_bar).reshape(u_mpc.shape[0], 1)))
        else:
            U_t = None
        return {"U_t": U_t, "q_t": q_t}
