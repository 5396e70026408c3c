"""Parent of every MPC class of the package: model, horizon, cost matrices, constraint sets and the solver name.

Drop-in for the reference module `LinearMPCOverNetworks.RegulatorMPC` as far as the tube-tracking path uses it
(reference RegulatorMPC.py:11-43: constructor and constraint setters; :93-94: `set_solver`), which the reference's own
modules import by this path (`TubeRegulatorMPC.py:12`, `TrackingMPC.py:16`).  The regulator's own QP
(RegulatorMPC.py:45-91, "bring the state to the origin") is a different controller and outside the accelerated path (SURVEY.md 2:
out of scope); the tracking controllers derived from this class bring their own `generate_optimization_problem` /
`solve_optimization_problem`.
"""
from __future__ import annotations

import numpy as np

from .polytope_lite import as_polytope


class RegulatorMPC:
    """State container + constraint setters (reference RegulatorMPC.py:11-43, :93)."""

    def __init__(self, A, B, Q, R, N: int) -> None:
        self._A = np.array(A, dtype=np.float64)
        self._B = np.array(B, dtype=np.float64)
        self._N = int(N)
        self._nx = self._A.shape[1]
        self._nu = self._B.shape[1]
        self._Q = np.array(Q, dtype=np.float64)
        self._R = np.atleast_2d(np.array(R, dtype=np.float64))
        self._X = None
        self._U = None
        # the reference stores cp.CLARABEL here (RegulatorMPC.py:31); this build has
        # exactly one back-end, the HIP library
        self._solver = "hip"

    def set_state_constraints(self, X) -> None:
        self._X = as_polytope(X)

    def set_input_constraints(self, U) -> None:
        self._U = as_polytope(U)

    def set_solver(self, solver) -> None:
        """Reference RegulatorMPC.py:93-94.  Only the HIP back-end exists here."""
        if str(solver).lower() not in ("hip", "clarabel"):
            raise ValueError("this build solves on the MI355X only (solver='hip')")
        self._solver = "hip"
