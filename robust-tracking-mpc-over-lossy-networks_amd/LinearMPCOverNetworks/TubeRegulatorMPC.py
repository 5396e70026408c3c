"""Parent pieces of the tube-tracking controller that the hot path consumes.

Only what `TubeTrackingMPC` inherits is provided: the constructor chain
(reference `RegulatorMPC.py:11-43`, `TubeRegulatorMPC.py:16-24`) that produces
`K`, `P`, `Acl`, and the mRPI driver with its s_max x10 retry
(`TubeRegulatorMPC.py:26-78`).  The regulator QPs of those two reference classes
are different controllers and are not part of the accelerated path.
"""
from __future__ import annotations

import numpy as np

from . import utils_polytope as up
from .RegulatorMPC import RegulatorMPC
from .control_lite import dlqr, dlyap
from .polytope_lite import reduce


class TubeRegulatorMPC(RegulatorMPC):
    def __init__(self, A, B, Q, R, N: int) -> None:
        super().__init__(A, B, Q, R, N)
        K, _, _ = dlqr(self._A, self._B, self._Q, self._R)      # TubeRegulatorMPC.py:19
        self._K = K
        Q_lyap = self._Q + K.T @ self._R @ K
        Q_lyap = (Q_lyap + Q_lyap.T) / 2
        self._Acl = self._A - self._B @ K
        # python-control convention  Acl P Acl^T - P + Q_lyap = 0  (TubeRegulatorMPC.py:23)
        self._P = dlyap(self._Acl, Q_lyap)
        self._Z = None

    def determine_mRPI(self, W, eps_var: float = 1.9e-5, Acl=None, rpi_method: int = 0, K=None):
        """Reference TubeRegulatorMPC.py:26-78."""
        if K is None:
            K = self._K
        if Acl is None:
            Acl = self._Acl
        if np.max(np.abs(np.linalg.eigvals(Acl))) >= 1:
            print("The matrix Acl is not stable, such that the algorithm will never converge. \n"
                  " Therefore, None is returned")
            return None
        s_max = 200
        while True:
            if rpi_method == 1:
                Fs_temp, status = up.calculate_RPI(Acl, W, self._X, self._U, K, eps_var=eps_var, s_max=s_max)
            else:
                if rpi_method != 0:
                    print("The method chosen to determine the RPI does not exists, so we use the default method 0")
                Fs_temp, status = up.calculate_minimal_robust_positively_invariant_set(
                    Acl, W=W, eps_var=eps_var, s_max=s_max)
            if status == 0:
                break
            if status == -2:
                raise ValueError("determine_mRPI: the disturbance set is too large for the state/input constraints "
                                 "(no RPI set fits inside them); the reference would retry with a larger s_max forever")
            if status == -3:
                raise ValueError("determine_mRPI: the container set of the Darup-Teichrib construction fails its contraction "
                                 "test for this model (independent of s_max); try rpi_method=0")
            print(f"RPI not determined in {s_max} steps. Increasing s_max to 10*s_max = {10 * s_max}")
            s_max *= 10
        self._Z = reduce(Fs_temp)
        return self._Z
