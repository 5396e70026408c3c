"""Tube-tracking MPC whose per-timestep QP is solved on the MI355X.

Drop-in for the reference classes `TubeTrackingMPC` (TubeTrackingMPC.py:20-246)
and `ExtendedTubeTrackingMPC` (TubeTrackingMPC.py:249-369): same constructor,
same `set_*_constraints` / `setup_optimization` / `solve_optimization_problem` /
`determine_packet` / `encapsulate` / gain and timing accessors, same private
attributes the reference's scripts read (`_Z`, `_Xf`, `_Xc`, `_Uc`, `_K`, `_P`).

What differs:

* `generate_optimization_problem` does not build a cvxpy problem
  (TubeTrackingMPC.py:104-156).  It hands the model, weights and the four
  polytopes to `tmpc_create` (include/tmpc.h), which condenses the QP once on
  the host and keeps it resident in HBM.
* `solve_optimization_problem(x_init, ref)` calls `tmpc_solve_batch`.  Inputs may
  carry a leading batch axis, `(B, nx)`; every trajectory of a Monte-Carlo sweep
  is then solved by one kernel launch.  1-D inputs return exactly the reference's
  shapes: `x_nom (nx, N+1)`, `u_nom (nu, N)`, `x_ss (nx,)`, `u_ss (nu,)`, or four
  `None` when the instance is infeasible (TubeTrackingMPC.py:189-194).
* The caller's arrays are not reshaped in place (TubeTrackingMPC.py:175-180 do).

There is no CPU fall-back: without the HIP library `setup_optimization` raises.
"""
from __future__ import annotations

import time

import numpy as np

from . import utils_polytope as up
from .TubeRegulatorMPC import TubeRegulatorMPC
from .polytope_lite import Polytope, as_polytope

STATUS_OPTIMAL = 0
STATUS_MAX_ITER = 1
STATUS_INFEASIBLE = 2
STATUS_NUMERICAL = 3
_STATUS_TEXT = {0: "optimal", 1: "optimal_inaccurate", 2: "infeasible", 3: "solver_error"}


class TubeTrackingMPC(TubeRegulatorMPC):

    _VARIANTS = 1
    _smart_actuator = False            # closed loop: ConsistentActuator (tube MPC); TrackingMPC pairs with the plain SmartActuator

    def __init__(self, A, B, Q, R, N: int, lambda_param: float = 0.99999):
        super().__init__(A, B, Q, R, N)
        self._lambda = float(lambda_param)
        self._Tout = 10 * self._P                      # TubeTrackingMPC.py:27
        self._K_ancillary = None
        self._Acl_plant = None
        self._computational_times = []
        self._Xf = self._Xc = self._Uc = None
        self._ZmW = None
        self._XfP = None                               # terminal set of the packet-received problem, auxiliaries eliminated
        self._eliminate_auxiliaries = True
        self._fixed_initial_state = False
        self._handle = None
        self._device = 0
        self._tol = 1e-7
        self._max_iter = 60
        self.last_status = None
        self.last_iters = None

    # ------------------------------------------------------------------ offline
    def determine_Xf(self, verbose: bool = True):
        """Terminal set = maximal output-admissible set of the augmented
        (x, x_bar, u_bar) dynamics (TubeTrackingMPC.py:35-61)."""
        nx, nu, K = self._nx, self._nu, self._K
        Hu, hu = self._Uc.A, self._Uc.b
        Hx, hx = self._Xc.A, self._Xc.b
        Ae = np.block([[self._Acl, self._B @ K, self._B],
                       [np.zeros((nx, nx)), np.eye(nx), np.zeros((nx, nu))],
                       [np.zeros((nu, nx)), np.zeros((nu, nx)), np.eye(nu)]])
        rx, ru = Hx.shape[0], Hu.shape[0]
        Hcl = np.block([[Hx, np.zeros((rx, nx)), np.zeros((rx, nu))],
                        [-Hu @ K, Hu @ K, Hu],
                        [np.zeros((rx, nx)), Hx, np.zeros((rx, nu))],
                        [np.zeros((ru, nx)), np.zeros((ru, nx)), Hu]])
        hcl = np.r_[hx, hu, self._lambda * hx, self._lambda * hu]
        self._Xf = up.calculate_maximum_admissible_output_set(Ae, Polytope(Hcl, hcl), verbose=verbose)
        return self._Xf

    def determine_mRPI(self, W, epsilon: float = 1e-4, Acl=None, rpi_method: int = 0):
        """TubeTrackingMPC.py:63-88: eps = 1e-4 and the ancillary loop if one is set."""
        if Acl is None:
            Acl = self._Acl if self._Acl_plant is None else self._Acl_plant
        K = self._K if self._K_ancillary is None else self._K_ancillary
        return super().determine_mRPI(W, epsilon, Acl=Acl, K=K, rpi_method=rpi_method)

    def tighten_constraints(self):
        """Uc = U (-) (-K_anc Z), Xc = X (-) Z (TubeTrackingMPC.py:90-102)."""
        K = self._K if self._K_ancillary is None else self._K_ancillary
        self._Uc = up.pont_diff(self._U, up.scale(self._Z, -K))
        self._Xc = up.pont_diff(self._X, self._Z)

    def generate_optimization_problem(self, fixed_initial_state: bool = False):
        """Build the device-resident QP (replaces TubeTrackingMPC.py:104-156)."""
        from . import _native
        self._fixed_initial_state = bool(fixed_initial_state)
        self._close()
        self._handle = _native.create(self._problem_dict(), self._device)

    def setup_optimization(self, W, fixed_initial_state: bool = False, rpi_method: int = 0):
        """TubeTrackingMPC.py:158-168."""
        self._W = as_polytope(W)
        self.determine_mRPI(self._W, rpi_method=rpi_method)
        self.tighten_constraints()
        self.determine_Xf()
        self.generate_optimization_problem(fixed_initial_state)

    # -- cached sets: the LP-heavy set-up is minutes for the cartpole; its result
    #    can be saved and re-loaded so tests and benchmarks start from the same sets
    def export_sets(self) -> dict:
        d = {"Z_A": self._Z.A, "Z_b": self._Z.b, "Xc_A": self._Xc.A, "Xc_b": self._Xc.b,
             "Uc_A": self._Uc.A, "Uc_b": self._Uc.b, "Xf_A": self._Xf.A, "Xf_b": self._Xf.b}
        if self._ZmW is not None:
            d["ZmW_A"], d["ZmW_b"] = self._ZmW.A, self._ZmW.b
        if self._XfP is not None:
            d["XfP_A"], d["XfP_b"] = self._XfP.A, self._XfP.b
        return d

    def setup_from_sets(self, sets: dict, fixed_initial_state: bool = False, create: bool = True):
        self._Z = Polytope(sets["Z_A"], sets["Z_b"])
        self._Xc = Polytope(sets["Xc_A"], sets["Xc_b"])
        self._Uc = Polytope(sets["Uc_A"], sets["Uc_b"])
        self._Xf = Polytope(sets["Xf_A"], sets["Xf_b"])
        if "ZmW_A" in sets:
            self._ZmW = Polytope(sets["ZmW_A"], sets["ZmW_b"])
        if "XfP_A" in sets:
            self._XfP = Polytope(sets["XfP_A"], sets["XfP_b"])
        self._fixed_initial_state = bool(fixed_initial_state)
        if create:
            self.generate_optimization_problem(fixed_initial_state)

    def _problem_dict(self) -> dict:
        """Flat description handed across the C ABI (include/tmpc.h: tmpc_problem)."""
        K_anc = self._K if self._K_ancillary is None else self._K_ancillary
        d = dict(nx=self._nx, nu=self._nu, N=self._N,
                 A=self._A, B=self._B, Q=self._Q, R=self._R, P=self._P, T=self._Tout,
                 K=self._K, K_anc=K_anc,
                 Hx=self._Xc.A, hx=self._Xc.b, Hu=self._Uc.A, hu=self._Uc.b,
                 HT=self._Xf.A, hT=self._Xf.b,
                 fixed_x0=int(self._fixed_initial_state), extended=0,
                 tol=self._tol, max_iter=self._max_iter)
        if not self._fixed_initial_state:
            d["HZ"], d["hZ"] = self._Z.A, self._Z.b
        return d

    # ------------------------------------------------------------------ online
    def _solve(self, x_init, ref, variant=None, want_traj: bool = True, timing: bool = False):
        from . import _native
        if self._handle is None:
            raise RuntimeError("setup_optimization() has not been called")
        x = np.ascontiguousarray(np.asarray(x_init, dtype=np.float64).reshape(-1, self._nx))
        r = np.ascontiguousarray(np.asarray(ref, dtype=np.float64).reshape(-1, self._nx))
        if r.shape[0] == 1 and x.shape[0] > 1:
            r = np.ascontiguousarray(np.broadcast_to(r, x.shape))
        out = _native.solve_batch(self._handle, x, r, variant, want_traj, timing=timing)
        self.last_status, self.last_iters = out["status"], out["iters"]
        if timing:
            self.last_solve_time = out["solve_time"]       # seconds per instance on the device (tmpc_set_solve_timing)
        return out

    def solve_optimization_problem(self, x_init, ref):
        """TubeTrackingMPC.py:170-194, batched over a leading axis when given one."""
        batched = self._is_batched(x_init)
        out = self._solve(x_init, ref)
        return self._unpack(out, batched, "tube tracking MPC")

    def solve(self, x_k, ref):
        """Alias of `solve_optimization_problem` under the name BASELINE.json's north_star uses (`.solve(x_k, ref)`); the
        reference itself has no method of this name."""
        return self.solve_optimization_problem(x_k, ref)

    def _is_batched(self, x) -> bool:
        """(nx,) and the reference's column vector (nx,1) are single instances; (B,nx) is a batch."""
        return np.ndim(x) == 2 and np.shape(x) != (self._nx, 1)

    def _unpack(self, out, batched: bool, who: str):
        st = out["status"]
        if batched:
            return out["x_nom"], out["u_nom"], out["x_ss"], out["u_ss"]
        if st[0] != STATUS_OPTIMAL:
            print(f"Status of {who} is: {_STATUS_TEXT.get(int(st[0]), st[0])}")
        if st[0] >= STATUS_INFEASIBLE:
            return None, None, None, None
        # reference layout: states / inputs stacked column-wise
        return (out["x_nom"][0].T.copy(), out["u_nom"][0].T.copy(),
                out["x_ss"][0].copy(), out["u_ss"][0].copy())

    def determine_packet(self, x_hat, ref, q_t):
        """TubeTrackingMPC.py:196-209."""
        start = time.time()
        x_nom, u_nom, x_ss, u_ss = self.solve_optimization_problem(np.asarray(x_hat).reshape(-1), ref)
        self._computational_times.append(time.time() - start)
        return self.encapsulate(u_nom, u_ss, x_ss, q_t)

    def encapsulate(self, u_nom_traj, u_steady_state, x_steady_state, q_t):
        """U_t = [u_nom | u_ss + K x_ss]  (TubeTrackingMPC.py:211-227)."""
        if x_steady_state is not None:
            u_ss = (u_steady_state + self._K @ x_steady_state).reshape(u_nom_traj.shape[0], 1)
            U_t = np.hstack((u_nom_traj, u_ss))
        else:
            U_t = None
        return {"U_t": U_t, "q_t": q_t}

    def determine_packets(self, x_hat, ref, variant=None):
        """Batched form of `determine_packet`: returns U_t stacked as (B, nu, N+1),
        x_nom_0 as (B, nx), and per-instance status."""
        out = self._solve(x_hat, ref, variant, want_traj=False)
        u_ss = out["u_ss"] + out["x_ss"] @ self._K.T
        U = np.concatenate([out["u_nom"], u_ss[:, None, :]], axis=1)     # (B, N+1, nu)
        return np.ascontiguousarray(U.transpose(0, 2, 1)), out["x_nom0"], out["status"]

    def run_closed_loop(self, p_loss, ref, th_u=None, ga_u=None, w=None, x0=None, extended: bool = False, plant=None,
                        warm_start: bool = False, capture=None, timing: bool = False, device_rng=None, fused=None) -> dict:
        """The lossy-network closed loop of the reference's Monte-Carlo scripts (results_linear_system.py:209-291)
        for a batch of trajectories, resident on the device (include/tmpc.h: tmpc_mc_run): ONE launch for the whole sweep
        where the controller has one QP (a wavefront keeps its trajectory for all T steps: solve, state machines, solve, ...;
        fused = "on" / "off" / "auto", tmpc_mc_set_fused; out["fused"] says what ran), else one solve launch per problem
        plus one state-machine launch per time step; nothing returns to the host in between.  warm_start: every solve first
        tries the working set of the trajectory's previous solve in the exact refinement (same results, fewer iterations);
        capture: index of one trajectory whose x_t / x_nom_t / u_t are returned (the scripts' sample run, :298-301).
        timing: per trajectory the mean / maximum device time of a solve (solve_time_mean, solve_time_max, seconds); the
        means are also appended to get_computational_times(), the list the scripts take their quantiles of (:305-315).
        device_rng = (seed, first_trajectory, w_bound): draw the realisations on the device instead of taking th_u, ga_u, w
        (tmpc_mc_set_device_rng; montecarlo.draw_realisations_philox is the host twin)."""
        from . import _native
        if self._handle is None:
            raise RuntimeError("setup_optimization() has not been called")
        _native.mc_set_plant(self._handle, plant)        # None: the linear model; 'cartpole': the nonlinear cart-pole (RK4, 500 Hz)
        _native.mc_set_actuator(self._handle, self._smart_actuator)
        out = _native.mc_run(self._handle, p_loss, ref, th_u, ga_u, w, x0=x0, Z=None if self._smart_actuator else self._Z,
                             extended=extended, warm_start=warm_start, capture=capture, timing=timing,
                             physics_substeps=0 if plant in (None, "linear") else 10, device_rng=device_rng, fused=fused)
        if timing:
            self._computational_times.extend(out["solve_time_mean"].tolist())
        return out

    # ------------------------------------------------------------------ accessors
    def set_ancillary_controller_gain(self, K_ancillary):
        self._K_ancillary = np.atleast_2d(np.asarray(K_ancillary, dtype=np.float64))
        self._Acl_plant = self._A - self._B @ self._K_ancillary

    def get_ancillary_controller_gain(self):
        return self._K if self._K_ancillary is None else self._K_ancillary

    def get_steady_state_controller_gain(self):
        return self._K

    def get_computational_times(self):
        return self._computational_times

    def reset_computational_times(self):
        self._computational_times = []

    def set_device(self, device: int):
        self._device = int(device)

    def set_kernel_path(self, path: str):
        """'auto' (default) | 'wave' | 'block' -- include/tmpc.h: tmpc_set_kernel_path."""
        from . import _native
        _native.set_kernel_path(self._handle, path)

    def get_kernel_path(self, variant: int = 0) -> str:
        from . import _native
        return _native.get_kernel_path(self._handle, variant)

    def _close(self):
        if self._handle is not None:
            from . import _native
            _native.destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass


class ExtendedTubeTrackingMPC(TubeTrackingMPC):
    """Section IV.F variant (TubeTrackingMPC.py:249-369): a second QP with the
    initial-state constraint Z (-) W is used whenever the previous plant packet
    arrived (`gamma_t == 1`)."""

    _VARIANTS = 2

    def generate_optimization_problem_when_packet_received(self, W):
        """TubeTrackingMPC.py:253-299; re-creates the device problem with both variants."""
        self._ZmW = up.pont_diff(self._Z, as_polytope(W))
        self.generate_optimization_problem(self._fixed_initial_state)

    def setup_optimization(self, W, fixed_initial_state: bool = False, rpi_method: int = 0):
        self._W = as_polytope(W)
        self.determine_mRPI(self._W, rpi_method=rpi_method)
        self.tighten_constraints()
        self.determine_Xf()
        self._fixed_initial_state = bool(fixed_initial_state)
        self.generate_optimization_problem_when_packet_received(self._W)

    def _problem_dict(self) -> dict:
        d = super()._problem_dict()
        if self._ZmW is not None:
            d["extended"] = 1
            d["HZW"], d["hZW"] = self._ZmW.A, self._ZmW.b
            if self._eliminate_auxiliaries:
                # TubeTrackingMPC.py:293 puts the terminal rows on free variables of the other problem; what
                # remains for this problem is x_bar in proj(Xf), computed once here (include/tmpc.h: HTP)
                if self._XfP is None:
                    self._XfP = up.eliminate_terminal_auxiliaries(self._Xf, self._A, self._B)
                d["HTP"], d["hTP"] = self._XfP.A, self._XfP.b
        return d

    def solve_optimization_problem(self, x_init, ref, gamma_t=0):
        """TubeTrackingMPC.py:307-349; `gamma_t` may be an array of 0/1 per instance."""
        batched = self._is_batched(x_init)
        g = np.asarray(gamma_t, dtype=np.uint8).reshape(-1)
        nb = np.asarray(x_init).reshape(-1, self._nx).shape[0]
        if g.size == 1 and nb > 1:
            g = np.full(nb, g[0], dtype=np.uint8)
        out = self._solve(x_init, ref, np.ascontiguousarray(g))
        who = "extended tube MPC when packet has been received" if g[0] == 1 else \
            "extended tube MPC when packet has not been received"
        return self._unpack(out, batched, who)

    def determine_packet(self, x_hat, ref, q_t, gamma_t=0):
        """TubeTrackingMPC.py:351-369."""
        start = time.time()
        x_nom, u_nom, x_ss, u_ss = self.solve_optimization_problem(np.asarray(x_hat).reshape(-1), ref, gamma_t)
        self._computational_times.append(time.time() - start)
        packet = self.encapsulate(u_nom, u_ss, x_ss, q_t)
        x_nom_0 = None if x_nom is None else x_nom[:, 0]
        packet["x_nom_0"] = x_nom_0
        return packet, x_nom_0


# BASELINE.json's north_star speaks of "TubeTrackingMPCOverLossyNet"; the reference has no such class (the lossy-network loop is a
# script-level composition of MPC + Estimator + ConsistentActuator).  The name is provided as an alias.
TubeTrackingMPCOverLossyNet = TubeTrackingMPC
