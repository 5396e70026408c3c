"""Tube-based tracking MPC on the disturbed double integrator -- the scenario of the reference's
"Examples of Model Predictive Controllers/Example_of_Tube_Tracking_MPC.py" (BASELINE configs[0]) run through this
package: same class, same calls, the QP of every time step solved on the MI355X.

    python examples/tube_tracking_mpc.py [--N 10]

Prints what the reference script plots: input / state ranges against their constraint sets and the tracking of the
piecewise-constant reference (5, -9, 9, 4; -9 and 9 lie outside the state constraints, the controller settles at the
closest admissible steady state)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC          # noqa: E402
from LinearMPCOverNetworks.polytope_lite import Polytope                   # noqa: E402  (stands in for polytope.Polytope)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=10)
    args = ap.parse_args()
    rng_w = np.random.default_rng(1)
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([[0.0], [1.0]])
    nx, nu = 2, 1
    X = Polytope(np.r_[np.eye(nx), -np.eye(nx)], 8.0 * np.ones(2 * nx))
    U = Polytope(np.array([[1.0], [-1.0]]), np.ones(2 * nu))
    W = Polytope(np.r_[np.eye(nx), -np.eye(nx)], 0.1 * np.ones(2 * nx))

    mpc = TubeTrackingMPC(A, B, np.eye(nx), np.eye(nu), args.N)
    mpc.set_input_constraints(U)
    mpc.set_state_constraints(X)
    mpc.setup_optimization(W)             # offline stage (batched LP kernel) + device QP; x_0 is a decision variable
    K = mpc.get_ancillary_controller_gain()

    T = 120
    ref = np.r_[5.0 * np.ones(30), -9.0 * np.ones(30), 9.0 * np.ones(30), 4.0 * np.ones(30)]
    x = np.array([1.0, 2.0])
    xs, us, in_tube = [x.copy()], [], 0
    for t in range(T):
        x_nom, u_nom, x_ss, u_ss = mpc.solve_optimization_problem(x, np.array([ref[t], 0.0]))
        u = u_nom[:, 0] - K @ (x - x_nom[:, 0])
        if u.reshape(-1, 1) not in U:
            print(f"Input constraints violated at t = {t} with input u = {u}")
        in_tube += (x - x_nom[:, 0]) in mpc._Z
        x = A @ x + B @ u + rng_w.uniform(-0.1, 0.1, nx)
        xs.append(x.copy())
        us.append(u.copy())
    xs, us = np.array(xs), np.array(us)
    print(f"N = {args.N}: {T} steps")
    print(f"  u in [{us.min():+.3f}, {us.max():+.3f}] (U = +-1);  x1 in [{xs[:, 0].min():+.3f}, {xs[:, 0].max():+.3f}],"
          f" x2 in [{xs[:, 1].min():+.3f}, {xs[:, 1].max():+.3f}] (X = +-8)")
    print(f"  x - x_nom_0 inside the tube Z in {in_tube} of {T} steps")
    for k, (a, b) in enumerate(((0, 30), (30, 60), (60, 90), (90, 120))):
        print(f"  reference {ref[a]:+.0f}: x1 at the end of the segment = {xs[b, 0]:+.4f}")


if __name__ == "__main__":
    main()
