"""Generates tests/golden/cartpole_N10_states.npy: 600 (x_k, ref) pairs visited by the
nominal closed loop of the cartpole tube MPC (N=10, fixed initial state) for step
references, harvested with the CPU oracle, plus their oracle solutions
(cartpole_N10_oracle.npz).  Inputs and expected outputs only -- no code travels.

    python tests/golden/make_states.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import common  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

if __name__ == "__main__":
    refs = [[0.5, 0.0], [3.0, 0.0], [-2.0, 0.0], [0.5, -0.5], [4.0, 4.0], [1.0, 2.0, -1.0]]
    S = common.harvest_states("cartpole", 10, True, refs, steps=100)
    np.save(os.path.join(HERE, "cartpole_N10_states.npy"), S)
    mpc, _ = common.make_mpc("cartpole", 10, True)
    sol = Oracle(mpc._problem_dict()).solve(S[:, :4], S[:, 4:])
    np.savez_compressed(os.path.join(HERE, "cartpole_N10_oracle.npz"),
                        u_nom=sol["u_nom"], x_nom0=sol["x_nom0"], xu_ss=sol["xu_ss"], status=sol["status"])
    print(S.shape, "status", np.bincount(sol["status"], minlength=4), "iters mean", sol["iters"].mean(), "max", sol["iters"].max())
