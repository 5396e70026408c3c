"""Developer script: interior-point iteration counts of the wave kernel against the CPU oracle on the golden states
(instances whose refinement failed are re-run with a tighter tolerance and show up with about twice the count)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import common
from oracle.oracle import Oracle
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, _ = common.make_mpc("cartpole", 10, True, create=True, device=0)
out = mpc._solve(S[:, :4], S[:, 4:])
ref = Oracle(mpc._problem_dict()).solve(S[:, :4], S[:, 4:])
ki, oi = out["iters"], ref["iters"]
print("kernel iters", np.bincount(ki))
print("oracle iters", np.bincount(oi))
d = ki - oi
print("kernel - oracle", dict(zip(*np.unique(d, return_counts=True))))
