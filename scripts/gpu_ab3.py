"""Developer script: spread of the solve time over random orders of the same 4096 states."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from LinearMPCOverNetworks import _native, workloads
mpc, w = workloads.make_controller("cartpole", 10, True)
B = 4096
X, R, _ = workloads.harvest_closed_loop_states(mpc, w, B // 32, 32, seed=1000)
res = []
for seed in range(12):
    p = np.random.default_rng(seed).permutation(B)
    ms = []
    for _ in range(10):
        o = mpc._solve(X[p], R[p], want_traj=False)
        ms.append(_native.last_kernel_ms(mpc._handle))
    res.append(float(np.median(ms)))
print("per-order median kernel ms:", np.round(res, 3), "mean", np.mean(res))
it = o["iters"]
print("iters histogram", np.bincount(it))
