import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from LinearMPCOverNetworks import _native, workloads, montecarlo
_native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_native.__file__)), "..", "lib", "libtmpc_dbg.so")
mpc, w = workloads.make_controller("cartpole", 10, True)
B, T = 2, 40
th, ga, wd = montecarlo.draw_realisations(B, T, w["w_bound"], seed=99)
cl = mpc.run_closed_loop(np.full(B, 0.3), np.full(T, 0.5), th, ga, wd, warm_start=True)
print("iters/solve", cl["iters_mean"])
