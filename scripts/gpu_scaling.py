"""Developer script: kernel time of the bench problem as a function of the batch size (occupancy / clock study)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
if os.environ.get('TMPC_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['TMPC_LIB'])
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
rng = np.random.default_rng(0)
for B in (64, 256, 1024, 2048, 4096, 8192, 16384, 65536):
    idx = rng.integers(0, len(S), B)
    X, R = S[idx, :4].copy(), S[idx, 4:].copy()
    ts = []
    for _ in range(4):
        o = mpc._solve(X, R, want_traj=False)
        ts.append(_native.last_kernel_ms(mpc._handle))
    print("B=%6d  kernel ms %8.3f  (min of 4)  -> %.3e solves/s   mean iters %.2f" % (B, min(ts), B / min(ts) * 1e3, o["iters"].mean()), flush=True)
