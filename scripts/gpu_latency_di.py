import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
for N in (5, 10):
    mpc, w = common.make_mpc("double_integrator", N, False, create=True)
    h = mpc._handle
    print("N", N, "dims", _native.get_dims(h))
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (4096, 2)) * [3.0, 0.5]; R = np.c_[rng.uniform(-9, 9, 4096), np.zeros(4096)]
    o = mpc._solve(X, R, want_traj=False)
    print(" status", np.bincount(o["status"], minlength=4), "iters mean", o["iters"].mean())
    for B in (1, 1024, 4096):
        mpc._solve(X[:B], R[:B], want_traj=False); mpc._solve(X[:B], R[:B], want_traj=False)
        ms = _native.last_kernel_ms(h)
        print("  B=%5d kernel %.3f ms -> %.3e solves/s; per iter %.2f us" % (B, ms, B / ms * 1e3, ms*1e3/max(o["iters"][:B].mean(),1)))
