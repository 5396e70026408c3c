import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
from oracle.oracle import Oracle
S = common.harvest_states("cartpole", 20, True, [[0.5, 0.0], [3.0, 0.0], [-2.0, 1.0]], steps=60)
mpc, w = common.make_mpc("cartpole", 20, True, create=True)
print("dims", _native.get_dims(mpc._handle))
orc = Oracle(mpc._problem_dict())
ref = orc.solve(S[:, :4], S[:, 4:])
out = mpc._solve(S[:, :4], S[:, 4:])
print("status hip", np.bincount(out["status"], minlength=4), "oracle", np.bincount(ref["status"], minlength=4), "iters", out["iters"].mean(), ref["iters"].mean())
ok = (out["status"] == 0) & (ref["status"] == 0)
print("max |u_nom - oracle|", np.abs(out["u_nom"] - ref["u_nom"])[ok].max(), "xu_ss", np.abs(out["xu_ss"] - ref["xu_ss"])[ok].max())
idx = np.random.default_rng(0).integers(0, len(S), 4096)
X, R = S[idx, :4].copy(), S[idx, 4:].copy()
for _ in range(2):
    mpc._solve(X, R, want_traj=False)
    ms = _native.last_kernel_ms(mpc._handle)
    print("B=4096 N=20 kernel ms", ms, "->", 4096 / ms * 1e3, "solves/s")

for path in ("block", "wave"):
    mpc.set_kernel_path(path)
    out2 = mpc._solve(S[:, :4], S[:, 4:])
    print(path, "max |u - oracle|", np.abs(out2["u_nom"] - ref["u_nom"]).max())
    for _ in range(2):
        mpc._solve(X, R, want_traj=False)
    print("B=4096 N=20", path, "kernel ms", _native.last_kernel_ms(mpc._handle))
