"""Developer script: first-light check of the HIP path against the oracle on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
from oracle.oracle import Oracle
if os.environ.get('TMPC_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['TMPC_LIB'])
    print('using', _native.LIB_PATH)

S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
print("dims", _native.get_dims(mpc._handle))
t = time.time(); out = mpc._solve(S[:, :4], S[:, 4:]); dt = time.time() - t
print("hip: wall %.3f s, kernel %.3f ms" % (dt, _native.last_kernel_ms(mpc._handle)))
print("status", np.bincount(out["status"], minlength=4), "iters mean", out["iters"].mean(), "max", out["iters"].max())
gold = np.load(os.path.join(common.GOLDEN, "cartpole_N10_oracle.npz"))
ok = (out["status"] == 0) & (gold["status"] == 0)
du = np.abs(out["u_nom"] - gold["u_nom"])[ok]
print("max |u_nom - oracle|", du.max(), " max |u0 - oracle|", np.abs(out["u_nom"][ok, 0] - gold["u_nom"][ok, 0]).max())
print("max |xu_ss - oracle|", np.abs(out["xu_ss"] - gold["xu_ss"])[ok].max())
bad = np.flatnonzero(~ok)
print("non-optimal:", bad[:20], out["status"][bad[:20]], out["iters"][bad[:20]])
worst = np.argsort(-np.abs(out["u_nom"][:, 0, 0] - gold["u_nom"][:, 0, 0]))[:5]
print("worst idx", worst, np.abs(out["u_nom"][worst, 0, 0] - gold["u_nom"][worst, 0, 0]), out["iters"][worst])
# timing at the bench batch size
B = 4096
idx = np.random.default_rng(0).integers(0, len(S), B)
X, R = S[idx, :4].copy(), S[idx, 4:].copy()
for _ in range(3):
    o = mpc._solve(X, R, want_traj=False)
    print("B=4096 kernel ms", _native.last_kernel_ms(mpc._handle), "-> %.3e solves/s" % (B / _native.last_kernel_ms(mpc._handle) * 1e3))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/check_hip_out.npz", u_nom=out["u_nom"], x_nom=out["x_nom"], xu_ss=out["xu_ss"], status=out["status"], iters=out["iters"])
