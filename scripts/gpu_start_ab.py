"""Developer script: per-instance device times, iterations, re-runs and refinement rounds of the bench problem for the libraries
given on the command line (diagnostic builds with -DTMPC_ITERS_TOTAL fold re-runs / rounds into `iters`)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
N, ext = int(sys.argv[2]), int(sys.argv[3])
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", N, True, extended=bool(ext), create=True)
rng = np.random.default_rng(0)
idx = rng.integers(0, len(S), 4096)
X, R = S[idx, :4].copy(), S[idx, 4:].copy()
G = None
if ext:
    X = X + rng.uniform(-1, 1, X.shape) * 0.5 * w["w_bound"]
    G = np.ones(len(X), np.uint8)
for _ in range(2):
    o = mpc._solve(X, R, G, want_traj=False, timing=True)
it = o["iters"]; tm = o["solve_time"] * 1e6
total = it %% 100; reruns = (it %% 10000) // 100; rounds = (it %% 1000000) // 10000
ms = []
for _ in range(4):
    mpc._solve(X, R, G, want_traj=False); ms.append(_native.last_kernel_ms(mpc._handle))
print("%%-60s N=%%d ext=%%d kernel ms %%.3f | iters mean %%.2f max %%d | re-runs %%s | rounds hist %%s | time us mean %%.1f q99 %%.1f max %%.1f | status %%s" %% (
    os.path.basename(sys.argv[1]), N, ext, min(ms), total.mean(), total.max(), np.bincount(reruns).tolist(), np.bincount(rounds).tolist(),
    tm.mean(), np.quantile(tm, .99), tm.max(), np.bincount(o["status"]).tolist()), flush=True)
slow = np.argsort(-tm)[:6]
print("   slowest:", [(int(tm[i]), int(total[i]), int(reruns[i]), int(rounds[i])) for i in slow])
''' % ROOT
for lib in sys.argv[1:]:
    for N, ext in ((10, 0), (20, 0), (20, 1)):
        subprocess.call([sys.executable, "-c", CODE, lib, str(N), str(ext)])
