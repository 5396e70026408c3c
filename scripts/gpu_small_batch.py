"""Developer script: kernel time of SMALL batches (1 ... 512 instances: the reference's own experiment sizes) through the
wave-per-QP and the workgroup-per-QP kernel, cart-pole N = 10 and N = 20 (base and packet-received problem)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
if os.environ.get("TMPC_LIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["TMPC_LIB"])
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
rng = np.random.default_rng(0)
for N, ext in ((10, False), (20, False), (20, True)):
    mpc, w = common.make_mpc("cartpole", N, True, extended=ext, create=True)
    for path in ("wave", "block"):
        mpc.set_kernel_path(path)
        row = []
        for B in (1, 16, 200, 512):
            idx = rng.integers(0, len(S), B)
            X, R = S[idx, :4].copy(), S[idx, 4:].copy()
            G = np.ones(B, np.uint8) if ext else None
            if ext:
                X = X + rng.uniform(-1, 1, X.shape) * 0.5 * w["w_bound"]
            ms = []
            for _ in range(5):
                o = mpc._solve(X, R, G, want_traj=False)
                ms.append(_native.last_kernel_ms(mpc._handle))
            row.append("B=%d: %.3f ms (iters %.1f)" % (B, min(ms[1:]), o["iters"].mean()))
        print("N=%d %s %-5s  %s" % (N, "packet-received" if ext else "base", path, "   ".join(row)), flush=True)
