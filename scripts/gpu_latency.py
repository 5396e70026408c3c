"""Developer script: per-instance kernel latency distribution (B=1 launches) on the GPU box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
h = mpc._handle
mpc._solve(S[:8, :4], S[:8, 4:], want_traj=False)
lat = []; its = []
for i in range(0, len(S), 3):
    o = mpc._solve(S[i:i+1, :4].copy(), S[i:i+1, 4:].copy(), want_traj=False)
    lat.append(_native.last_kernel_ms(h)); its.append(int(o["iters"][0]))
lat = np.array(lat); its = np.array(its)
print("B=1 latency ms: min %.3f med %.3f mean %.3f p90 %.3f max %.3f" % (lat.min(), np.median(lat), lat.mean(), np.quantile(lat, .9), lat.max()))
for k in sorted(set(its)):
    m = its == k
    print("  iters %2d: n=%3d  mean %.3f ms  max %.3f ms  -> %.1f us/iter" % (k, m.sum(), lat[m].mean(), lat[m].max(), 1e3 * lat[m].mean() / max(k, 1)))
worst = np.argsort(-lat)[:5]
print("worst", worst * 3, lat[worst], its[worst])
for B in (4, 64, 256, 1024, 2048, 4096, 8192, 16384):
    idx = np.random.default_rng(0).integers(0, len(S), B)
    X, R = S[idx, :4].copy(), S[idx, 4:].copy()
    mpc._solve(X, R, want_traj=False)
    mpc._solve(X, R, want_traj=False)
    ms = _native.last_kernel_ms(h)
    print("B=%5d kernel %.3f ms -> %.3e solves/s" % (B, ms, B / ms * 1e3))
