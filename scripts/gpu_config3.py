"""Developer script: BASELINE config 3 -- cartpole N = 20, ExtendedTubeTrackingMPC, batch 65536, gamma per instance."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
if os.environ.get('TMPC_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['TMPC_LIB'])
mpc, w = common.make_mpc("cartpole", 20, True, extended=True, create=True)
if len(sys.argv) > 1:
    mpc.set_kernel_path(sys.argv[1])          # "wave" / "block": developer knob, both problems through that kernel
SX = common.harvest_states("cartpole", 20, True, [[0.5], [-0.4, 0.3], [0.2, -0.5, 0.1]], 60, seed=4, disturb=False, extended=True)
rng = np.random.default_rng(0)
B = 65536
idx = rng.integers(0, len(SX), B)
X = SX[idx, :4] + rng.uniform(-1, 1, (B, 4)) * 0.2 * w["w_bound"]
R = SX[idx, 4:].copy()
for frac in (0.0, 0.3, 0.7, 1.0):
    gam = (rng.uniform(size=B) < frac).astype(np.uint8)
    for _ in range(2):
        o = mpc._solve(X, R, gam, want_traj=False)
    ms = _native.last_kernel_ms(mpc._handle)
    print("gamma=1 fraction %.1f: %.1f ms for %d solves -> %.3e solves/s; status %s; mean iters g0 %.1f g1 %.1f" % (
        frac, ms, B, B / ms * 1e3, np.bincount(o["status"], minlength=4), o["iters"][gam == 0].mean() if (gam == 0).any() else 0,
        o["iters"][gam == 1].mean() if (gam == 1).any() else 0), flush=True)
