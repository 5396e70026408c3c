import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
if len(sys.argv) > 1 and sys.argv[1] == "dbg":
    _native.LIB_PATH = os.path.join(common.PKG, "lib", "libtmpc_dbg.so")
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
for sel in ([36], [36, 39], list(range(32, 40)), list(range(0, 64)), list(range(600))):
    out = mpc._solve(S[sel, :4], S[sel, 4:])
    bad = np.flatnonzero(out["status"] != 0)
    print("batch", len(sel), "bad", [sel[k] for k in bad], out["status"][bad], out["iters"][bad], flush=True)
