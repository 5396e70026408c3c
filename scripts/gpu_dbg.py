"""Developer script: one instance through a debug build (lib/libtmpc_dbg.so, -DTMPC_DEBUG_PRINT)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
_native.LIB_PATH = os.path.join(common.PKG, "lib", "libtmpc_dbg.so")
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
for i in [int(a) for a in sys.argv[1:]]:
    out = mpc._solve(S[i:i + 1, :4], S[i:i + 1, 4:])
    print("instance", i, "status", out["status"], "iters", out["iters"], flush=True)
