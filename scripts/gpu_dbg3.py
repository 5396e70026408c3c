"""Developer script: reproduce tests/test_kkt_all_configs.py::test_extended_packet_received_problem[20] inputs, find the
instances that are neither optimal nor infeasible, re-run them alone (optionally with the debug-print build)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
if len(sys.argv) > 1 and sys.argv[1] == "dbg":
    _native.LIB_PATH = os.path.join(common.PKG, "lib", "libtmpc_dbg.so")
import test_kkt_all_configs as T
N = 20
S = T.S
mpc, w = common.make_mpc("cartpole", N, True, extended=True, create=True)
rng = np.random.default_rng(3 + N)
idx = rng.choice(len(S), 192, replace=False)
Xb = T.boundary_states(rng, mpc._Xc.b[:4] * np.array([0.5, 0.4, 0.9, 0.5]), 128, 0.7)
X = np.r_[S[idx, :4], Xb]
X = X + rng.uniform(-1, 1, X.shape) * w["w_bound"] * 3.0
R = np.r_[S[idx, 4:], np.c_[rng.uniform(-2, 2, len(Xb)), np.zeros((len(Xb), 3))]]
gam = (rng.uniform(size=len(X)) < 0.75).astype(np.uint8)
if len(sys.argv) > 2:
    bad = np.array([int(a) for a in sys.argv[2:]])
else:
    out = mpc._solve(X, R, gam)
    bad = np.flatnonzero((out["status"] != 0) & (out["status"] != 2))
    print("bad", bad, out["status"][bad], out["iters"][bad], gam[bad], flush=True)
from oracle.oracle import Oracle
orc = Oracle(mpc._problem_dict())
ref = orc.solve(X[bad], R[bad], gam[bad])
print("oracle", ref["status"], ref["iters"])
for k in bad:
    o = mpc._solve(X[k:k + 1], R[k:k + 1], gam[k:k + 1])
    print("alone", k, o["status"], o["iters"], flush=True)
