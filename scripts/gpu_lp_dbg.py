"""Developer script: LP kernel cases that miss 1e-9 against HiGHS."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native, polytope_lite as pl
for fn, key in (("synthetic_sets.npz", "Xf"), ("cartpole_sets.npz", "Xf"), ("cartpole_sets.npz", "Z")):
    s = dict(np.load(os.path.join(common.GOLDEN, fn)))
    A, b = s[key + "_A"], s[key + "_b"]
    rng = np.random.default_rng(0)
    n = 60
    ridx = rng.integers(0, len(b), n)
    Cm = np.r_[rng.standard_normal((n, A.shape[1])), A[rng.integers(0, len(b), n)], A[ridx]]
    rel = np.r_[np.full(2 * n, -1), ridx].astype(np.int32)
    out = _native.lp_batch(A, b, Cm, relax=rel, relax_by=1.0, want_x=True)
    for i, (c, r) in enumerate(zip(Cm, rel)):
        b2 = b.copy()
        if r >= 0:
            b2[r] += 1.0
        ref, _ = pl._lp_max(c, A, b2)
        e = abs(out["val"][i] - ref) / max(abs(ref), 1.0)
        if e > 1e-9 or out["status"][i] != 0:
            print(fn, key, "case", i, "rel", r, "err %.2e" % e, "status", out["status"][i], "iters", out["iters"][i],
                  "viol %.2e" % np.max(A @ out["x"][i] - b2), "val-ref %.3e" % (out["val"][i] - ref))
    print(fn, key, "iters mean", out["iters"].mean(), "status", np.bincount(out["status"], minlength=5))
