"""Developer script: the batched LP kernel (tmpc_lp_batch) against scipy's HiGHS on the cartpole sets."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common  # noqa: F401  (path set-up)
from LinearMPCOverNetworks import _native, polytope_lite as pl

S = dict(np.load(os.path.join(common.GOLDEN, "cartpole_sets.npz")))
rng = np.random.default_rng(0)
for name in ["Z", "Xf"]:
    A, b = S[name + "_A"], S[name + "_b"]
    nr, d = A.shape
    n = 200
    Cm = np.r_[rng.standard_normal((n, d)), A[rng.integers(0, nr, n)], A[rng.integers(0, nr, n)]]
    ridx = rng.integers(0, nr, n)
    Cm[2 * n:] = A[ridx]
    rel = np.r_[np.full(2 * n, -1), ridx].astype(np.int32)
    t0 = time.time()
    out = _native.lp_batch(A, b, Cm, relax=rel, relax_by=1.0, want_x=True)
    t1 = time.time()
    out = _native.lp_batch(A, b, Cm, relax=rel, relax_by=1.0, want_x=True)
    t2 = time.time()
    ref = np.empty(len(Cm))
    for i, (c, r) in enumerate(zip(Cm, rel)):
        b2 = b.copy()
        if r >= 0:
            b2[r] += 1.0
        ref[i], _ = pl._lp_max(c, A, b2)
    t3 = time.time()
    err = np.abs(out["val"] - ref) / np.maximum(np.abs(ref), 1.0)
    feas = max(float(np.max(A @ x - (b + (np.arange(nr) == r) * 1.0))) for x, r in zip(out["x"], rel))
    print(f"{name} {A.shape}: B={len(Cm)} status {np.bincount(out['status'], minlength=5)} iters mean {out['iters'].mean():.1f} max {out['iters'].max()}"
          f" max rel err {err.max():.3e} (random {err[:n].max():.1e} rows {err[n:2*n].max():.1e} relaxed {err[2*n:].max():.1e}) max row violation {feas:.2e}")
    print(f"   first call {t1 - t0:.3f}s second {t2 - t1:.3f}s scipy {t3 - t2:.3f}s")
