"""Developer script: random models end to end.  For each random stable (A, B): offline stage with the batched LP kernel,
device QP, a batch of states -> compare with the CPU oracle (status, u_nom, steady state).  Exercises kernel shapes and
set geometries that none of the named workloads has."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import common
from oracle.oracle import Oracle
from LinearMPCOverNetworks import polytope_lite as pl
from LinearMPCOverNetworks.polytope_lite import box2poly
from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC, ExtendedTubeTrackingMPC

pl.set_lp_backend("hip")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
long_horizon = len(sys.argv) > 3 and sys.argv[3] == "long"       # aim at the largest wave-kernel shape (24 variables)
force_block = len(sys.argv) > 3 and sys.argv[3] == "block"       # every case through the workgroup-per-QP kernel
worst = 0.0
for case in range(ncase):
    n = int(rng.integers(2, 9)); m = int(rng.integers(1, 3)); N = int(rng.integers(3, 27 if m == 1 else 16))
    fixed = bool(rng.integers(0, 2)); ext = bool(rng.integers(0, 2)) and n <= 4
    if long_horizon:
        n, m, N, fixed = int(rng.integers(3, 5)), 1, int(rng.integers(15, 24)), True
    A0 = rng.standard_normal((n, n)); A = rng.uniform(0.7, 1.05) * A0 / np.max(np.abs(np.linalg.eigvals(A0)))
    Bm = rng.standard_normal((n, m))
    Q = np.diag(rng.uniform(0.5, 5.0, n)); R = np.diag(rng.uniform(0.05, 1.0, m))
    xb = rng.uniform(3.0, 10.0, n); ub = rng.uniform(0.5, 2.0, m); wb = rng.uniform(0.002, 0.02, n)
    t0 = time.time()
    try:
        cls = ExtendedTubeTrackingMPC if ext else TubeTrackingMPC
        mpc = cls(A, Bm, Q, R, N)
        mpc.set_input_constraints(box2poly(np.c_[-ub, ub])); mpc.set_state_constraints(box2poly(np.c_[-xb, xb]))
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            mpc.setup_optimization(box2poly(np.c_[-wb, wb]), fixed_initial_state=fixed, rpi_method=1)
    except Exception as e:      # set-up can legitimately fail (constraints too tight for the tube, kernel shape not covered)
        print(f"case {case}: n={n} m={m} N={N} fixed={fixed} ext={ext}: set-up refused: {type(e).__name__}: {str(e)[:100]}")
        continue
    tset = time.time() - t0
    Bsz = 96
    X = rng.uniform(-0.6, 0.6, (Bsz, n)) * mpc._Xc.b[:n]
    X[:24] *= 1.5
    Rf = np.zeros((Bsz, n)); Rf[:, 0] = rng.uniform(-0.5, 0.5, Bsz) * xb[0]
    var = rng.integers(0, 2, Bsz).astype(np.uint8) if ext else None
    orc = Oracle(mpc._problem_dict())
    ref = orc.solve(X, Rf, var) if ext else orc.solve(X, Rf)
    if force_block:
        mpc.set_kernel_path("block")
    out = mpc._solve(X, Rf, var)
    same = np.array_equal(out["status"], ref["status"])
    ok = (ref["status"] == 0) & (out["status"] == 0)
    du = float(np.max(np.abs(out["u_nom"][ok] - ref["u_nom"][ok]))) if ok.any() else 0.0
    dss = float(np.max(np.abs(out["xu_ss"][ok] - ref["xu_ss"][ok]))) if ok.any() else 0.0
    worst = max(worst, du)
    from LinearMPCOverNetworks import _native
    nv, nc, npar = _native.get_dims(mpc._handle)
    print(f"case {case}: n={n} m={m} N={N} fixed={fixed} ext={ext} rows Z {mpc._Z.A.shape[0]} Xf {mpc._Xf.A.shape[0]} nv {nv} nc {nc} path {mpc.get_kernel_path()}"
          f" set-up {tset:.2f}s status hip {np.bincount(out['status'], minlength=4)} oracle {np.bincount(ref['status'], minlength=4)} same {same}"
          f" max|du| {du:.2e} max|dss| {dss:.2e}", flush=True)
print("worst |du|", worst)
