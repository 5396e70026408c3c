"""Developer script: the offline stage (mRPI, tightening, terminal set, Z (-) W) timed with the batched LP kernel
and with scipy's HiGHS (one call per LP, what the reference does) on the same host, plus the raw LP rate."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native, polytope_lite as pl, utils_polytope as up
from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC


def offline(name, method):
    w = common.workload(name)
    mpc = TubeTrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
    mpc.set_input_constraints(w["U"]); mpc.set_state_constraints(w["X"])
    t = [time.time()]
    mpc.determine_mRPI(w["W"], rpi_method=method); t.append(time.time())
    mpc.tighten_constraints(); t.append(time.time())
    mpc.determine_Xf(verbose=False); t.append(time.time())
    up.pont_diff(mpc._Z, w["W"]); t.append(time.time())
    return mpc, np.diff(t)


_native.lp_batch(np.eye(2), np.ones(2), np.ones((1, 2)))      # context creation is not part of the timing
for name, method in (("double_integrator", 1), ("cartpole", 1), ("synthetic", 1)):
    res = {}
    for be in ("hip", "scipy"):
        pl.set_lp_backend(be)
        mpc, dt = offline(name, method)
        res[be] = (mpc, dt)
        print(f"{name}: backend {be:5s} mRPI {dt[0]:7.3f}s tighten {dt[1]:6.3f}s Xf {dt[2]:7.3f}s Z-W {dt[3]:6.3f}s total {dt.sum():7.3f}s"
              f"  rows Z {mpc._Z.A.shape[0]} Xf {mpc._Xf.A.shape[0]}", flush=True)
    a, b = res["hip"][0], res["scipy"][0]
    same = all(p.A.shape == q.A.shape and np.allclose(p.A, q.A, atol=1e-9) and np.allclose(p.b, q.b, atol=1e-8)
               for p, q in ((a._Z, b._Z), (a._Xf, b._Xf), (a._Xc, b._Xc), (a._Uc, b._Uc)))
    print(f"   sets equal row for row: {same}; speed-up {res['scipy'][1].sum() / res['hip'][1].sum():.1f}x")
    if name == "cartpole":
        cart = a

# raw rate: the redundancy batch of the cartpole RPI (3080 LPs, 3080 rows, d = 4) and random directions on Xf (d = 9)
w = common.workload("cartpole")
mpc = TubeTrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
K = mpc._K
rpi, _ = up.calculate_RPI(mpc._Acl, w["W"], w["X"], w["U"], K, 1e-4, 2000, verbose=False)
P = pl.Polytope(rpi.A, rpi.b, normalize=True)
for tag, A, b, Cm, rel in (("reduce batch RPI 3080x4", P.A, P.b, P.A, np.arange(len(P.b), dtype=np.int32)),
                           ("random dirs Xf 420x9", cart._Xf.A, cart._Xf.b,
                            np.random.default_rng(0).standard_normal((65536, 9)), None)):
    for _ in range(2):
        t0 = time.time()
        out = _native.lp_batch(A, b, Cm, relax=rel, relax_by=1.0)
        dt = time.time() - t0
    print(f"{tag}: B={len(Cm)} {dt * 1e3:.1f} ms incl. transfers -> {len(Cm) / dt:.3e} LP/s; iters mean {out['iters'].mean():.1f};"
          f" status {np.bincount(out['status'], minlength=5)}")
