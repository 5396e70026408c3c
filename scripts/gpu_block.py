"""Developer script: the workgroup-per-QP kernel (csrc/tmpc_block.hip) against the oracle / the wave kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
from oracle.oracle import Oracle

def cmp(tag, out, gold):
    ok = (out["status"] == 0) & (gold["status"] == 0)
    print(tag, "status hip", np.bincount(out["status"], minlength=4), "oracle", np.bincount(gold["status"], minlength=4),
          "iters", out["iters"].mean(), "max|du|", np.abs(out["u_nom"] - gold["u_nom"])[ok].max() if ok.any() else None,
          "max|dss|", np.abs(out["xu_ss"] - gold["xu_ss"])[ok].max() if ok.any() else None, flush=True)

which = sys.argv[1:] or ["n10", "n20", "ext", "syn"]
if "n10" in which:
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
    gold = np.load(os.path.join(common.GOLDEN, "cartpole_N10_oracle.npz"))
    mpc, w = common.make_mpc("cartpole", 10, True, create=True)
    mpc.set_kernel_path("block"); print("path", mpc.get_kernel_path())
    out = mpc._solve(S[:, :4], S[:, 4:])
    cmp("N10 block vs golden", out, gold)
    idx = np.random.default_rng(0).integers(0, len(S), 4096)
    X, R = S[idx, :4].copy(), S[idx, 4:].copy()
    for p in ("block", "wave"):
        mpc.set_kernel_path(p)
        for _ in range(2):
            mpc._solve(X, R, want_traj=False)
        print("N10 B=4096", p, "kernel ms", _native.last_kernel_ms(mpc._handle), flush=True)
if "n20" in which:
    S = common.harvest_states("cartpole", 20, True, [[0.5], [-0.4, 0.3], [0.2, -0.5, 0.1], [1.0]], 32, seed=3, disturb=True)
    mpc, w = common.make_mpc("cartpole", 20, True, create=True)
    orc = Oracle(mpc._problem_dict())
    gold = orc.solve(S[:, :4], S[:, 4:]); gold["xu_ss"] = np.c_[gold["x_ss"], gold["u_ss"]]
    for p in ("wave", "block"):
        mpc.set_kernel_path(p)
        out = mpc._solve(S[:, :4], S[:, 4:])
        cmp("N20 %s vs oracle" % p, out, gold)
    idx = np.random.default_rng(0).integers(0, len(S), 4096)
    X, R = S[idx, :4].copy(), S[idx, 4:].copy()
    for p in ("block", "wave"):
        mpc.set_kernel_path(p)
        for _ in range(2):
            mpc._solve(X, R, want_traj=False)
        print("N20 B=4096", p, "kernel ms", _native.last_kernel_ms(mpc._handle), flush=True)
if "ext" in which:
    for N in (10, 20):
        mpc, w = common.make_mpc("cartpole", N, True, extended=True, create=True)
        print("ext N", N, "dims", _native.get_dims(mpc._handle, 0), _native.get_dims(mpc._handle, 1), "paths", mpc.get_kernel_path(0), mpc.get_kernel_path(1))
        S = common.harvest_states("cartpole", N, True, [[0.5], [-0.4, 0.3], [0.2, -0.5, 0.1], [1.0]], 32, seed=4, disturb=True, extended=True)
        orc = Oracle(mpc._problem_dict())
        rng = np.random.default_rng(1)
        var = rng.integers(0, 2, len(S)).astype(np.uint8)
        gold = orc.solve(S[:, :4], S[:, 4:], var); gold["xu_ss"] = np.c_[gold["x_ss"], gold["u_ss"]]
        out = mpc._solve(S[:, :4], S[:, 4:], var)
        cmp("ext N%d vs oracle" % N, out, gold)
        print("   max |x_nom0 diff|", np.nanmax(np.abs(out["x_nom0"] - gold["x_nom0"])))
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/ext_N%d.npz" % N, S=S, var=var, u_nom=out["u_nom"], x_nom0=out["x_nom0"], xu_ss=out["xu_ss"],
                            status=out["status"], iters=out["iters"], g_u_nom=gold["u_nom"], g_x_nom0=gold["x_nom0"], g_xu_ss=gold["xu_ss"],
                            g_status=gold["status"], g_iters=gold["iters"])
        idx = rng.integers(0, len(S), 4096)
        X, R, V = S[idx, :4].copy(), S[idx, 4:].copy(), var[idx].copy()
        for _ in range(2):
            mpc._solve(X, R, V, want_traj=False)
        print("ext N%d B=4096 kernel ms" % N, _native.last_kernel_ms(mpc._handle), flush=True)
if "syn" in which:
    mpc, w = common.make_mpc("synthetic", 30, True, create=True)
    print("syn dims", _native.get_dims(mpc._handle, 0), "path", mpc.get_kernel_path(0))
    orc = Oracle(mpc._problem_dict())
    rng = np.random.default_rng(0)
    B = 128
    X = rng.uniform(-0.5, 0.5, (B, 12)) * mpc._Xc.b[:12]
    R = np.zeros((B, 12)); R[:, 0] = rng.uniform(-2, 2, B)
    gold = orc.solve(X, R); gold["xu_ss"] = np.c_[gold["x_ss"], gold["u_ss"]]
    out = mpc._solve(X, R)
    cmp("syn vs oracle", out, gold)
    B = 2048
    X = rng.uniform(-0.5, 0.5, (B, 12)) * mpc._Xc.b[:12]
    R = np.zeros((B, 12)); R[:, 0] = rng.uniform(-2, 2, B)
    for _ in range(2):
        o = mpc._solve(X, R, want_traj=False)
    print("syn B=2048 kernel ms", _native.last_kernel_ms(mpc._handle), "iters", o["iters"].mean(), flush=True)
