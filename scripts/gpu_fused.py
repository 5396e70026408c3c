"""Developer script: the fused closed loop (one launch for all T steps, tmpc_mc_set_fused) against the launch pair per step: same
numbers bit for bit, and the time of both.  python scripts/gpu_fused.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import montecarlo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
KEYS = ("err2", "tube_violations", "not_optimal", "x_final", "consistent", "iters_sum")
for name, N, fixed in (("cartpole", 10, True), ("cartpole", 20, True), ("double_integrator", 10, False)):
    mpc, w = common.make_mpc(name, N, fixed, create=True)
    th, ga, wd = montecarlo.draw_realisations(B, T, w["w_bound"], seed=99)
    pl = np.full(B, 0.3)
    ref = np.where(np.arange(T) < T // 2, 0.5, -0.5)
    mpc.run_closed_loop(pl[:64], ref, th[:64], ga[:64], wd[:64])
    for warm in (False, True):
        res = {}
        for mode in ("off", "on", "off", "on"):
            t0 = time.perf_counter()
            cl = mpc.run_closed_loop(pl, ref, th, ga, wd, warm_start=warm, fused=mode)
            dt = time.perf_counter() - t0
            assert cl["fused"] == (mode == "on"), (mode, cl["fused"])
            res[mode] = cl
            print(f"{name} N={N} warm={warm} fused={mode}: wall {dt*1e3:8.1f} ms = {B*T/dt:.3e} steps/s; iters/solve {cl['iters_mean']:.2f}; "
                  f"not optimal {int(cl['not_optimal'].sum())}", flush=True)
        same = all(np.array_equal(res["on"][k], res["off"][k], equal_nan=True) for k in KEYS)
        print(f"   fused == per-step, bit for bit: {same}", flush=True)
        if not same:
            for k in KEYS:
                a, b = np.asarray(res["on"][k], float), np.asarray(res["off"][k], float)
                print("     ", k, float(np.nanmax(np.abs(a - b))))
