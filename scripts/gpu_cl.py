"""Developer script: where the time of the device-resident closed loop goes (solve launches vs the rest), cold and warm."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from LinearMPCOverNetworks import _native, workloads, montecarlo
N = int(os.environ.get("N", 10)); ext = bool(int(os.environ.get("EXT", 0)))
mpc, w = workloads.make_controller("cartpole", N, True, extended=ext)
B, T = int(os.environ.get("B", 4096)), 100
th, ga, wd = montecarlo.draw_realisations(B, T, w["w_bound"], seed=99)
pl = np.full(B, 0.3)
ref = np.where(np.arange(T) < T // 2, 0.5, -0.5)
mpc.run_closed_loop(pl[:64], ref, th[:64], ga[:64], wd[:64], extended=ext)
for warm in (False, True, False, True):
    _native.kernel_ms_total(mpc._handle, reset=True)
    t0 = time.perf_counter()
    cl = mpc.run_closed_loop(pl, ref, th, ga, wd, extended=ext, warm_start=warm)
    dt = time.perf_counter() - t0
    ms, n = _native.kernel_ms_total(mpc._handle, reset=True)
    print(f"warm={warm}: wall {dt*1e3:.1f} ms, solve launches {n}: {ms:.1f} ms total ({ms/n*1e3:.0f} us each), rest {dt*1e3-ms:.1f} ms; iters/solve {cl['iters_mean']:.2f}; steps/s {B*T/dt:.3e}")
