"""Developer script: average kernel time of successive chunks of 20 back-to-back launches of the bench batch (after set-up idle time):
how long the card takes to reach its steady launch time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np, torch
from LinearMPCOverNetworks import _native, workloads
from bench import DeviceBatch
dev = torch.device("cuda", 0)
mpc, w = workloads.make_controller("cartpole", 10, True, device=0)
X, R, _ = workloads.harvest_closed_loop_states(mpc, w, 128, 32, seed=1000)
bs = [DeviceBatch(torch, dev, X[p], R[p], None, 10, 1) for p in (np.random.default_rng(2000 + k).permutation(len(X)) for k in range(8))]
h = mpc._handle
for idle in (0.5, 0.0):
    time.sleep(idle)
    out = []
    for chunk in range(12):
        _native.kernel_ms_total(h, reset=True)
        t0 = time.perf_counter()
        for k in range(20): bs[k % 8].solve(_native, h)
        _native.synchronize(h)
        dt = time.perf_counter() - t0
        ms, n = _native.kernel_ms_total(h, reset=True)
        out.append((ms / n, dt / 20 * 1e3))
    print(f"after {idle} s idle: kernel avg per chunk of 20:", " ".join(f"{a:.3f}" for a, _ in out))
    print(f"                     wall per step per chunk:  ", " ".join(f"{b:.3f}" for _, b in out))
# the same with 8 launches per chunk, every batch order once
out = []
for chunk in range(12):
    _native.kernel_ms_total(h, reset=True)
    for k in range(8): bs[k].solve(_native, h)
    _native.synchronize(h)
    ms, n = _native.kernel_ms_total(h, reset=True)
    out.append(ms / n)
print("chunks of 8 (each order once):", " ".join(f"{a:.3f}" for a in out))
