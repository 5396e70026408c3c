"""Developer script: successive batches of 4096 on ONE handle (one stream: every launch waits for the slowest instance of
the previous one) against two / three handles on their own streams taking turns (the tail of a launch overlaps with the
head of the next: CUs whose eight waves are done take workgroups of the next launch)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np, torch
from LinearMPCOverNetworks import _native, workloads
from bench import DeviceBatch
dev = torch.device("cuda", 0)
mpcs = [workloads.make_controller("cartpole", 10, True, device=0)[0] for _ in range(3)]
_, w = workloads.make_controller("cartpole", 10, True, device=0)
X, R, _ = workloads.harvest_closed_loop_states(mpcs[0], w, 128, 32, seed=1000)
rng = np.random.default_rng(0)
batches = [[DeviceBatch(torch, dev, X[p], R[p], None, 10, 1) for p in (rng.permutation(len(X)) for _ in range(8))] for _ in range(3)]
for nh in (1, 2, 3):
    hs = [m._handle for m in mpcs[:nh]]
    for rep in range(2):
        for k in range(16):
            batches[k % nh][k % 8].solve(_native, hs[k % nh])
        for h in hs: _native.synchronize(h)
        K = 200
        t0 = time.perf_counter()
        for k in range(K):
            batches[k % nh][k % 8].solve(_native, hs[k % nh])
        for h in hs: _native.synchronize(h)
        dt = time.perf_counter() - t0
    print(f"{nh} handle(s): {dt / K * 1e3:.4f} ms per batch of 4096 -> {4096 * K / dt:.3e} solves/s")
