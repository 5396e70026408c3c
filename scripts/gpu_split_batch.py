"""Developer script: ONE batch of 4096 as a single launch against the same batch split into two halves on two handles (two streams)
launched together and joined -- does overlapping the halves shorten a step?  (The round-2 verdict's suggestion; the `pipelined` extra
of bench.py overlaps DIFFERENT batches.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np, torch
from LinearMPCOverNetworks import _native, workloads
from bench import DeviceBatch
dev = torch.device("cuda", 0)
mpcs = [workloads.make_controller("cartpole", 10, True, device=0)[0] for _ in range(2)]
_, w = workloads.make_controller("cartpole", 10, True, device=0)
X, R, _ = workloads.harvest_closed_loop_states(mpcs[0], w, 128, 32, seed=1000)
rng = np.random.default_rng(0)
perms = [rng.permutation(len(X)) for _ in range(8)]
full = [DeviceBatch(torch, dev, X[p], R[p], None, 10, 1) for p in perms]
halves = [(DeviceBatch(torch, dev, X[p[:2048]], R[p[:2048]], None, 10, 1), DeviceBatch(torch, dev, X[p[2048:]], R[p[2048:]], None, 10, 1)) for p in perms]
h0, h1 = mpcs[0]._handle, mpcs[1]._handle
K = 200
def run_full():
    for k in range(K):
        full[k % 8].solve(_native, h0)
        _native.synchronize(h0)                 # a step ends before the next begins
def run_split():
    for k in range(K):
        a, b = halves[k % 8]
        a.solve(_native, h0); b.solve(_native, h1)
        _native.synchronize(h0); _native.synchronize(h1)
for name, fn in (("one launch of 4096", run_full), ("two launches of 2048 on two streams", run_split), ("one launch of 4096", run_full)):
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / K * 1e3:.4f} ms per step (host-synchronised steps)")
