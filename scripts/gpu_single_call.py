"""Developer script: wall time of ONE determine_packet call at batch 1 (and a few more sizes) through tmpc_solve_batch -- the reference's own
timing table (results_linear_system.py:305-315) -- for the cart-pole at N = 10 and N = 20.  (Used for the A/B of a zero-copy path -- the kernel
reading / writing the pinned staging block directly for <= 16 instances instead of two DMA operations: 0.165 -> 0.155 ... 0.169 ms median at batch 1
over two runs on one box, i.e. inside the run-to-run spread; dropped.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
for N in (10, 20):
    mpc, w = common.make_mpc("cartpole", N, True, create=True)
    for B in (1, 8, 16, 32):
        ii = np.random.default_rng(11).integers(0, len(S) - B, 600)
        for k in ii[:30]:
            mpc._solve(S[k:k + B, :4], S[k:k + B, 4:], want_traj=False)
        t = []
        for k in ii:
            t0 = time.perf_counter()
            mpc._solve(S[k:k + B, :4], S[k:k + B, 4:], want_traj=False)
            t.append(time.perf_counter() - t0)
        t = 1e3 * np.array(t)
        print(f"N={N} B={B:3d}: median {np.median(t):.4f} ms  mean {t.mean():.4f}  q95 {np.quantile(t, .95):.4f}  max {t.max():.4f}", flush=True)
