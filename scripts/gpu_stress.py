"""Developer script: large randomised batch through both builds of the wave kernel and the block kernel; statuses, agreement
between the paths, and a sampled comparison with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
from oracle.oracle import Oracle

S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
orc = Oracle(mpc._problem_dict())
rng = np.random.default_rng(2024)
B = 65536
idx = rng.integers(0, len(S), B)
X = S[idx, :4] + rng.uniform(-1, 1, (B, 4)) * 0.5 * w["w_bound"]          # perturbed closed-loop states
R = S[idx, 4:].copy()
R[:, 0] += rng.uniform(-0.5, 0.5, B)
t = time.time(); big = mpc._solve(X, R, want_traj=False); print("B=65536 two-waves build: %.1f ms kernel" % _native.last_kernel_ms(mpc._handle), "status", np.bincount(big["status"], minlength=4), "iters max", big["iters"].max())
# the same instances in chunks of 512 (one-wave-per-SIMD build) and through the block kernel (subset)
small = [mpc._solve(X[i:i + 512], R[i:i + 512], want_traj=False) for i in range(0, 8192, 512)]
u_small = np.concatenate([o["u_nom"] for o in small]); st_small = np.concatenate([o["status"] for o in small])
ok = (big["status"][:8192] == 0) & (st_small == 0)
print("one-wave vs two-waves builds: status equal", np.array_equal(big["status"][:8192], st_small), " max |du|", np.abs(big["u_nom"][:8192] - u_small)[ok].max())
mpc.set_kernel_path("block")
blk = mpc._solve(X[:4096], R[:4096], want_traj=False)
mpc.set_kernel_path("auto")
ok = (big["status"][:4096] == 0) & (blk["status"] == 0)
print("block vs wave: status equal", np.array_equal(big["status"][:4096], blk["status"]), " max |du|", np.abs(big["u_nom"][:4096] - blk["u_nom"])[ok].max())
sub = rng.choice(B, 4096, replace=False)
ref = orc.solve(X[sub], R[sub])
ok = (big["status"][sub] == 0) & (ref["status"] == 0)
print("oracle sample: status equal", np.array_equal(big["status"][sub], ref["status"]), " max |du|", np.abs(big["u_nom"][sub] - ref["u_nom"])[ok].max(),
      " max |du0|", np.abs(big["u_nom"][sub, 0] - ref["u_nom"][:, 0])[ok].max(), " oracle status", np.bincount(ref["status"], minlength=4))
bad = np.flatnonzero(big["status"][sub] != ref["status"])
if len(bad):
    print("mismatching statuses:", big["status"][sub][bad][:10], ref["status"][bad][:10], big["iters"][sub][bad][:10], ref["iters"][bad][:10])
    np.save("gpurun_out/stress_bad.npy", np.c_[X[sub][bad], R[sub][bad]])
