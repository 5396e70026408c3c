"""Developer script: how well do cheap features of an instance predict its interior-point iteration count / device time,
and what would starting the predicted-hard instances first buy at batch 4096 (list scheduling over 2048 resident waves)?"""
import heapq, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from LinearMPCOverNetworks import _native, workloads
mpc, w = workloads.make_controller("cartpole", 10, True, device=0)
X, R, _ = workloads.harvest_closed_loop_states(mpc, w, 128, 32, seed=1000)
out = mpc._solve(X, R, timing=True)
it, tm = out["iters"], out["solve_time"] * 1e6
c = _native.get_condensed(mpc._handle, 0)
q = X @ c["F1"].T + R @ c["F2"].T
z = -np.linalg.solve(c["H"], q.T).T
s = c["g0"][None, :] + X @ c["E"].T - z @ c["G"].T
hn = np.maximum(np.abs(c["g0"][None, :] + X @ c["E"].T).max(axis=1), 1.0)
feats = {"nviol": (s < 0).sum(axis=1), "smin": -s.min(axis=1) / hn, "sumviol": np.maximum(-s, 0).sum(axis=1),
         "near": (s < 0.05 * hn[:, None]).sum(axis=1), "log smin": np.log1p(np.maximum(-s.min(axis=1), 0) / hn * 100)}
print("iters hist", np.bincount(it)); print(f"time us: mean {tm.mean():.1f} max {tm.max():.1f}; corr(time, iters) {np.corrcoef(tm, it)[0,1]:.3f}")
for k, f in feats.items():
    print(f"  {k:10s} corr with iters {np.corrcoef(f, it)[0,1]:+.3f}   with time {np.corrcoef(f, tm)[0,1]:+.3f}")
A = np.c_[np.ones(len(it)), feats["nviol"], feats["log smin"], feats["near"], feats["sumviol"]]
coef, *_ = np.linalg.lstsq(A, tm, rcond=None)
pred = A @ coef
print(f"linear model R^2 on time: {1 - np.var(tm - pred) / np.var(tm):.3f}")
def makespan(T, P=2048):
    h = list(T[:P]); heapq.heapify(h)
    for t in T[P:]:
        heapq.heappush(h, heapq.heappop(h) + t)
    return max(h)
rng = np.random.default_rng(0)
rnd = np.mean([makespan(tm[rng.permutation(len(tm))]) for _ in range(8)])
print(f"makespan us (durations = measured times at full occupancy): random order {rnd:.0f}, perfect LPT {makespan(np.sort(tm)[::-1]):.0f}, "
      f"by linear model {makespan(tm[np.argsort(-pred)]):.0f}, by nviol {makespan(tm[np.argsort(-feats['nviol'])]):.0f}, "
      f"by iters {makespan(tm[np.argsort(-it)]):.0f}, sum/2048 {tm.sum() / 2048:.0f}")
o = np.argsort(-tm)[:24]
print("slowest instances (us, iters):", [(int(tm[i]), int(it[i])) for i in o])
for k in sorted(set(it.tolist())):
    m = it == k
    print(f"  iters {k:2d}: n {m.sum():4d}  time mean {tm[m].mean():6.1f}  min {tm[m].min():6.1f}  max {tm[m].max():6.1f}")
# the same batch again, one instance per launch slot order reversed: is the time a property of the instance?
out2 = mpc._solve(X[::-1].copy(), R[::-1].copy(), timing=True)
tm2 = out2["solve_time"][::-1] * 1e6
print(f"corr(time, time in reversed order) {np.corrcoef(tm, tm2)[0,1]:.3f}; slowest now: {[(int(tm2[i]), int(it[i])) for i in np.argsort(-tm2)[:10]]}")
small = mpc._solve(X[:256], R[:256], timing=True)
print(f"256 instances alone (one per CU): mean {small['solve_time'].mean()*1e6:.1f} us, max {small['solve_time'].max()*1e6:.1f}, "
      f"per iteration {np.polyfit(small['iters'], small['solve_time']*1e6, 1)}")

# diagnostic build (-DTMPC_ITERS_TOTAL): iters = last run's iterations + 100 * re-runs + 10000 * refinement rounds
dbg = os.path.join(os.path.dirname(_native.__file__), "..", "lib", "libtmpc_dbg.so")
if os.path.exists(dbg):
    import subprocess
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r); from LinearMPCOverNetworks import _native, workloads; _native.LIB_PATH = %r; "
            "mpc, w = workloads.make_controller('cartpole', 10, True, device=0); X = np.load('/tmp/X.npy'); R = np.load('/tmp/R.npy'); "
            "o = mpc._solve(X, R, timing=True); np.save('/tmp/it2.npy', o['iters']); np.save('/tmp/tm2.npy', o['solve_time'])") % (
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"), dbg)
    np.save('/tmp/X.npy', X); np.save('/tmp/R.npy', R)
    subprocess.check_call([sys.executable, "-c", code])
    it2, t2 = np.load('/tmp/it2.npy'), np.load('/tmp/tm2.npy') * 1e6
    rounds, reruns = it2 // 10000, (it2 % 10000) // 100
    print("refinement rounds hist", np.bincount(rounds), "re-runs hist", np.bincount(reruns))
    for k in sorted(set(rounds.tolist())):
        m = rounds == k
        print(f"  rounds {k:2d}: n {m.sum():4d} mean time {t2[m].mean():6.1f} us, max {t2[m].max():6.1f}, re-runs {reruns[m].sum()}")
# trivially cheap proxies from (x_k, ref) alone
e = X - R
prox = {"|pos err|": np.abs(e[:, 0]), "|e|_2": np.linalg.norm(e, axis=1), "|x|_inf scaled": np.max(np.abs(X) / np.array([5, 5, .3, 2.0]), axis=1),
        "|q|_inf": np.abs(q).max(axis=1), "|z_unc|_inf": np.abs(z).max(axis=1), "|u0_unc|": np.abs(z[:, 0])}
for k, f in prox.items():
    print(f"  proxy {k:16s} corr with iters {np.corrcoef(f, it)[0,1]:+.3f}  with time {np.corrcoef(f, tm)[0,1]:+.3f}  makespan if sorted by it {makespan(tm[np.argsort(-f)]):.0f}")
