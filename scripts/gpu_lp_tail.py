"""Developer script: the LPs of the synthetic (config 5) offline stage that the batched LP kernel returns WITHOUT its certificate
(status 1): saves them (polytope, direction, relaxed row) and compares the kernel's value with HiGHS at default and at tight tolerances."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from scipy.optimize import linprog
from LinearMPCOverNetworks import _native, polytope_lite, workloads
found = []
orig = _native.lp_batch
def spy(A, b, C, relax=None, relax_by=1.0, device=0, want_x=False):
    out = orig(A, b, C, relax=relax, relax_by=relax_by, device=device, want_x=want_x)
    bad = np.flatnonzero(out["status"] == 1)
    for k in bad:
        found.append(dict(A=np.array(A), b=np.array(b), c=np.array(C[k]), rel=-1 if relax is None else int(relax[k]), relax_by=relax_by,
                          val=float(out["val"][k]), iters=int(out["iters"][k])))
    spy.total += len(out["status"])
    return out
spy.total = 0
_native.lp_batch = spy
mpc, w = workloads.make_controller("synthetic", 30, True, device=0)
print("LPs solved on the device:", spy.total, " without certificate:", len(found))
for i, f in enumerate(found):
    bk = f["b"].copy()
    if f["rel"] >= 0: bk[f["rel"]] += f["relax_by"]
    r0 = linprog(-f["c"], A_ub=f["A"], b_ub=bk, bounds=(None, None), method="highs")
    r1 = linprog(-f["c"], A_ub=f["A"], b_ub=bk, bounds=(None, None), method="highs", options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
    if r0.status != 0 or r1.status != 0:
        print("%2d rows %4d dim %2d rel %4d iters %3d | HiGHS status default %d, at 1e-10 %d (kernel value %.12g, HiGHS default %s)" % (
            i, f["A"].shape[0], f["A"].shape[1], f["rel"], f["iters"], r0.status, r1.status, f["val"], None if r0.fun is None else -r0.fun))
        continue
    sc = max(abs(r1.fun), 1.0)
    print("%2d rows %4d dim %2d rel %4d iters %3d | kernel - HiGHS(default) %.2e  kernel - HiGHS(1e-10) %.2e  HiGHS default - tight %.2e (relative to %.3g)" % (
        i, f["A"].shape[0], f["A"].shape[1], f["rel"], f["iters"], (f["val"] + r0.fun) / sc, (f["val"] + r1.fun) / sc, (-r0.fun + r1.fun) / sc, sc))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/lp_tail_cases.npz", **{f"{i}/{k}": np.asarray(v) for i, f in enumerate(found) for k, v in f.items()})
