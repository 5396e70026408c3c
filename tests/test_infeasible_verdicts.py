"""An infeasible QP must come back as TMPC_STATUS_INFEASIBLE (2), never as an iterate under MAX_ITER (1): the closed loop and the
R-MPC "dead run" accounting of results_linear_system.py:262-287 only stop on status >= 2, so a status-1 answer to an empty
feasible set would feed a constraint-violating input into the plant (cvxpy reports `infeasible` there and the reference
returns None, TubeTrackingMPC.py:185-194).  ABI 5 declares INFEASIBLE only with a Farkas-type certificate of the interior-point
phase (or by the rows that depend on x_k alone); this file checks that the certificate fires on states whose infeasibility
only shows LATER in the horizon -- the stage-0 rows hold -- for both kernels and inside the warm-started closed loop.
Ground truth: HiGHS on the un-condensed constraint set (oracle/qp_sparse.lp_infeasible, the LP solver the reference calls)."""
import numpy as np
import pytest

import common
from LinearMPCOverNetworks import montecarlo
from oracle import qp_sparse

pytestmark = pytest.mark.gpu


def _late_infeasible_states(mpc, n, seed):
    """States INSIDE the tightened stage set (so the x_k-only rows pass) near its boundary, moving outwards fast: the horizon
    cannot brake them.  Returns X, R and the HiGHS verdicts (True = empty feasible set; with a 1e-6 margin either way)."""
    rng = np.random.default_rng(seed)
    hx = mpc._Xc.b[:4]
    X = rng.uniform(-1, 1, (n, 4)) * hx * np.array([1.0, 1.0, 0.5, 0.5]) * 0.9
    side = rng.choice([-1.0, 1.0], n)
    X[:, 0] = side * hx[0] * rng.uniform(0.7, 0.98, n)            # close to the position bound ...
    X[:, 1] = side * hx[1] * rng.uniform(0.0, 0.5, n)             # ... and heading for it (about half of them cannot brake in time)
    R = np.zeros((n, 4))
    R[:, 0] = rng.uniform(-1, 1, n)
    tpl = qp_sparse.SparseTemplate(mpc._problem_dict(), 0)
    empty, nonempty = np.zeros(n, bool), np.zeros(n, bool)
    for k in range(n):
        qp = tpl.instance(X[k], R[k])
        tight, loose = dict(qp), dict(qp)
        tight["h"] = qp["h"] - 1e-6 * np.maximum(1.0, np.abs(qp["h"]))
        loose["h"] = qp["h"] + 1e-6 * np.maximum(1.0, np.abs(qp["h"]))
        empty[k] = qp_sparse.lp_infeasible(loose)                # empty even with the rows relaxed
        nonempty[k] = not qp_sparse.lp_infeasible(tight)          # non-empty even with the rows tightened
    return X, R, empty, nonempty


@pytest.mark.parametrize("N,path", [(10, "wave"), (10, "block"), (20, "wave")])
def test_late_infeasibility_is_status_2_never_1(hip_lib, N, path):
    mpc, w = common.make_mpc("cartpole", N, True, create=True)
    mpc.set_kernel_path(path)
    X, R, empty, nonempty = _late_infeasible_states(mpc, 160, seed=5 + N)
    assert empty.sum() >= 30 and nonempty.sum() >= 30, (empty.sum(), nonempty.sum())
    out = mpc._solve(X, R, want_traj=False)
    st = out["status"]
    print(f"N = {N}, {path} kernel: {int(empty.sum())} empty / {int(nonempty.sum())} non-empty by HiGHS; device statuses {np.bincount(st, minlength=4)}")
    assert np.all((st == 0) | (st == 2)), np.bincount(st)          # no MAX_ITER, no NUMERICAL
    assert np.all(st[empty] == 2)
    assert np.all(st[nonempty] == 0)
    assert np.all(np.isnan(out["u_nom"][st == 2])) and np.all(np.isfinite(out["u_nom"][st == 0]))


def test_infeasible_solves_inside_the_warm_started_closed_loop(hip_lib):
    """Extended controller (the estimate follows the measured state, results_linear_system_with_extendedMPC.py:276-279): a
    disturbance far outside W at step 0 puts the estimates of the following solves outside the feasible set.  Cold and
    warm-started device loops count the same failed solves as the host loop around the same solver, whose statuses are only
    ever 0 or 2."""
    nb, T, N = 32, 6, 10
    mpc, w = common.make_mpc("cartpole", N, True, extended=True, create=True)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=3)
    p_loss = np.zeros(nb)
    dist[: nb // 2, 0, 1] += 3.0 * np.sign(dist[: nb // 2, 0, 1] + 1e-300)        # a kick in the cart velocity for half of the batch
    ref = 0.5 * np.ones(T)
    seen = []

    def packets(x_hat, r, gamma):
        U, x0, st = mpc.determine_packets(x_hat, r, gamma)
        seen.append(st.copy())
        return U, x0, st
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    host = montecarlo.run_remote_tube_mpc(packets, w["A"], w["B"], K, Kp, N, mpc._Z, p_loss, ref, th, ga, dist, extended=True)
    seen = np.array(seen)
    assert np.all((seen == 0) | (seen == 2)), np.bincount(seen.ravel())
    assert (seen == 2).sum() >= nb // 4                                               # the kick really produced infeasible solves
    cold = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=True)
    warm = mpc.run_closed_loop(p_loss, ref, th, ga, dist, extended=True, warm_start=True)
    assert np.array_equal(cold["not_optimal"], (seen != 0).sum(axis=0))
    assert np.array_equal(warm["not_optimal"], cold["not_optimal"])
    np.testing.assert_allclose(warm["x_final"], cold["x_final"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(cold["x_final"], host["x_final"], atol=1e-8, rtol=0)
