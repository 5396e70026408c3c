"""Regression anchors of SURVEY.md Appendix D, asserted on the very objects that feed BOTH the
oracle and the kernels (tests/common.make_mpc + the committed sets under tests/golden/).

K, P, T and the four sets reach the oracle and the library through the same `_problem_dict()`,
so a wrong Lyapunov convention (SURVEY F-2) or a wrong tightening would pass every parity test.
The numbers below were obtained in the survey by an independent scipy restatement of
TubeRegulatorMPC.py:16-24, TubeTrackingMPC.py:35-102 and utils_polytope.py:270-414 (they are not
reference outputs -- the reference cannot run here -- but they do not come from this package).
CPU only."""
import numpy as np
import scipy.linalg

import common
from oracle import qp_sparse
from oracle.oracle import Oracle


def test_cartpole_gains_weights_and_sets():
    mpc, w = common.make_mpc("cartpole", 10, True)
    A, B = w["A"], w["B"]
    # model (results_linear_system.py:26-61)
    np.testing.assert_allclose(A[0], [1, 0.02, -1.8787e-4, -1.2521e-6], rtol=2e-4)
    np.testing.assert_allclose(B[:, 0], [1.9925e-4, 1.99258e-2, -3.8341e-4, -3.83669e-2], rtol=2e-4)
    np.testing.assert_allclose(sorted(np.abs(np.linalg.eigvals(A))), [0.9131, 1.0, 1.0, 1.0951], atol=2e-4)
    # K: dlqr, control law u = -K x (TubeRegulatorMPC.py:19)
    np.testing.assert_allclose(mpc._K, [[-23.4054, -22.3848, -104.2138, -24.3312]], rtol=1e-5)
    Acl = A - B @ mpc._K
    assert abs(np.max(np.abs(np.linalg.eigvals(Acl))) - 0.95197) < 1e-5
    # P: python-control's dlyap(Acl, Q_lyap) solves  Acl X Acl' - X + Q_lyap = 0  (TubeRegulatorMPC.py:21-23) -- NOT the
    # LQR cost-to-go and NOT the transposed equation; the two differ by orders of magnitude here (SURVEY F-2)
    Ql = w["Q"] + mpc._K.T @ w["R"] @ mpc._K
    Ql = (Ql + Ql.T) / 2
    P = mpc._P
    np.testing.assert_allclose(Acl @ P @ Acl.T - P + Ql, 0, atol=1e-6 * np.abs(P).max())
    np.testing.assert_allclose(np.diag(P), [8.02e4, 3.33e5, 1.04e4, 5.08e5], rtol=5e-3)
    S = scipy.linalg.solve_discrete_are(A, B, w["Q"], w["R"])
    np.testing.assert_allclose(np.diag(S), [4.78e3, 1.38e3, 1.17e4, 5.03e2], rtol=5e-3)
    assert np.linalg.norm(P - S) / np.linalg.norm(S) > 10              # the other convention is far away
    Pt = scipy.linalg.solve_discrete_lyapunov(Acl.T, Ql)                # transposed equation: also far away
    assert np.linalg.norm(P - Pt) / np.linalg.norm(Pt) > 0.5
    np.testing.assert_allclose(mpc._Tout, 10 * P)                       # TubeTrackingMPC.py:27
    assert mpc._lambda == 0.99999                                        # TubeTrackingMPC.py:22
    # sets (Darup eps = 1e-4: k* = 308 -> 3080 raw rows; tests/test_offline_sets.py checks k* itself)
    assert mpc._Z.A.shape == (854, 4)
    assert mpc._Xf.A.shape == (420, 9)
    assert mpc._Xc.A.shape == (8, 4) and mpc._Uc.A.shape == (2, 1)
    hx = np.abs(mpc._Xc.b / np.abs(mpc._Xc.A).sum(1))
    np.testing.assert_allclose(sorted(hx), sorted([4.4419, 3.8296, 0.07586, 1.15947] * 2), rtol=2e-4)
    hu = np.abs(mpc._Uc.b / np.abs(mpc._Uc.A).sum(1))
    np.testing.assert_allclose(hu, [3.9107, 3.9107], rtol=2e-4)
    p = mpc._problem_dict()
    assert np.array_equal(p["K"], mpc._K) and np.array_equal(p["P"], P) and np.array_equal(p["T"], 10 * P)
    assert np.array_equal(p["HT"], mpc._Xf.A) and np.array_equal(p["Hx"], mpc._Xc.A)


def test_double_integrator_gains_and_sets():
    mpc, w = common.make_mpc("double_integrator_darup", 5, False)
    w = common.workload("double_integrator")
    np.testing.assert_allclose(mpc._K, [[0.4221, 1.2439]], atol=5e-5)
    Acl = w["A"] - w["B"] @ mpc._K
    assert abs(np.max(np.abs(np.linalg.eigvals(Acl))) - 0.4221) < 1e-4
    S = scipy.linalg.solve_discrete_are(w["A"], w["B"], w["Q"], w["R"])
    assert abs(np.linalg.norm(mpc._P - S) / np.linalg.norm(S) - 1.43) < 0.01
    assert mpc._Z.A.shape[0] == 42 and mpc._Xf.A.shape == (26, 5)
    hx = np.abs(mpc._Xc.b / np.abs(mpc._Xc.A).sum(1))
    np.testing.assert_allclose(sorted(hx), sorted([7.465, 7.699] * 2), atol=2e-3)
    np.testing.assert_allclose(np.abs(mpc._Uc.b / np.abs(mpc._Uc.A).sum(1)), [0.7426, 0.7426], atol=2e-4)


def test_known_minimiser_double_integrator(oracle_lib):
    """double integrator, N = 5, free x_0, Darup sets, x_k = [1, 2], r = [5, 0]: sparse and condensed
    forms solved with scipy trust-constr and polished on the active set gave u*_0 = -0.737182900857,
    x*_0 = [1.534713867, 1.745580949], x_bar* = [4.93424, 0], cost 20.38383 (one active inequality)."""
    mpc, _ = common.make_mpc("double_integrator_darup", 5, False)
    p = mpc._problem_dict()
    sol = Oracle(p).solve(np.array([[1.0, 2.0]]), np.array([[5.0, 0.0]]))
    assert sol["status"][0] == 0
    assert abs(sol["u_nom"][0, 0, 0] - (-0.737182900857)) < 1e-9
    np.testing.assert_allclose(sol["x_nom0"][0], [1.534713867, 1.745580949], atol=1e-8)
    np.testing.assert_allclose(sol["x_ss"][0], [4.93424, 0.0], atol=1e-5)
    qp = qp_sparse.build_sparse_qp(p, [1.0, 2.0], [5.0, 0.0])
    v = qp_sparse.pack(qp, sol["x_nom"][0], sol["u_nom"][0], sol["x_ss"][0], sol["u_ss"][0])
    assert abs(qp_sparse.objective(qp, v) - 20.38383) < 1e-5
    assert qp_sparse.kkt_certificate(qp, v)["n_active"] == 1
