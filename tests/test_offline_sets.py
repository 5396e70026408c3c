"""Host-side set-up stage (scipy): known answers and invariants.  CPU only."""
import os

import numpy as np
import pytest

import common
from LinearMPCOverNetworks import polytope_lite as pl
from LinearMPCOverNetworks import utils_polytope as up
from LinearMPCOverNetworks import workloads
from LinearMPCOverNetworks.control_lite import c2d, dlqr, dlyap
from LinearMPCOverNetworks.polytope_lite import Polytope, box2poly, is_subset, reduce


def test_darup_kstar_known_answer():
    """The one printed known-answer of the reference: k* = 5, 6, 10 for eps = 1e-1, 1e-2, 1e-3
    (Examples of Set Operations/Example of Approximation of mRPI_Darup.py:50-55)."""
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([[0.5], [1.0]])
    W = box2poly([[-0.1, 0.1]] * 2)
    X = Polytope(np.r_[np.eye(2), -np.eye(2)], [4, 2, 8, 4])
    U = box2poly([[-1.0, 1.0]])
    K, _, _ = dlqr(A, B, np.eye(2), np.eye(1))
    for eps, k_expected in ((1e-1, 5), (1e-2, 6), (1e-3, 10)):
        rpi, status = up.calculate_RPI(A - B @ K, W, X, U, K, eps, 50, verbose=False)
        assert status == 0 and rpi.k_star == k_expected


def test_cartpole_kstar_and_retry():
    """Cartpole (results_linear_system.py:26-110): not found within s_max=200, k*=308 at 2000
    (the s_max x10 retry of TubeRegulatorMPC.py:68-71)."""
    w = workloads.cartpole()
    K, _, _ = dlqr(w["A"], w["B"], w["Q"], w["R"])
    Acl = w["A"] - w["B"] @ K
    assert abs(np.max(np.abs(np.linalg.eigvals(Acl))) - 0.95197) < 1e-4
    r, st = up.calculate_RPI(Acl, w["W"], w["X"], w["U"], K, 1e-4, 200, verbose=False)
    assert r is None and st == -1
    r, st = up.calculate_RPI(Acl, w["W"], w["X"], w["U"], K, 1e-4, 2000, verbose=False)
    assert st == 0 and r.k_star == 308 and r.A.shape == (3080, 4)


def test_support_closed_form_matches_lp():
    rng = np.random.default_rng(0)
    box = box2poly([[-1.0, 2.0], [-0.5, 0.25], [-3.0, 3.0]])
    gen = Polytope(box.A @ np.linalg.qr(rng.normal(size=(3, 3)))[0], box.b)   # rotated: LP path
    for _ in range(10):
        a = rng.normal(size=3)
        lp = pl._lp_max(a, box.A, box.b)[0]
        assert abs(up.support(box, a) - lp) < 1e-9
        assert np.isfinite(up.support(gen, a))


def test_pontryagin_difference_and_linear_image():
    Z = box2poly([[-0.2, 0.3], [-0.1, 0.1]])
    X = box2poly([[-1.0, 1.0], [-2.0, 2.0]])
    D = up.pont_diff(X, Z)
    np.testing.assert_allclose(D.b, [0.7, 1.9, 0.8, 1.9])
    K = np.array([[2.0, -1.0]])
    img = up.scale(Z, -K)               # 1-D image -> interval
    lo, hi = -img.b[1], img.b[0]
    V = np.array([[-0.2, -0.1], [-0.2, 0.1], [0.3, -0.1], [0.3, 0.1]]) @ (-K.T)
    assert abs(lo - V.min()) < 1e-12 and abs(hi - V.max()) < 1e-12


def test_reduce_and_subset():
    P = Polytope(np.r_[np.eye(2), -np.eye(2), [[1.0, 1.0]], [[1.0, 0.0]]], [1, 1, 1, 1, 5.0, 2.0])
    R = reduce(P)
    assert R.A.shape[0] == 4
    assert is_subset(R, P) and is_subset(P, R)
    assert np.array([0.5, -0.5]) in R and not (np.array([1.5, 0.0]) in R)


def test_gilbert_tan_invariance_double_integrator():
    """O_inf is positively invariant and inside X (utils_polytope.py:247-268)."""
    w = workloads.double_integrator()
    K, _, _ = dlqr(w["A"], w["B"], w["Q"], w["R"])
    Acl = w["A"] - w["B"] @ K
    XU = Polytope(np.r_[w["X"].A, -w["U"].A @ K], np.r_[w["X"].b, w["U"].b])
    O = up.calculate_maximum_admissible_output_set(Acl, XU, verbose=False)
    V = up.extreme(O)
    for v in V:
        assert (Acl @ v) in O
        assert v in XU


def test_cached_double_integrator_sets_reproduce():
    """tests/golden/double_integrator_darup_sets.npz is what the set-up stage computes."""
    w = workloads.double_integrator()
    from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC
    mpc = TubeTrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    mpc.determine_mRPI(w["W"], rpi_method=1)
    mpc.tighten_constraints()
    mpc.determine_Xf(verbose=False)
    gold = np.load(os.path.join(common.GOLDEN, "double_integrator_darup_sets.npz"))
    np.testing.assert_allclose(np.abs(mpc._K), [[0.4221, 1.2439]], atol=1e-4)
    np.testing.assert_allclose(mpc._Xc.b, gold["Xc_b"], atol=1e-10)
    np.testing.assert_allclose(mpc._Uc.b, gold["Uc_b"], atol=1e-10)
    assert mpc._Z.A.shape == gold["Z_A"].shape and mpc._Xf.A.shape == gold["Xf_A"].shape
    assert is_subset(mpc._Xf, Polytope(gold["Xf_A"], gold["Xf_b"]), 1e-6)


def test_rakovic_mrpi_is_invariant():
    """eps-mRPI of Rakovic (utils_polytope.py:180-245): A Z (+) W inside Z."""
    w = workloads.double_integrator()
    K, _, _ = dlqr(w["A"], w["B"], w["Q"], w["R"])
    Acl = w["A"] - w["B"] @ K
    Z, st = up.calculate_minimal_robust_positively_invariant_set(Acl, w["W"], 1e-4, 200)
    assert st == 0
    VZ, VW = up.extreme(Z), up.extreme(w["W"])
    for vz in VZ:
        for vw in VW:
            assert Z.contains(Acl @ vz + vw, 1e-7)


def test_control_helpers():
    w = workloads.cartpole()
    A, B = w["A"], w["B"]
    np.testing.assert_allclose(A[0], [1, 0.02, -1.8787e-4, -1.2521e-6], rtol=1e-3)
    K, S, _ = dlqr(A, B, w["Q"], w["R"])
    np.testing.assert_allclose(K, [[-23.4054, -22.3848, -104.2138, -24.3312]], rtol=1e-5)
    Acl = A - B @ K
    Ql = w["Q"] + K.T @ w["R"] @ K
    P = dlyap(Acl, Ql)
    np.testing.assert_allclose(Acl @ P @ Acl.T - P + Ql, 0, atol=1e-5 * np.abs(P).max())
    Ad, Bd = c2d(np.zeros((1, 1)), np.ones((1, 1)), 0.5)
    np.testing.assert_allclose([Ad[0, 0], Bd[0, 0]], [1.0, 0.5])


def test_projection_by_convex_hull_method():
    """project_polytope against the projected vertices of a rotated cube (exact V-representation)."""
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.standard_normal((4, 4)))
    cube = Polytope(np.r_[np.eye(4), -np.eye(4)] @ Q.T, np.ones(8))
    for k in (1, 2, 3):
        E = rng.standard_normal((k, 4))
        Pk = up.project_polytope(cube, E)
        V = up.extreme(cube) @ E.T
        assert np.max(Pk.A @ V.T - Pk.b[:, None]) < 1e-9              # contains every projected vertex
        Hk = up.determine_convex_hull(V)
        assert Pk.A.shape[0] == Hk.A.shape[0]
        assert is_subset(Pk, Hk, 1e-8) and is_subset(Hk, Pk, 1e-8)


def test_terminal_auxiliaries_elimination_is_exact():
    """HTP [x_bar; u_bar] <= hTP  <=>  some (x_aux, u_aux) satisfies the literal rows of TubeTrackingMPC.py:293,
    for steady states (x_bar, u_bar) on either side of the boundary."""
    from scipy.optimize import linprog
    import os
    sets = dict(np.load(os.path.join(common.GOLDEN, "cartpole_sets.npz")))
    w = workloads.cartpole()
    Xf = Polytope(sets["Xf_A"], sets["Xf_b"])
    PT = up.eliminate_terminal_auxiliaries(Xf, w["A"], w["B"])
    assert PT.A.shape == (2, 5)                                        # nu = 1: an interval of steady states
    Nss = up.steady_state_basis(w["A"], w["B"])
    assert np.max(np.abs(np.c_[w["A"] - np.eye(4), w["B"]] @ Nss)) < 1e-12
    # boundary values of phi along the basis
    hi = min(b / (a @ Nss[:, 0]) for a, b in zip(PT.A, PT.b) if a @ Nss[:, 0] > 0)
    lo = max(b / (a @ Nss[:, 0]) for a, b in zip(PT.A, PT.b) if a @ Nss[:, 0] < 0)
    HT, hT = Xf.A, Xf.b
    for phi, inside in ((0.0, True), (hi * (1 - 1e-6), True), (lo * (1 - 1e-6), True), (hi * (1 + 1e-4), False), (lo * (1 + 1e-4), False)):
        xu = Nss[:, 0] * phi
        res = linprog(np.zeros(5), A_ub=np.c_[HT[:, :4], HT[:, 8:]], b_ub=hT - HT[:, 4:8] @ xu[:4], bounds=(None, None), method="highs")
        assert (res.status == 0) == inside
        assert bool(np.all(PT.A @ xu <= PT.b + 1e-12)) == inside


def test_rpi_that_cannot_fit_is_refused():
    """Disturbance too large for the constraints: condition (9b) of the Darup-Teichrib construction fails for every k.
    The reference retries with s_max x 10 forever (TubeRegulatorMPC.py:48-71); here set-up stops with an error."""
    from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([[0.5], [1.0]])
    mpc = TubeTrackingMPC(A, B, np.eye(2), np.eye(1), 5)
    mpc.set_input_constraints(box2poly([[-0.1, 0.1]]))
    mpc.set_state_constraints(box2poly([[-1.0, 1.0]] * 2))
    with pytest.raises(ValueError):
        mpc.determine_mRPI(box2poly([[-0.5, 0.5]] * 2), rpi_method=1)
