"""Batched LP kernel of the offline stage (include/tmpc.h: tmpc_lp_batch, csrc/tmpc_lp.hip).

Oracle: scipy.optimize.linprog(method="highs"), the call the reference makes for every support
function (reference utils_polytope.py:19).  Tolerance: the kernel hands its interior-point iterate over
to exact active-set steps, HiGHS answers to its own 1e-7 feasibility tolerance, so values agree to 1e-8
relative (observed: 1e-10 and better), and sets built from them agree row for row.
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
from LinearMPCOverNetworks import polytope_lite as pl
from LinearMPCOverNetworks import utils_polytope as up
from LinearMPCOverNetworks.polytope_lite import Polytope

RTOL = 1e-8


def _sets(name):
    return dict(np.load(os.path.join(common.GOLDEN, name)))


def _reduce_one_by_one(P, abs_tol=pl.ABS_TOL):
    """The reference's order of work: one LP per row, each against the rows still kept."""
    P = Polytope(P.A, P.b, normalize=True)
    A, b = P.A, P.b
    keep = np.ones(len(b), dtype=bool)
    for i in range(len(b)):
        keep[i] = False
        if not keep.any():
            keep[i] = True
            continue
        val, st = pl._lp_max(A[i], np.r_[A[keep], A[i:i + 1]], np.r_[b[keep], b[i] + 1.0])
        if st != 0 or val > b[i] + abs_tol:
            keep[i] = True
    return Polytope(A[keep], b[keep])


# ------------------------------------------------------------------ CPU: host logic around the kernel
def test_batched_reduce_equals_row_by_row_reduce():
    """reduce() tests all rows in one batch and walks only the ties; same rows kept as the sequential walk
    (scipy back-end on both sides) -- including exact duplicates and rows touching a vertex."""
    rng = np.random.default_rng(3)
    for trial in range(6):
        d = 2 + trial % 3
        A = rng.standard_normal((40, d))
        b = 1.0 + rng.random(40)
        A = np.r_[A, A[:5], np.eye(d), -np.eye(d)]                 # duplicates, a box
        b = np.r_[b, b[:5], np.full(2 * d, 0.8)]
        P = Polytope(A, b)
        got, want = pl.reduce(P), _reduce_one_by_one(P)
        assert got.A.shape == want.A.shape
        assert np.allclose(got.A, want.A) and np.allclose(got.b, want.b)
    s = _sets("double_integrator_darup_sets.npz")
    P = Polytope(np.r_[s["Z_A"], s["Z_A"][::3]], np.r_[s["Z_b"], s["Z_b"][::3] + 1e-3])
    got, want = pl.reduce(P), _reduce_one_by_one(P)
    assert got.A.shape == want.A.shape and np.allclose(got.b, want.b)


def test_lp_batch_rejects_bad_arguments(hip_lib):
    """Argument checks happen before any device call (no GPU needed)."""
    L = hip_lib.lib()
    H = np.eye(33)
    h = np.ones(33)
    c = np.ones((1, 33))
    val, st, it = np.zeros(1), np.zeros(1, np.int32), np.zeros(1, np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert L.tmpc_lp_batch(0, 33, 33, ptr(H), ptr(h), 1, ptr(c), None, 1.0, ptr(val), None, ptr(st), ptr(it)) == -2
    assert b"d <= 32" in L.tmpc_last_error(None)
    rel = np.array([5], np.int32)
    assert L.tmpc_lp_batch(0, 2, 4, ptr(H), ptr(h), 1, ptr(c), ptr(rel), 1.0, ptr(val), None, ptr(st), ptr(it)) == -1
    assert L.tmpc_lp_batch(0, 2, 4, None, ptr(h), 1, ptr(c), None, 1.0, ptr(val), None, ptr(st), ptr(it)) == -1
    assert L.tmpc_lp_batch(0, 2, 4, ptr(H), ptr(h), 0, None, None, 1.0, None, None, None, None) == 0
    # rows without a normal say 0 <= h_r: decided on the host -- the set is empty (h_r < 0) or, with no normal at all, the
    # whole space
    Hz = np.zeros((2, 2))
    c2 = np.ones((1, 2))
    assert L.tmpc_lp_batch(0, 2, 2, ptr(Hz), ptr(np.array([1.0, -1.0])), 1, ptr(c2), None, 1.0, ptr(val), None, ptr(st), ptr(it)) == 0
    assert st[0] == 2 and np.isnan(val[0])
    assert L.tmpc_lp_batch(0, 2, 2, ptr(Hz), ptr(np.array([1.0, 1.0])), 1, ptr(c2), None, 1.0, ptr(val), None, ptr(st), ptr(it)) == 0
    assert st[0] == 4 and np.isinf(val[0])
    with pytest.raises(ValueError):
        old = pl.set_lp_backend("hip")
        try:
            pl.lp_max_batch(np.ones((1, 33)), np.eye(33), np.ones(33))
        finally:
            pl.set_lp_backend(old)


# ------------------------------------------------------------------ GPU: kernel against HiGHS
def _against_highs(A, b, Cm, rel, hip_lib):
    raw = hip_lib.lp_batch(A, b, Cm, relax=rel, relax_by=1.0, want_x=True)
    val, xs = raw["val"], raw["x"]
    st = (raw["status"] > 1).astype(int)
    # status 1 = the active-set steps did not finish on a highly degenerate vertex and the interior-point iterate was
    # returned: allowed for at most 1 % of a batch, and then accurate to 1e-6 (HiGHS' own feasibility tolerance is 1e-7)
    inexact = raw["status"] == 1
    assert inexact.mean() <= 0.01
    ref = np.empty(len(Cm))
    for i, (c, r) in enumerate(zip(Cm, rel)):
        b2 = b.copy()
        if r >= 0:
            b2[r] += 1.0
        ref[i], s1 = pl._lp_max(c, A, b2)
        assert s1 == 0
    assert np.all(st == 0)
    err = np.abs(val - ref) / np.maximum(np.abs(ref), 1.0)
    assert err[~inexact].max() <= RTOL, err[~inexact].max()
    assert not inexact.any() or err[inexact].max() <= 1e-6
    for x, r, v, c in zip(xs, rel, val, Cm):                        # the maximiser is feasible and attains the value
        b2 = b + (np.arange(len(b)) == r) * 1.0
        assert np.max(A @ x - b2) <= 1e-9 * max(1.0, np.abs(b2).max())
        assert abs(c @ x - v) <= 1e-9 * max(1.0, abs(v))


@pytest.fixture()
def hip_lp():
    old = pl.set_lp_backend("hip")
    yield
    pl.set_lp_backend(old)


@pytest.mark.gpu
@pytest.mark.parametrize("name,key", [("cartpole_sets.npz", "Z"), ("cartpole_sets.npz", "Xf"), ("cartpole_sets.npz", "ZmW"),
                                      ("double_integrator_darup_sets.npz", "Z"), ("double_integrator_rakovic_sets.npz", "Xf"),
                                      ("synthetic_sets.npz", "Z"), ("synthetic_sets.npz", "Xf")])
def test_support_values_match_highs(hip_lib, hip_lp, name, key):
    """Random directions, directions along rows (degenerate: whole facets optimal) and the redundancy tests of
    polytope.reduce (row relaxed by one), d = 2 ... 28, 26 ... 854 rows."""
    s = _sets(name)
    A, b = s[key + "_A"], s[key + "_b"]
    rng = np.random.default_rng(0)
    n = 60
    ridx = rng.integers(0, len(b), n)
    Cm = np.r_[rng.standard_normal((n, A.shape[1])), A[rng.integers(0, len(b), n)], A[ridx]]
    rel = np.r_[np.full(2 * n, -1), ridx].astype(np.int32)
    _against_highs(A, b, Cm, rel, hip_lib)


@pytest.mark.gpu
def test_lp_edge_cases(hip_lib, hip_lp):
    box = np.r_[np.eye(3), -np.eye(3)]
    # unbounded: the cone x <= 1 only
    val, st = pl.lp_max_batch(np.array([[-1.0, 0, 0], [1.0, 0, 0]]), np.eye(3), np.ones(3))
    assert st[0] == 3 and np.isinf(val[0]) and st[1] == 0 and abs(val[1] - 1.0) < 1e-12
    # infeasible: x <= -1 and -x <= -1
    val, st = pl.lp_max_batch(np.ones((1, 3)), np.r_[box, -np.eye(3)[:1]], np.r_[np.ones(6), -2.0])
    assert st[0] != 0 and not np.isfinite(val[0])
    # zero objective, single objective, tiny and huge scales, offset box (origin outside)
    val, st = pl.lp_max_batch(np.zeros((1, 3)), box, np.ones(6))
    assert st[0] == 0 and val[0] == 0.0
    for scale in (1e-6, 1.0, 1e6):
        val, st = pl.lp_max_batch(np.array([[1.0, 2.0, -3.0]]), box, scale * np.ones(6))
        assert st[0] == 0 and abs(val[0] - 6.0 * scale) <= 1e-10 * scale
    val, st = pl.lp_max_batch(np.array([[1.0, 1.0, 1.0]]), box, np.r_[5.0, 5.0, 5.0, -4.0, -4.0, -4.0])
    assert st[0] == 0 and abs(val[0] - 15.0) < 1e-10
    # a row whose normal is round-off (1e-18) constrains nothing and must not be scaled into a constraint
    val, st = pl.lp_max_batch(np.array([[1.0, 2.0, -3.0]]), np.r_[box, [[1e-18, -2e-18, 0.0]]], np.r_[np.ones(6), 0.5])
    assert st[0] == 0 and abs(val[0] - 6.0) <= 1e-10
    # many objectives, more than one pass of the persistent grid
    rng = np.random.default_rng(1)
    Cm = rng.standard_normal((5000, 3))
    val, st = pl.lp_max_batch(Cm, box, np.array([1.0, 2.0, 3.0, 1.0, 2.0, 3.0]))
    assert np.all(st == 0) and np.allclose(val, np.abs(Cm) @ np.array([1.0, 2.0, 3.0]), rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name,N,method", [("double_integrator", 5, 0), ("double_integrator", 10, 1), ("cartpole", 10, 1),
                                           ("synthetic", 30, 1)])
def test_offline_sets_with_the_lp_kernel_equal_the_committed_sets(hip_lib, hip_lp, name, N, method):
    """The whole offline stage (mRPI / tightening / Gilbert-Tan terminal set / Z (-) W) driven by the LP kernel
    reproduces the committed sets, which were computed with HiGHS: same rows, offsets to 1e-8 (config 5: same sets)."""
    from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC
    w = common.workload(name)
    mpc = TubeTrackingMPC(w["A"], w["B"], w["Q"], w["R"], N)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    mpc.determine_mRPI(w["W"], rpi_method=method)
    mpc.tighten_constraints()
    mpc.determine_Xf()
    ZmW = up.pont_diff(mpc._Z, w["W"])
    fix = {("double_integrator", 0): "double_integrator_rakovic_sets.npz", ("double_integrator", 1): "double_integrator_darup_sets.npz",
           ("cartpole", 1): "cartpole_sets.npz", ("synthetic", 1): "synthetic_sets.npz"}[(name, method)]
    s = _sets(fix)
    for key, P in (("Z", mpc._Z), ("Xc", mpc._Xc), ("Uc", mpc._Uc), ("Xf", mpc._Xf), ("ZmW", ZmW)):
        A, b = s[key + "_A"], s[key + "_b"]
        if name != "synthetic":
            assert P.A.shape == A.shape, (key, P.A.shape, A.shape)
            assert np.allclose(P.A, A, rtol=0, atol=1e-9), key
            assert np.allclose(P.b, b, rtol=0, atol=1e-8), key
            continue
        # config 5 (28-dimensional terminal set): a couple of rows sit within HiGHS' 1e-7 tolerance of being redundant
        # and may be decided either way; the SETS agree -- each contains the other to 1e-6 (checked with HiGHS)
        assert abs(P.A.shape[0] - A.shape[0]) <= max(2, A.shape[0] // 100), (key, P.A.shape, A.shape)
        old = pl.set_lp_backend("scipy")
        try:
            v1, s1 = pl.lp_max_batch(A, P.A, P.b)
            v2, s2 = pl.lp_max_batch(P.A, A, b)
        finally:
            pl.set_lp_backend(old)
        assert np.all(s1 == 0) and np.all(s2 == 0)
        assert np.max(v1 - b) <= 1e-6 and np.max(v2 - P.b) <= 1e-6, key


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["double_integrator_darup_sets.npz", "cartpole_sets.npz"])
def test_projection_with_the_lp_kernel(hip_lib, name):
    """eliminate_terminal_auxiliaries (convex-hull projection, one LP batch per refinement round) gives the same
    polytope with the LP kernel as with HiGHS."""
    s = _sets(name)
    w = common.workload("cartpole" if "cartpole" in name else "double_integrator")
    Xf = Polytope(s["Xf_A"], s["Xf_b"])
    got = {}
    for be in ("scipy", "hip"):
        old = pl.set_lp_backend(be)
        try:
            got[be] = up.eliminate_terminal_auxiliaries(Xf, w["A"], w["B"])
        finally:
            pl.set_lp_backend(old)
    a, b = got["hip"], got["scipy"]
    assert a.A.shape == b.A.shape
    # facets may come out in a different order: compare as sets of rows
    ka = np.lexsort(np.round(np.c_[a.A, a.b], 7).T)
    kb = np.lexsort(np.round(np.c_[b.A, b.b], 7).T)
    assert np.allclose(a.A[ka], b.A[kb], atol=1e-7) and np.allclose(a.b[ka], b.b[kb], atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("d,nr", [(1, 2), (1, 7), (2, 3), (3, 40), (5, 64), (5, 65), (9, 300), (13, 200), (16, 128), (17, 90), (24, 400), (32, 150),
                                  (4, 4000)])
def test_random_polytopes_all_dimensions(hip_lib, hip_lp, d, nr):
    """Random bounded polytopes (a box plus random cuts, shifted so that the origin is OUTSIDE, with duplicated rows)
    in every padded dimension class of the kernel (4, 8, 12, 16, 32) and at the row-count edges (one lane pass,
    one row over, thousands of rows)."""
    rng = np.random.default_rng(100 * d + nr)
    nbox = min(2 * d, nr)
    A = np.r_[np.eye(d), -np.eye(d)][:nbox]
    b = np.full(nbox, 2.0)
    if nr > nbox:
        G = rng.standard_normal((nr - nbox, d))
        A = np.r_[A, G]
        b = np.r_[b, 0.5 + rng.random(nr - nbox) * np.linalg.norm(G, axis=1)]
    if nr >= 2 * d + 4:
        A[-2:] = A[2 * d:2 * d + 2]                    # exact duplicates
        b[-2:] = b[2 * d:2 * d + 2]
    shift = 3.0 * rng.standard_normal(d)               # origin outside the set
    b = b + A @ shift
    if nr < 2 * d:                                     # not enough rows for a bounded set: only directions it bounds
        Cm = A[rng.integers(0, nr, 20)]
    else:
        Cm = np.r_[rng.standard_normal((40, d)), A[rng.integers(0, nr, 20)]]
    rel = np.full(len(Cm), -1, dtype=np.int32)
    _against_highs(A, b, Cm, rel, hip_lib)


@pytest.mark.gpu
def test_darup_known_answer_with_the_lp_kernel(hip_lib, hip_lp):
    """The reference's printed known answer k* = 5, 6, 10 (Examples of Set Operations/Example of Approximation of
    mRPI_Darup.py:50-55) with the container-set LPs solved by the kernel, and the Gilbert-Tan example set
    (Example of Output Admissible Set Calculation.py scenario: double integrator under LQR, |x| <= 5, |u| <= 1)
    equal to the HiGHS result."""
    from LinearMPCOverNetworks.control_lite import dlqr
    from LinearMPCOverNetworks.polytope_lite import box2poly
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([[0.5], [1.0]])
    W = box2poly([[-0.1, 0.1]] * 2)
    X = Polytope(np.r_[np.eye(2), -np.eye(2)], [4, 2, 8, 4])
    U = box2poly([[-1.0, 1.0]])
    K, _, _ = dlqr(A, B, np.eye(2), np.eye(1))
    for eps, k_expected in ((1e-1, 5), (1e-2, 6), (1e-3, 10)):
        rpi, status = up.calculate_RPI(A - B @ K, W, X, U, K, eps, 50, verbose=False)
        assert status == 0 and rpi.k_star == k_expected
    Acl = A - B @ K
    Xu = Polytope(np.r_[np.eye(2), -np.eye(2), -K, K], np.r_[5.0 * np.ones(4), 1.0, 1.0])
    got = up.calculate_maximum_admissible_output_set(Acl, Xu, verbose=False)
    old = pl.set_lp_backend("scipy")
    try:
        want = up.calculate_maximum_admissible_output_set(Acl, Xu, verbose=False)
    finally:
        pl.set_lp_backend(old)
    assert got.A.shape == want.A.shape and np.allclose(got.A, want.A, atol=1e-9) and np.allclose(got.b, want.b, atol=1e-8)


@pytest.mark.gpu
def test_degenerate_faces_finish_with_certificate(hip_lib):
    """tests/golden/lp_degenerate_cases.npz (the six LPs per cartpole model that the round-2 kernel returned at its
    iteration cap, see tests/test_wavesim.py): status 0 through the C ABI, values against HiGHS run with 1e-10 tolerances."""
    from scipy.optimize import linprog
    Z = np.load(os.path.join(common.GOLDEN, "lp_degenerate_cases.npz"))
    for name in ("a", "b"):
        H, h, Cm, rel = (Z[f"{name}_{k}"] for k in ("H", "h", "C", "rel"))
        raw = hip_lib.lp_batch(H, h, Cm, relax=rel, relax_by=1.0, want_x=True)
        assert np.all(raw["status"] == 0) and raw["iters"].max() <= 40
        for c, r, v, x in zip(Cm, rel, raw["val"], raw["x"]):
            hk = h + (np.arange(len(h)) == r) * 1.0
            res = linprog(-c, A_ub=H, b_ub=hk, bounds=(None, None), method="highs",
                          options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
            assert res.status == 0 and abs(v + res.fun) <= 1e-9 * max(1.0, abs(res.fun))
            assert np.max((H @ x - hk) / np.maximum(np.abs(hk), 1.0)) <= 1e-10
