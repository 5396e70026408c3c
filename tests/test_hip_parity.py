"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and
the committed golden fixtures.  All tests here need the MI355X (-m gpu).

Tolerance: BASELINE.json's north_star asks for u*_0 within 1e-6 of the reference CPU
solver.  Both sides here run to the exact minimiser, so the assertions use 1e-8 on inputs
(1e-6 would also pass) and 1e-9 on the steady state."""
import os

import numpy as np
import pytest

import common
from oracle import qp_sparse
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

ATOL_U = 1e-8
ATOL_SS = 1e-9
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))


@pytest.fixture(scope="module")
def cartpole(hip_lib, oracle_lib):
    mpc, w = common.make_mpc("cartpole", 10, True, create=True)
    return mpc, w, Oracle(mpc._problem_dict())


def test_native_library_is_the_one_in_tree(hip_lib):
    assert os.path.samefile(hip_lib.LIB_PATH, os.path.join(common.PKG, "lib", "libtmpc_hip.so"))
    assert hip_lib.lib().tmpc_abi_version() == 5


def test_golden_fixture_cartpole_N10(cartpole, hip_lib):
    """600 closed-loop (x_k, ref) pairs; expected outputs committed under tests/golden/."""
    mpc, _, _ = cartpole
    gold = np.load(os.path.join(common.GOLDEN, "cartpole_N10_oracle.npz"))
    out = mpc._solve(S[:, :4], S[:, 4:])
    assert np.array_equal(out["status"], gold["status"]) and np.all(out["status"] == 0)
    np.testing.assert_allclose(out["u_nom"], gold["u_nom"], atol=ATOL_U, rtol=0)
    np.testing.assert_allclose(out["u_nom"][:, 0], gold["u_nom"][:, 0], atol=ATOL_U, rtol=0)     # u*_0
    np.testing.assert_allclose(out["x_nom0"], gold["x_nom0"], atol=1e-12, rtol=0)
    np.testing.assert_allclose(out["xu_ss"], gold["xu_ss"], atol=ATOL_SS, rtol=0)
    assert out["iters"].max() < 40


def test_live_oracle_on_fresh_states(cartpole):
    """Seeded disturbed closed loops (states the fixture does not contain)."""
    mpc, w, orc = cartpole
    rng = np.random.default_rng(20240301)
    idx = rng.integers(0, len(S), 256)
    X = S[idx, :4].copy()
    R = S[idx, 4:].copy()
    R[:, 0] += rng.uniform(-0.3, 0.3, len(idx))          # move the reference, keep x_k feasible
    hip = mpc._solve(X, R)
    ref = orc.solve(X, R)
    assert np.array_equal(hip["status"], ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() > 200
    np.testing.assert_allclose(hip["u_nom"][ok], ref["u_nom"][ok], atol=ATOL_U, rtol=0)
    np.testing.assert_allclose(hip["xu_ss"][ok], ref["xu_ss"][ok], atol=ATOL_SS, rtol=0)
    np.testing.assert_allclose(hip["x_nom"][ok], ref["x_nom"][ok], atol=1e-8, rtol=0)


def test_kkt_certificate_of_hip_outputs(cartpole):
    """Independent of the oracle's solver: the HIP outputs satisfy the KKT conditions of the QP
    as the reference states it (TubeTrackingMPC.py:104-156)."""
    mpc, _, _ = cartpole
    p = mpc._problem_dict()
    idx = np.arange(3, len(S), 37)
    out = mpc._solve(S[idx, :4], S[idx, 4:])
    for k, i in enumerate(idx):
        qp = qp_sparse.build_sparse_qp(p, S[i, :4], S[i, 4:])
        v = qp_sparse.pack(qp, out["x_nom"][k], out["u_nom"][k], out["x_ss"][k], out["u_ss"][k])
        c = qp_sparse.kkt_certificate(qp, v)
        assert c["r_eq"] < 1e-9 and c["r_ineq"] < 1e-9 and c["r_stat"] < 1e-7 and c["min_lam"] >= 0, c


def test_edge_cases_infeasible_unconstrained_empty(cartpole, hip_lib):
    mpc, _, orc = cartpole
    X = np.array([[0.0, 0.0, 0.2, 0.0],        # angle outside the tightened set, x_0 fixed -> infeasible
                  [0.5, 0.0, 0.0, 0.0],        # at the reference: unconstrained minimiser feasible
                  [0.0, 0.0, 0.0, 0.0]])
    R = np.array([[0.5, 0, 0, 0.0]] * 3)
    out = mpc._solve(X, R)
    ref = orc.solve(X, R)
    assert list(out["status"]) == [2, 0, 0] == list(ref["status"])
    assert np.all(np.isnan(out["u_nom"][0])) and np.all(np.isnan(out["xu_ss"][0])) and np.all(np.isnan(out["x_nom"][0]))
    assert out["iters"][1] == 0
    np.testing.assert_allclose(out["u_nom"][1], 0.0, atol=1e-9)
    np.testing.assert_allclose(out["u_nom"][2], ref["u_nom"][2], atol=ATOL_U)
    # a NaN-poisoned neighbour must not leak: instance 2 solved alone gives the same bits
    alone = mpc._solve(X[2:3], R[2:3])
    assert np.array_equal(alone["u_nom"][0], out["u_nom"][2])
    # empty batch
    e = mpc._solve(np.zeros((0, 4)), np.zeros((0, 4)))
    assert e["u_nom"].shape == (0, 10, 1) and e["status"].shape == (0,)


def test_reference_api_shapes_single_instance(cartpole):
    """solve_optimization_problem / determine_packet with the reference's 1-D calling convention
    (TubeTrackingMPC.py:170-227)."""
    mpc, _, orc = cartpole
    x, r = S[17, :4].copy(), S[17, 4:].copy()
    x_nom, u_nom, x_ss, u_ss = mpc.solve_optimization_problem(x, r)
    assert x_nom.shape == (4, 11) and u_nom.shape == (1, 10) and x_ss.shape == (4,) and u_ss.shape == (1,)
    ref = orc.solve(x[None], r[None])
    np.testing.assert_allclose(u_nom[0], ref["u_nom"][0, :, 0], atol=ATOL_U)
    pkt = mpc.determine_packet(x.reshape(4, 1), r, 7)
    assert pkt["q_t"] == 7 and pkt["U_t"].shape == (1, 11)
    np.testing.assert_allclose(pkt["U_t"][0, -1], u_ss[0] + (mpc._K @ x_ss)[0], atol=1e-9)       # TubeTrackingMPC.py:217
    assert len(mpc.get_computational_times()) == 1
    mpc.reset_computational_times()
    # infeasible -> four None / U_t None (TubeTrackingMPC.py:189-194, :220-221)
    res = mpc.solve_optimization_problem(np.array([0.0, 0.0, 0.2, 0.0]), r)
    assert res == (None, None, None, None)
    assert mpc.determine_packet(np.array([0.0, 0.0, 0.2, 0.0]), r, 0)["U_t"] is None


def test_full_batch_properties(cartpole):
    """BASELINE config 2 size (B=4096): size-independent properties instead of a 4096-instance
    oracle run: determinism, permutation equivariance, constraint satisfaction, dynamics."""
    mpc, w, _ = cartpole
    rng = np.random.default_rng(7)
    idx = rng.integers(0, len(S), 4096)
    X, R = S[idx, :4].copy(), S[idx, 4:].copy()
    a = mpc._solve(X, R)
    b = mpc._solve(X, R)
    assert np.array_equal(a["u_nom"], b["u_nom"]) and np.array_equal(a["iters"], b["iters"])     # deterministic
    perm = rng.permutation(4096)
    c = mpc._solve(X[perm], R[perm])
    assert np.array_equal(c["u_nom"], a["u_nom"][perm])                                          # equivariant
    assert np.all(a["status"] == 0)
    # duplicates of one instance inside the batch agree bit-for-bit with the fixture solve
    first = {}
    for k, i in enumerate(idx):
        if i in first:
            assert np.array_equal(a["u_nom"][k], a["u_nom"][first[i]])
        else:
            first[i] = k
    # outputs obey the model and the tightened constraints of the reference formulation
    A, Bm = w["A"], w["B"]
    xn, un = a["x_nom"], a["u_nom"]
    np.testing.assert_allclose(xn[:, 0], X, atol=1e-12)
    np.testing.assert_allclose(xn[:, 1:], xn[:, :-1] @ A.T + un @ Bm.T, atol=1e-9)
    assert np.max(np.abs(un)) <= mpc._Uc.b.max() + 1e-9
    assert np.max(xn[:, :-1] @ mpc._Xc.A.T - mpc._Xc.b) < 1e-9
    st = np.c_[xn[:, -1], a["xu_ss"]]
    assert np.max(st @ mpc._Xf.A.T - mpc._Xf.b) < 1e-8
    ss = a["xu_ss"]
    np.testing.assert_allclose(ss[:, :4] @ (A - np.eye(4)).T + ss[:, 4:] @ Bm.T, 0, atol=1e-9)   # steady state (:147)


def test_device_pointer_entry_with_torch(cartpole, hip_lib):
    """tmpc_solve_batch_device on torch-allocated HBM buffers == host-pointer entry."""
    import torch
    mpc, _, _ = cartpole
    h = mpc._handle
    B = 512
    X, R = S[:B, :4].copy(), S[:B, 4:].copy()
    dev = torch.device("cuda:0")
    x = torch.from_numpy(X).to(dev)
    r = torch.from_numpy(R).to(dev)
    u = torch.empty((B, 10, 1), dtype=torch.float64, device=dev)
    x0 = torch.empty((B, 4), dtype=torch.float64, device=dev)
    ss = torch.empty((B, 5), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    hip_lib.solve_batch_device(h, B, x.data_ptr(), r.data_ptr(), None, u.data_ptr(), x0.data_ptr(), ss.data_ptr(), None,
                               st.data_ptr(), it.data_ptr())
    hip_lib.synchronize(h)
    assert hip_lib.last_kernel_ms(h) > 0
    host = mpc._solve(X, R, want_traj=False)
    assert np.array_equal(u.cpu().numpy(), host["u_nom"])
    assert np.array_equal(st.cpu().numpy(), host["status"])


def test_variant_ids_without_a_problem_are_flagged(cartpole, hip_lib):
    """include/tmpc.h: on the device-pointer entry an instance whose variant id names no problem of the handle comes back with
    status NUMERICAL and NaN outputs (no kernel solves it), its neighbours are unaffected; the host-pointer entry refuses the
    call."""
    import torch
    mpc, _, _ = cartpole
    h = mpc._handle
    X, R = S[:8, :4].copy(), S[:8, 4:].copy()
    var = np.zeros(8, np.uint8)
    var[[2, 5]] = [1, 7]                              # the plain controller has one problem (variant 0)
    base = mpc._solve(X, R, want_traj=False)
    with pytest.raises(RuntimeError):
        mpc._solve(X, R, var)
    dev = torch.device("cuda:0")
    x, r, v = (torch.from_numpy(a).to(dev) for a in (X, R, var))
    u = torch.zeros((8, 10, 1), dtype=torch.float64, device=dev)
    x0 = torch.zeros((8, 4), dtype=torch.float64, device=dev)
    ss = torch.zeros((8, 5), dtype=torch.float64, device=dev)
    st = torch.zeros(8, dtype=torch.int32, device=dev)
    it = torch.zeros(8, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    hip_lib.solve_batch_device(h, 8, x.data_ptr(), r.data_ptr(), v.data_ptr(), u.data_ptr(), x0.data_ptr(), ss.data_ptr(), None,
                               st.data_ptr(), it.data_ptr())
    hip_lib.synchronize(h)
    un, stn = u.cpu().numpy(), st.cpu().numpy()
    assert list(stn[[2, 5]]) == [3, 3] and np.all(np.isnan(un[[2, 5]])) and np.all(np.isnan(ss.cpu().numpy()[[2, 5]]))
    keep = [0, 1, 3, 4, 6, 7]
    assert np.array_equal(un[keep], base["u_nom"][keep]) and np.all(stn[keep] == 0)


def test_reference_horizon_N20(hip_lib, oracle_lib):
    """The horizon the reference's cartpole scripts use (results_linear_system.py:64, N = 20):
    nv = 21, 184 dense rows + the 420-row terminal block in factored form."""
    S20 = common.harvest_states("cartpole", 20, True, [[0.5, 0.0], [3.0, 0.0], [-2.0, 1.0]], steps=40)
    mpc, _ = common.make_mpc("cartpole", 20, True, create=True)
    assert hip_lib.get_dims(mpc._handle)[0] == 21
    ref = Oracle(mpc._problem_dict()).solve(S20[:, :4], S20[:, 4:])
    out = mpc._solve(S20[:, :4], S20[:, 4:])
    assert np.array_equal(out["status"], ref["status"]) and np.all(ref["status"] == 0)
    np.testing.assert_allclose(out["u_nom"], ref["u_nom"], atol=ATOL_U, rtol=0)
    np.testing.assert_allclose(out["xu_ss"], ref["xu_ss"], atol=ATOL_SS, rtol=0)
    np.testing.assert_allclose(out["x_nom"], ref["x_nom"], atol=1e-8, rtol=0)


@pytest.mark.parametrize("N", [5, 10])
def test_config1_double_integrator_closed_loop(hip_lib, oracle_lib, N):
    """BASELINE config 1: Example_of_Tube_Tracking_MPC.py (free initial state, Rakovic sets,
    x0=[1,2], T=120, reference 5/-9/9/4, w ~ default_rng(1).uniform(-0.1,0.1)); N=5 as BASELINE
    says and N=10 as the example does.  Every step's QP: HIP == oracle; applied input stays in U
    (the example's own runtime check, :99-100)."""
    mpc, w = common.make_mpc("double_integrator", N, False, create=True)
    orc = Oracle(mpc._problem_dict())
    A, B, K = w["A"], w["B"], mpc.get_ancillary_controller_gain()
    rng = np.random.default_rng(1)
    x = np.array([1.0, 2.0])
    T = 120
    ref = np.r_[5 * np.ones(30), -9 * np.ones(30), 9 * np.ones(30), 4 * np.ones(30)]
    worst = 0.0
    for t in range(T):
        r = np.array([ref[t], 0.0])
        x_nom, u_nom, x_ss, u_ss = mpc.solve_optimization_problem(x.copy(), r)
        o = orc.solve(x[None], r[None])
        assert o["status"][0] == 0
        worst = max(worst, np.abs(u_nom[0] - o["u_nom"][0, :, 0]).max(), np.abs(x_nom[:, 0] - o["x_nom0"][0]).max())
        u = u_nom[:, 0] - K @ (x - x_nom[:, 0])
        assert w["U"].contains(u, 1e-7)
        assert mpc._Z.contains(x - x_nom[:, 0], 1e-7)                  # x_k in x_0 (+) Z  (:132)
        x = A @ x + B @ u + rng.uniform(-0.1, 0.1, 2)
    assert worst < ATOL_U
    assert abs(x[0] - 4.0) < 0.5


def test_block_kernel_matches_fixture_and_wave_kernel(cartpole, hip_lib):
    """The workgroup-per-QP kernel (csrc/tmpc_block.hip: MFMA G'DG, LDS Cholesky), forced onto the bench
    problem: same minimisers as the fixture, same statuses and iteration counts as the wave kernel on the
    edge cases."""
    mpc, _, _ = cartpole
    gold = np.load(os.path.join(common.GOLDEN, "cartpole_N10_oracle.npz"))
    X = np.r_[S[:, :4], [[0.0, 0.0, 0.2, 0.0], [0.5, 0.0, 0.0, 0.0]]]      # + infeasible, + unconstrained
    R = np.r_[S[:, 4:], [[0.5, 0, 0, 0.0], [0.5, 0, 0, 0.0]]]
    assert mpc.get_kernel_path() == "wave"
    base = mpc._solve(X, R)
    mpc.set_kernel_path("block")
    try:
        assert mpc.get_kernel_path() == "block"
        out = mpc._solve(X, R)
    finally:
        mpc.set_kernel_path("auto")
    assert np.array_equal(out["status"], base["status"]) and list(out["status"][-2:]) == [2, 0]
    assert out["iters"][-1] == 0 and np.all(np.isnan(out["u_nom"][-2]))
    np.testing.assert_allclose(out["u_nom"][:600], gold["u_nom"], atol=ATOL_U, rtol=0)
    np.testing.assert_allclose(out["xu_ss"][:600], gold["xu_ss"], atol=ATOL_SS, rtol=0)
    np.testing.assert_allclose(out["x_nom"], base["x_nom"], atol=1e-8, rtol=0, equal_nan=True)
    assert np.abs(out["iters"][:600].astype(int) - base["iters"][:600]).max() <= 1


@pytest.mark.parametrize("N", [10, 20])
def test_extended_controller_both_problems(hip_lib, oracle_lib, N):
    """BASELINE config 3: ExtendedTubeTrackingMPC (TubeTrackingMPC.py:249-369), gamma_t per instance.  The
    packet-received problem (Z (-) W on x_0 as the factored block, auxiliaries of :293 eliminated) has its own wave-kernel shape."""
    mpc, _ = common.make_mpc("cartpole", N, True, extended=True, create=True)
    nv1, nc1, _ = hip_lib.get_dims(mpc._handle, 1)
    assert nv1 == N + 1 + 4 and nc1 > 900 and mpc.get_kernel_path(1) == "wave"
    assert hip_lib.kernel_name(mpc._handle, 1).startswith("tmpc::solve_kernel<%d,%d,0,4,7,0," % ((15, 1) if N == 10 else (26, 2)))      # the shape of exactly nv = N + 5 variables
    SX = common.harvest_states("cartpole", N, True, [[0.5], [-0.4, 0.3], [0.2, -0.5, 0.1]], 40, seed=4, disturb=True, extended=True)
    gam = np.random.default_rng(1).integers(0, 2, len(SX)).astype(np.uint8)
    ref = Oracle(mpc._problem_dict()).solve(SX[:, :4], SX[:, 4:], gam)
    x_nom, u_nom, x_ss, u_ss = mpc.solve_optimization_problem(SX[:, :4], SX[:, 4:], gam)
    assert np.array_equal(mpc.last_status, ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() > 30 and (ok & (gam == 1)).sum() > 10 and (~ok).sum() > 0        # infeasible instances (status 2) are part of the mix
    np.testing.assert_allclose(u_nom[ok], ref["u_nom"][ok], atol=1e-7, rtol=0)
    np.testing.assert_allclose(u_nom[ok, 0], ref["u_nom"][ok, 0], atol=ATOL_U, rtol=0)          # u*_0
    np.testing.assert_allclose(x_nom[ok, 0], ref["x_nom0"][ok], atol=1e-8, rtol=0)
    np.testing.assert_allclose(np.c_[x_ss, u_ss][ok], ref["xu_ss"][ok], atol=ATOL_SS, rtol=0)
    # gamma = 0 instances keep x_0 = x_k, gamma = 1 instances move it inside x_k (+) (Z (-) W)  (:278)
    g0, g1 = ok & (gam == 0), ok & (gam == 1)
    np.testing.assert_allclose(x_nom[g0, 0], SX[g0, :4], atol=1e-12)
    assert np.all(mpc._ZmW.contains((SX[g1, :4] - x_nom[g1, 0]).T, 1e-9))
    # the reference's single-instance API (:351-369): packet dict + x_nom_0
    i = int(np.flatnonzero(g1)[0])
    packet, x0 = mpc.determine_packet(SX[i, :4], SX[i, 4:], 7, gamma_t=1)
    assert packet["U_t"].shape == (1, N + 1) and packet["q_t"] == 7
    np.testing.assert_allclose(x0, ref["x_nom0"][i], atol=1e-8)


def test_config5_synthetic_n12_m4_N30(hip_lib, oracle_lib):
    """BASELINE config 5: random stable (A, B), n = 12, m = 4, N = 30 -> nv = 124, ~1.2e3 rows; block kernel,
    G'DG on the FP64 matrix cores."""
    mpc, _ = common.make_mpc("synthetic", 30, True, create=True)
    nv, nc, npar = hip_lib.get_dims(mpc._handle)
    assert nv == 124 and npar == 24 and mpc.get_kernel_path() == "block"
    rng = np.random.default_rng(0)
    B = 192
    X = rng.uniform(-0.5, 0.5, (B, 12)) * mpc._Xc.b[:12]
    X[:64] *= 1.9                                         # near the boundary of Xc: many active rows
    R = np.zeros((B, 12))
    R[:, 0] = rng.uniform(-2, 2, B)
    ref = Oracle(mpc._problem_dict()).solve(X, R)
    out = mpc._solve(X, R)
    assert np.array_equal(out["status"], ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() > 100 and ref["iters"][ok].max() >= 8
    np.testing.assert_allclose(out["u_nom"][ok], ref["u_nom"][ok], atol=ATOL_U, rtol=0)
    np.testing.assert_allclose(out["xu_ss"][ok], ref["xu_ss"][ok], atol=ATOL_SS, rtol=0)
    np.testing.assert_allclose(out["x_nom"][ok], ref["x_nom"][ok], atol=1e-8, rtol=0)


@pytest.mark.parametrize("name,fixed,horizons", [("cartpole", True, (3, 6, 9, 12, 15)),
                                                  ("cartpole", True, (17, 20, 23, 26)),
                                                  ("double_integrator", False, (3, 5, 8, 11, 14)),
                                                  ("double_integrator_darup", False, (4, 7, 10)),
                                                  ("double_integrator", True, (4, 9, 13))])
def test_horizon_sweep_covers_the_kernel_shapes(hip_lib, oracle_lib, name, fixed, horizons):
    """The offline sets do not depend on N, so the horizon can be swept freely: every N lands on some compiled shape of the
    wave kernel (or on the block kernel) and must reproduce the oracle."""
    rng = np.random.default_rng(7)
    seen = set()
    for N in horizons:
        mpc, w = common.make_mpc(name, N, fixed, create=True)
        nv, nc, _ = hip_lib.get_dims(mpc._handle)
        seen.add((mpc.get_kernel_path(), nv))
        nx = w["A"].shape[0]
        if name == "cartpole":
            idx = rng.integers(0, len(S), 48)
            X, R = S[idx, :4].copy(), S[idx, 4:].copy()
            R[:, 0] += rng.uniform(-0.2, 0.2, 48)
        else:
            X = rng.uniform(-1, 1, (48, nx)) * [5.0, 0.7]
            R = np.c_[rng.uniform(-7, 7, 48), np.zeros(48)]
        ref = Oracle(mpc._problem_dict()).solve(X, R)
        out = mpc._solve(X, R)
        assert np.array_equal(out["status"], ref["status"]), (N, out["status"], ref["status"])
        ok = ref["status"] == 0
        assert ok.sum() >= 24
        np.testing.assert_allclose(out["u_nom"][ok], ref["u_nom"][ok], atol=ATOL_U, rtol=0, err_msg=f"N={N}")
        np.testing.assert_allclose(out["xu_ss"][ok], ref["xu_ss"][ok], atol=ATOL_SS, rtol=0, err_msg=f"N={N}")
        np.testing.assert_allclose(out["x_nom"][ok], ref["x_nom"][ok], atol=1e-8, rtol=0, err_msg=f"N={N}")
    assert len(seen) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("N", [10, 20])
def test_full_size_batch_65536(hip_lib, oracle_lib, N):
    """BASELINE's full batch size through the wave kernel (N = 10: two waves per SIMD; N = 20, the reference's horizon,
    results_linear_system.py:64: one wave per SIMD): perturbed closed-loop states, some of them outside the feasible set.
    Size-independent properties on all 65536 instances -- every status is 0 / 2, optimal solutions respect the input box,
    a permuted batch gives the permuted answer, the block kernel agrees on a slice -- and a 2048-instance sample against the oracle."""
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))        # (x_hat, ref) pairs of closed loops
    mpc, w = common.make_mpc("cartpole", N, True, create=True)
    rng = np.random.default_rng(2024 + N)
    B = 65536
    idx = rng.integers(0, len(S), B)
    X = S[idx, :4] + rng.uniform(-1, 1, (B, 4)) * 0.5 * w["w_bound"]
    R = S[idx, 4:].copy()
    R[:, 0] += rng.uniform(-0.5, 0.5, B)
    big = mpc._solve(X, R, want_traj=False)
    assert hip_lib.kernel_name(mpc._handle).startswith("tmpc::solve_kernel")
    st = big["status"]
    assert np.all((st == 0) | (st == 1) | (st == 2)) and np.mean(st == 1) < 1e-3 and np.mean(st == 0) > 0.5
    good = st == 0
    assert np.all(np.abs(big["u_nom"][good]) <= 10.0 + 1e-7)                 # U = [-10, 10] (results_linear_system.py:104-106)
    assert np.all(np.isnan(big["u_nom"][st == 2]))
    perm = rng.permutation(B)
    again = mpc._solve(X[perm], R[perm], want_traj=False)
    assert np.array_equal(again["status"], st[perm])
    np.testing.assert_allclose(again["u_nom"][good[perm]], big["u_nom"][perm][good[perm]], rtol=0, atol=1e-9)
    mpc.set_kernel_path("block")
    blk = mpc._solve(X[:1024], R[:1024], want_traj=False)
    mpc.set_kernel_path("auto")
    both = (blk["status"] == 0) & good[:1024]
    assert np.mean(blk["status"] == st[:1024]) > 0.995
    np.testing.assert_allclose(blk["u_nom"][both], big["u_nom"][:1024][both], rtol=0, atol=1e-8)
    sub = rng.choice(B, 2048, replace=False)
    ref = Oracle(mpc._problem_dict()).solve(X[sub], R[sub])
    # certified-optimal vs uncertified (status 1) may differ between the implementations on a few instances; infeasibility may not
    assert np.array_equal(st[sub] == 2, ref["status"] == 2)
    ok = (st[sub] == 0) & (ref["status"] == 0)
    assert ok.sum() > 1000
    np.testing.assert_allclose(big["u_nom"][sub][ok], ref["u_nom"][ok], rtol=0, atol=1e-8)
