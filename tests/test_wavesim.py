"""The wave-per-QP kernel's SOURCE (csrc/tmpc_kernels.hip, the text the GPU build compiles) on a host execution model, under
sanitizers.  CPU only; no GPU sanitizer exists on the pool, and a build of this kernel that spilled 126 registers once
returned wrong statuses on the GPU: was that the spill code, or undefined behaviour in the source that only some register
allocations expose?  tests/wavesim/hip_sim.hpp runs every lane as a fiber and makes the wavefront's guarantees explicit
(cross-lane operations, LDS fences and barriers are rendezvous), so that

* AddressSanitizer + UndefinedBehaviorSanitizer see every LDS / global access and every arithmetic operation,
* MemorySanitizer sees every read of a register or LDS word that was never written (LDS, outputs and the work space
  start out poisoned),
* an LDS hand-over between lanes without a fence in between reads stale data (and the answers differ from the oracle's),
* a cross-lane operation under divergent control flow is a reported deadlock.

The layouts the kernel reads are the product's own (tmpc_create on a host-only handle, tmpc_debug_dump_layout)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import common

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "wavesim"))
import run_case  # noqa: E402

S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
SAN_ENV = {"ASAN_OPTIONS": "detect_stack_use_after_return=0:detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
           "MSAN_OPTIONS": "halt_on_error=1"}
CLEAN_MARKERS = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: MemorySanitizer", "ERROR: LeakSanitizer")


def _build_or_skip():
    """The harness needs a host clang++ with the x86-64 sanitizer runtimes (ASan / UBSan / MSan); a toolchain without them
    skips these tests instead of failing the suite (the product build does not depend on them)."""
    import subprocess
    try:
        return run_case.build_all()
    except (subprocess.CalledProcessError, OSError) as e:
        pytest.skip(f"tests/wavesim does not build on this host: {e}")


@pytest.fixture(scope="module")
def binaries():
    # (also built by __graft_entry__.build(); make leaves them alone when they are up to date)
    return _build_or_skip()


@pytest.fixture(scope="module")
def cartpole():
    mpc, _ = common.make_mpc("cartpole", 10, True, create=False)
    return mpc._problem_dict()


def assert_clean(out):
    for m in CLEAN_MARKERS:
        assert m not in out["stderr"], out["stderr"][-4000:]


def test_all_fixture_instances_through_the_kernel_source(binaries, cartpole, oracle_lib):
    """All 600 closed-loop states of the committed fixture: statuses bit-identical with the oracle's, u_nom to 1e-8 (the GPU
    parity tolerance), iteration counts in the same range.  Eight processes, one workgroup of eight waves each."""
    from oracle.oracle import Oracle
    ref = Oracle(cartpole).solve(S[:, :4], S[:, 4:])
    chunks = np.array_split(np.arange(len(S)), 8)
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(lambda ii: run_case.run(binaries["wavesim"], cartpole, S[ii, :4], S[ii, 4:]), chunks))
    st = np.concatenate([o["status"] for o in outs])
    u = np.concatenate([o["u_nom"] for o in outs])
    it = np.concatenate([o["iters"] for o in outs])
    assert np.array_equal(st, ref["status"]) and np.all(st == 0)
    assert np.max(np.abs(u - ref["u_nom"])) < 1e-8
    assert np.max(np.abs(np.concatenate([o["xu_ss"] for o in outs]) - ref["xu_ss"])) < 1e-8
    assert it.min() >= 0 and it.max() < 40 and abs(it.mean() - ref["iters"].mean()) < 2.0
    assert sum(o["rendezvous"] for o in outs) > 1e6          # the wavefront model was exercised, not bypassed


def test_address_and_undefined_behaviour_sanitizers_are_clean(binaries, cartpole, oracle_lib):
    from oracle.oracle import Oracle
    idx = np.arange(0, 600, 10)              # 60 instances, transients and settled states alike
    parts = np.array_split(idx, 6)
    with ThreadPoolExecutor(6) as ex:
        outs = list(ex.map(lambda ii: run_case.run(binaries["wavesim_asan"], cartpole, S[ii, :4], S[ii, 4:], env=SAN_ENV), parts))
    for o in outs:
        assert_clean(o)
    ref = Oracle(cartpole).solve(S[idx, :4], S[idx, 4:])
    assert np.array_equal(np.concatenate([o["status"] for o in outs]), ref["status"])
    assert np.max(np.abs(np.concatenate([o["u_nom"] for o in outs]) - ref["u_nom"])) < 1e-8


def test_memory_sanitizer_is_clean(binaries, cartpole, oracle_lib):
    """No register, LDS word or output element is read before it is written (the outputs are checked when the program
    writes them to its file)."""
    from oracle.oracle import Oracle
    idx = np.arange(3, 600, 15)              # 40 instances
    parts = np.array_split(idx, 5)
    with ThreadPoolExecutor(5) as ex:
        outs = list(ex.map(lambda ii: run_case.run(binaries["wavesim_msan"], cartpole, S[ii, :4], S[ii, 4:], env=SAN_ENV), parts))
    for o in outs:
        assert_clean(o)
    ref = Oracle(cartpole).solve(S[idx, :4], S[idx, 4:])
    assert np.array_equal(np.concatenate([o["status"] for o in outs]), ref["status"])


def test_edge_cases_and_the_dense_single_shape_under_sanitizers(binaries, oracle_lib):
    """BASELINE config 1 (double integrator, free x_0: every row dense and single, no factored block: another instantiation)
    with feasible, trivially optimal and infeasible instances, all three builds."""
    from oracle.oracle import Oracle
    mpc, _ = common.make_mpc("double_integrator", 5, False, create=False)
    d = mpc._problem_dict()
    rng = np.random.default_rng(4)
    X = np.r_[rng.uniform(-1, 1, (20, 2)) * [7.5, 0.9], [[0.0, 0.0]], [[30.0, 0.0]], [[-3.1437307257161446, 0.5378281719126672]]]
    R = np.c_[np.r_[rng.uniform(-9, 9, 20), 0.0, 0.0, 4.697848229977945], np.zeros(len(X))]
    ref = Oracle(d).solve(X, R)
    assert 0 in ref["status"] and 2 in ref["status"] and ref["iters"][20] == 0
    for t in ("wavesim", "wavesim_asan", "wavesim_msan"):
        o = run_case.run(binaries[t], d, X, R, env=SAN_ENV)
        assert_clean(o)
        assert np.array_equal(o["status"], ref["status"]), (t, o["status"], ref["status"])
        ok = ref["status"] == 0
        assert np.max(np.abs(o["u_nom"][ok] - ref["u_nom"][ok])) < 1e-8
        assert np.all(np.isnan(o["u_nom"][~ok]))


# ---------------------------------------------------------------- the workgroup-per-QP kernel (csrc/tmpc_block.hip)
def test_fused_closed_loop_source_under_sanitizers(binaries, cartpole, oracle_lib):
    """closed_loop_kernel (csrc/tmpc_fused.hip: the wave kernel's body with the trajectory's state machines of tmpc_mc_step.hpp between
    two solves, a wave per trajectory for all T steps) on the host execution model: the loop equals the numpy state machines driven by
    the ORACLE's solves (the reference's loop body, results_linear_system.py:209-259), and ASan / UBSan / MSan see every access of the
    state machines -- their LDS hand-overs, the packet buffer, the statistics -- with the per-trajectory arrays as exact-size heap
    blocks and the solve's outputs poisoned until written."""
    from LinearMPCOverNetworks import montecarlo
    from oracle.oracle import Oracle
    nb, T = 8, 12
    mpc, w = common.make_mpc("cartpole", 10, True)
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=7)
    ref = np.where(np.arange(T) < 6, 0.5, -0.3)
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    orc = Oracle(cartpole)

    def packets(x_hat, r, gamma=None):
        sol = orc.solve(x_hat, r, gamma)
        u_ss = sol["u_ss"] + sol["x_ss"] @ mpc._K.T
        return np.ascontiguousarray(np.concatenate([sol["u_nom"], u_ss[:, None, :]], axis=1).transpose(0, 2, 1)), sol["x_nom0"], sol["status"]
    host = montecarlo.run_remote_tube_mpc(packets, w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist)
    jobs = [("wavesim", False), ("wavesim_asan", True), ("wavesim_msan", False)]
    with ThreadPoolExecutor(3) as ex:
        outs = list(ex.map(lambda j: run_case.run_loop(binaries[j[0]], cartpole, K, Kp, mpc._Z, p_loss, ref, th, ga, dist, warm=j[1], env=SAN_ENV), jobs))
    for (name, warm), o in zip(jobs, outs):
        assert_clean(o)
        np.testing.assert_allclose(o["x_final"], host["x_final"], atol=1e-8, rtol=0, err_msg=name)
        np.testing.assert_allclose(o["tracking_error"], host["tracking_error"], atol=1e-10, rtol=0, err_msg=name)
        assert np.array_equal(o["tube_violations"], host["tube_violations"]) and np.all(o["tube_violations"] == 0), name
        assert np.array_equal(o["not_optimal"], host["not_optimal"]), name
        assert o["consistent"].max() < 1e-9, name                       # Proposition 1
    # the warm-started loop does the same steps with fewer interior-point iterations
    assert outs[1]["iters_sum"].sum() < outs[0]["iters_sum"].sum()
    assert np.array_equal(outs[0]["iters_sum"], outs[2]["iters_sum"])


def test_extended_closed_loop_source_under_sanitizers(binaries, oracle_lib):
    """closed_loop_step_kernel -- the extended controller's form: per time step one launch per problem (base / packet-received), each QP
    followed by its trajectory's state machines, the arrival flags in two alternating buffers (csrc/tmpc_api.cpp: mc_run_impl; the harness
    steps the two launches the same way) -- on the host execution model under ASan + UBSan, warm-started: equal to the numpy state
    machines (RobustEstimator, ConsistentActuator with x_nom_0 adoption) driven by the ORACLE's solves of
    results_linear_system_with_extendedMPC.py:247-378."""
    from LinearMPCOverNetworks import montecarlo
    from oracle.oracle import Oracle
    nb, T = 8, 8
    mpc, w = common.make_mpc("cartpole", 10, True, extended=True)
    d = mpc._problem_dict()
    p_loss = np.tile([0.0, 0.3, 0.6, 0.9], nb // 4)
    th, ga, dist = montecarlo.draw_realisations(nb, T, w["w_bound"], seed=9)
    ref = np.where(np.arange(T) < 4, 0.5, -0.3)
    K, Kp = mpc.get_steady_state_controller_gain(), mpc.get_ancillary_controller_gain()
    orc = Oracle(d)
    seen = []

    def packets(x_hat, r, gamma=None):
        seen.append(np.array(gamma))
        sol = orc.solve(x_hat, r, gamma)
        u_ss = sol["u_ss"] + sol["x_ss"] @ mpc._K.T
        return np.ascontiguousarray(np.concatenate([sol["u_nom"], u_ss[:, None, :]], axis=1).transpose(0, 2, 1)), sol["x_nom0"], sol["status"]
    host = montecarlo.run_remote_tube_mpc(packets, w["A"], w["B"], K, Kp, 10, mpc._Z, p_loss, ref, th, ga, dist, extended=True)
    assert {0, 1} <= set(np.concatenate(seen).tolist())                    # both problems are in use
    o = run_case.run_loop(binaries["wavesim_ext_asan"], d, K, Kp, mpc._Z, p_loss, ref, th, ga, dist, warm=True, extended=True, env=SAN_ENV)
    assert_clean(o)
    np.testing.assert_allclose(o["x_final"], host["x_final"], atol=1e-8, rtol=0)
    np.testing.assert_allclose(o["tracking_error"], host["tracking_error"], atol=1e-10, rtol=0)
    assert np.array_equal(o["tube_violations"], host["tube_violations"]) and np.all(o["tube_violations"] == 0)
    assert np.array_equal(o["not_optimal"], host["not_optimal"])


@pytest.fixture(scope="module")
def block_binaries():
    return _build_or_skip()


def test_block_kernel_source_under_sanitizers(block_binaries, cartpole, oracle_lib):
    """solve_block_kernel<1> (256 threads; the cart-pole through the general path) and solve_block_kernel<8> (512 threads, BASELINE
    config 5: n = 12, m = 4, N = 30, 124 variables) on the host execution model: answers equal the oracle's, ASan / UBSan / MSan
    report nothing -- the per-workgroup workspace, the LDS and the outputs start out poisoned."""
    from oracle.oracle import Oracle
    mpc5, _ = common.make_mpc("synthetic", 30, True, create=False)
    d5 = mpc5._problem_dict()
    rng = np.random.default_rng(5)
    X5 = rng.uniform(-0.5, 0.5, (4, 12)) * mpc5._Xc.b[:12]
    X5[:2] *= 1.9
    R5 = np.zeros((4, 12))
    R5[:, 0] = rng.uniform(-2, 2, 4)
    idx = np.linspace(0, 599, 16).astype(int)
    jobs = [("blocksim", cartpole, S[idx, :4], S[idx, 4:]), ("blocksim_asan", cartpole, S[idx[::3], :4], S[idx[::3], 4:]),
            ("blocksim_msan", cartpole, S[idx[1::3], :4], S[idx[1::3], 4:]), ("blocksim", d5, X5, R5), ("blocksim_asan", d5, X5[:2], R5[:2]),
            ("blocksim_msan", d5, X5[1:], R5[1:])]
    with ThreadPoolExecutor(6) as ex:
        outs = list(ex.map(lambda j: run_case.run(block_binaries[j[0]], j[1], j[2], j[3], env=SAN_ENV, block=True), jobs))
    nontrivial = 0
    for (name, d, X, R), o in zip(jobs, outs):
        assert_clean(o)
        ref = Oracle(d).solve(X, R)
        assert np.array_equal(o["status"], ref["status"]) and np.all(o["status"] == 0), (name, o["status"])
        assert np.max(np.abs(o["u_nom"] - ref["u_nom"])) < 1e-8, name
        nontrivial += int((o["iters"] > 0).sum())
        assert o["rendezvous"] > 1000
    assert nontrivial >= 20


# ---------------------------------------------------------------------------------------------------------------------
# the batched LP kernel of the offline stage (csrc/tmpc_lp.hip) on the same execution model
def _highs(H, h, C, rel, relax_by=1.0):
    from scipy.optimize import linprog
    val = np.empty(len(C))
    for k, c in enumerate(C):
        hk = h.copy()
        if rel is not None and rel[k] >= 0:
            hk[rel[k]] += relax_by
        # (HiGHS' default feasibility tolerance of 1e-7 leaves 3e-8 of violation, and of value, on these sets)
        res = linprog(-c, A_ub=H, b_ub=hk, bounds=(None, None), method="highs",
                      options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
        assert res.status == 0
        val[k] = -res.fun
    return val


def test_lp_kernel_source_under_asan_degenerate_faces(binaries):
    """tests/golden/lp_degenerate_cases.npz: support functions of the cartpole's terminal sets (452 and 588 rows, d = 9) in
    directions whose optimal face has dimension >= 1 -- the normal matrix loses its rank as the gap closes.  The round-2
    kernel divided by pivots that were round-off, drifted along the face, never got its dual residual back and returned
    the iterate at the iteration cap (status 1, value off by 6e-9); with the skipped pivots (lp_factor) the vertex steps
    finish.  Values against HiGHS run with 1e-10 feasibility tolerances: 1e-9 relative."""
    Z = np.load(os.path.join(common.GOLDEN, "lp_degenerate_cases.npz"))
    for name in ("a", "b"):
        H, h, C, rel = (Z[f"{name}_{k}"] for k in ("H", "h", "C", "rel"))
        out = run_case.run_lp(binaries["lpsim_asan"], H, h, C, relax=rel, env=SAN_ENV)
        assert_clean(out)
        assert np.all(out["status"] == 0), out["status"]
        assert out["iters"].max() <= 40
        ref = _highs(H, h, C, rel)
        assert np.max(np.abs(out["val"] - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-9
        for x, c, r, v in zip(out["x"], C, rel, out["val"]):
            hk = h + (np.arange(len(h)) == r) * 1.0
            assert np.max((H @ x - hk) / np.maximum(np.abs(hk), 1.0)) <= 1e-10 and abs(c @ x - v) <= 1e-9 * max(1.0, abs(v))


def test_lp_kernel_source_under_msan(binaries):
    """Random polytopes (d = 3, 6), a box with duplicated and nearly parallel rows and objectives along its rows (whole
    facets optimal), an unbounded direction and an empty set: the verdicts, the values against HiGHS, and no read of an
    uninitialised word (the per-wave work space starts out poisoned)."""
    rng = np.random.default_rng(4)
    for d, nr in ((3, 70), (6, 130)):
        H = rng.standard_normal((nr, d))
        h = 1.0 + rng.random(nr)
        C = rng.standard_normal((6, d))
        out = run_case.run_lp(binaries["lpsim_msan"], H, h, C, env=SAN_ENV)
        assert_clean(out)
        assert np.all(out["status"] == 0)
        ref = _highs(H, h, C, None)
        assert np.max(np.abs(out["val"] - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-9
    d = 5
    box = np.r_[np.eye(d), -np.eye(d)]
    tilt = box[:4] + 1e-7 * rng.standard_normal((4, d))
    H = np.r_[box, box[:3], tilt]
    h = np.r_[np.ones(2 * d), np.ones(3), np.ones(4) + 1e-9]
    C = np.r_[box[:4], tilt[:2], np.ones((1, d))]
    rel = np.array([-1, -1, 0, 1, -1, 10, -1], dtype=np.int32)
    out = run_case.run_lp(binaries["lpsim_msan"], H, h, C, relax=rel, env=SAN_ENV)
    assert_clean(out)
    assert np.all(out["status"] == 0), out["status"]
    ref = _highs(H, h, C, rel)
    assert np.max(np.abs(out["val"] - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-8
    # a half space is unbounded along its normal's orthogonal complement; x <= -1, -x <= -1 is empty
    out = run_case.run_lp(binaries["lpsim_msan"], np.array([[1.0, 0.0], [0.0, 1.0]]), np.ones(2), np.array([[-1.0, 0.0], [1.0, 1.0]]), env=SAN_ENV)
    assert_clean(out)
    assert out["status"][0] == 4 and np.isinf(out["val"][0]) and out["status"][1] == 0 and abs(out["val"][1] - 2.0) <= 1e-12
    out = run_case.run_lp(binaries["lpsim_msan"], np.array([[1.0], [-1.0]]), np.array([-1.0, -1.0]), np.array([[1.0]]), env=SAN_ENV)
    assert_clean(out)
    assert out["status"][0] == 2
