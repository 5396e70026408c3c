"""Product library, host side only (no GPU): symbol table, argument checking, and the
C++ condensing of tmpc_create checked against the un-condensed QP.  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common
from oracle import ipm_numpy, qp_sparse
from oracle.oracle import Oracle

S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))


def test_library_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(common.ROOT, "include", "tmpc.h")).read()
    declared = set(re.findall(r"\b(tmpc_[a-z_]+)\s*\(", hdr))
    assert {"tmpc_create", "tmpc_destroy", "tmpc_solve_batch", "tmpc_solve_batch_device", "tmpc_synchronize",
            "tmpc_last_kernel_ms", "tmpc_kernel_ms_total", "tmpc_get_dims", "tmpc_get_condensed", "tmpc_last_error",
            "tmpc_abi_version", "tmpc_set_kernel_path", "tmpc_get_kernel_path"} <= declared
    L = hip_lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/tmpc.h but not exported"
    assert L.tmpc_abi_version() == 5


def test_struct_layout_matches_header(hip_lib):
    """ctypes mirror == C struct: 12 int32, one double, 20 pointers, one int32 (+ tail padding)."""
    assert C.sizeof(hip_lib.TmpcProblem) == 12 * 4 + 8 + 20 * 8 + 8


def test_argument_errors_are_reported_not_thrown(hip_lib):
    mpc, _ = common.make_mpc("cartpole", 10, True)
    p = mpc._problem_dict()
    bad = dict(p)
    bad["HT"] = None
    bad["hT"] = None
    bad["Hx"] = p["Hx"][:, :3]
    with pytest.raises(ValueError):
        hip_lib.create(bad, device=-1)
    bad = dict(p)
    bad["fixed_x0"] = 0                      # free x_0 but no Z given
    with pytest.raises(RuntimeError, match="initial state"):
        hip_lib.create(bad, device=-1)
    bad = dict(p)
    bad["B"] = np.zeros_like(p["B"])         # [A-I, B] loses rank: steady states not parametrised
    with pytest.raises(RuntimeError, match="rank"):
        hip_lib.create(bad, device=-1)


def test_host_only_handle_refuses_to_solve(hip_lib):
    """No CPU solve path exists in the product: device < 0 can only be inspected."""
    mpc, _ = common.make_mpc("cartpole", 10, True)
    h = hip_lib.create(mpc._problem_dict(), device=-1)
    with pytest.raises(RuntimeError, match="without the GPU"):
        hip_lib.solve_batch(h, S[:2, :4].copy(), S[:2, 4:].copy())
    hip_lib.destroy(h)


@pytest.mark.parametrize("name,N,fixed,extended", [("cartpole", 10, True, False), ("double_integrator", 5, False, False),
                                                   ("double_integrator", 10, False, False), ("double_integrator", 5, True, True)])
def test_condensed_qp_has_the_same_minimiser_as_the_sparse_form(hip_lib, oracle_lib, name, N, fixed, extended):
    mpc, w = common.make_mpc(name, N, fixed, extended=extended)
    p = mpc._problem_dict()
    h = hip_lib.create(p, device=-1)
    orc = Oracle(p)
    rng = np.random.default_rng(1)
    nx = p["nx"]
    for variant in range(2 if extended else 1):
        nv, nc, npar = hip_lib.get_dims(h, variant)
        assert (nv, nc, npar) == orc.dims(variant)
        c = hip_lib.get_condensed(h, variant)
        assert np.allclose(c["H"], c["H"].T) and np.all(np.linalg.eigvalsh(c["H"]) > 0)
        if name == "cartpole":
            pts = [(S[i, :4], S[i, 4:]) for i in (0, 77, 200, 333, 555)]
        else:
            pts = [(rng.uniform(-1, 1, nx) * [3.0, 0.5], np.array([rng.uniform(-9, 9), 0.0])) for _ in range(5)]
        for x, r in pts:
            q = c["F1"] @ x + c["F2"] @ r
            hh = c["g0"] + c["E"] @ x
            sol = orc.solve(x[None], r[None], variant=np.array([variant], dtype=np.uint8))
            assert sol["status"][0] == 0
            u = sol["u_nom"][0].reshape(-1)
            # the oracle's inputs must be optimal for the condensed problem: feasibility + no better point
            res = ipm_numpy.solve_qp(c["H"], q, None, None, c["G"], hh, tol=1e-11, max_iter=300)
            z = res["v"]
            f = lambda zz: 0.5 * zz @ c["H"] @ zz + q @ zz
            assert np.max(c["G"] @ z - hh) < 1e-7
            np.testing.assert_allclose(z[:N * p["nu"]], u, atol=5e-3)     # flat directions, see test_oracle
            # sharp statement: swapping in the oracle's inputs does not change the optimal value
            z2 = z.copy()
            z2[:N * p["nu"]] = u
            assert abs(f(z2) - f(z)) <= 1e-6 * max(1.0, abs(f(z)))
    hip_lib.destroy(h)


def test_condensed_objective_and_constraints_equal_sparse_ones(hip_lib):
    """Exact algebraic check for the fixed-x0 cartpole: for arbitrary inputs u and steady-state
    parameter, cost(z) - cost(0) and G z - h agree with the un-condensed expressions evaluated on
    the trajectory that z generates."""
    mpc, w = common.make_mpc("cartpole", 10, True)
    p = mpc._problem_dict()
    h = hip_lib.create(p, device=-1)
    c = hip_lib.get_condensed(h, 0)
    A, B, N, nx, nu = p["A"], p["B"], p["N"], p["nx"], p["nu"]
    rng = np.random.default_rng(3)
    # steady-state direction: null([A-I, B]) is one-dimensional for the cartpole (position offset)
    Sm = np.c_[A - np.eye(nx), B]
    ns = np.linalg.svd(Sm)[2][-1]
    x = S[50, :4]
    r = S[50, 4:]
    qp = qp_sparse.build_sparse_qp(p, x, r)

    def traj(u, th):
        xs = [x]
        for i in range(N):
            xs.append(A @ xs[-1] + B @ u[i])
        xb = ns * th
        return qp_sparse.pack(qp, np.array(xs), u, xb[:nx], xb[nx:])

    f = lambda v: 0.5 * v @ qp["P"] @ v + qp["q"] @ v
    v0 = traj(np.zeros((N, nu)), 0.0)
    # theta in the library's basis: recover its scale from the constraint matrix via one probe
    u1 = rng.normal(size=(N, nu))
    for th_lib in (0.0, 0.7, -1.3):
        # library basis vector = +-ns (orthonormal null vector): resolve the sign by matching the cost
        z = np.r_[u1.reshape(-1), th_lib]
        fz = 0.5 * z @ c["H"] @ z + (c["F1"] @ x + c["F2"] @ r) @ z
        cands = [f(traj(u1, s * th_lib)) - f(v0) for s in (1.0, -1.0)]
        assert min(abs(fz - cands[0]), abs(fz - cands[1])) <= 1e-9 * max(1.0, abs(fz))
    hip_lib.destroy(h)


def test_closed_loop_setters_validate_their_arguments(hip_lib):
    """tmpc_mc_set_plant / tmpc_mc_set_actuator / tmpc_set_kernel_path on a host-only handle: argument errors come back as
    codes + message; tmpc_mc_run itself refuses without a device."""
    import ctypes as C
    mpc, _ = common.make_mpc("cartpole", 10, True)
    h = hip_lib.create(mpc._problem_dict(), device=-1)
    L = hip_lib.lib()
    assert L.tmpc_mc_set_actuator(h.ptr, 7) == -1 and b"actuator" in L.tmpc_last_error(h.ptr)
    assert L.tmpc_mc_set_actuator(h.ptr, 1) == 0 and L.tmpc_mc_set_actuator(h.ptr, 0) == 0
    assert L.tmpc_mc_set_plant(h.ptr, 1, None, 10) == -1                      # cart-pole plant without parameters
    par = (C.c_double * 7)(1.0, 0.1, 0.0, 0.001, 9.8, 0.5, 0.02)
    assert L.tmpc_mc_set_plant(h.ptr, 1, par, 0) == -1                         # substeps >= 1
    assert L.tmpc_mc_set_plant(h.ptr, 1, par, 10) == 0 and L.tmpc_mc_set_plant(h.ptr, 0, None, 0) == 0
    assert L.tmpc_set_kernel_path(h.ptr, 9) == -1
    one = np.zeros((1, 1))
    rc = L.tmpc_mc_run(h.ptr, 1, 1, 0, one.ctypes.data, one.ctypes.data, one.ctypes.data, one.ctypes.data, np.zeros((1, 1, 4)).ctypes.data,
                       None, None, None, 0, None, None, None, None, None, None)
    assert rc == -3 and b"GPU" in L.tmpc_last_error(h.ptr)                      # TMPC_E_DEVICE: no CPU path
    assert L.tmpc_mc_set_warm_start(h.ptr, 1) == 0 and L.tmpc_mc_set_warm_start(h.ptr, 0) == 0
    assert L.tmpc_mc_set_capture(h.ptr, 3) == 0 and L.tmpc_mc_set_capture(h.ptr, -1) == 0
    assert L.tmpc_mc_get_capture(h.ptr, 5, None, None, None) == -1 and b"recorded" in L.tmpc_last_error(h.ptr)
    # per-solve times, physics-rate error, device generator: settings are accepted, results need a run
    assert L.tmpc_set_solve_timing(h.ptr, 1) == 0
    ticks = np.zeros(4, dtype=np.int64)
    assert L.tmpc_get_solve_ticks(h.ptr, 4, ticks.ctypes.data) == -1 and b"timed" in L.tmpc_last_error(h.ptr)
    assert L.tmpc_mc_get_solve_ticks(h.ptr, 4, ticks.ctypes.data, None) == -1
    assert L.tmpc_set_solve_timing(h.ptr, 0) == 0 and L.tmpc_set_solve_timing(None, 1) == -1
    assert L.tmpc_mc_get_physics_error(h.ptr, 4, np.zeros(4).ctypes.data) == -1 and b"linear plant" in L.tmpc_last_error(h.ptr)
    wb = np.array([1e-4, 2.7e-3, 3e-4, 4.3e-2])
    assert L.tmpc_mc_set_device_rng(h.ptr, 1, 12345, 7, wb.ctypes.data) == 0 and L.tmpc_mc_set_device_rng(h.ptr, 0, 0, 0, None) == 0
    # with the generator on, tmpc_mc_run takes NULL realisations (and still has no CPU path)
    assert L.tmpc_mc_set_device_rng(h.ptr, 1, 1, 0, None) == 0
    rc = L.tmpc_mc_run(h.ptr, 1, 1, 0, one.ctypes.data, one.ctypes.data, None, None, None, None, None, None, 0, None, None, None, None, None, None)
    assert rc == -3
    hip_lib.destroy(h)
    # double integrator: the cart-pole plant needs nx = 4, nu = 1
    mpc2, _ = common.make_mpc("double_integrator", 5, False)
    h2 = hip_lib.create(mpc2._problem_dict(), device=-1)
    assert L.tmpc_mc_set_plant(h2.ptr, 1, par, 10) == -1 and b"nx = 4" in L.tmpc_last_error(h2.ptr)
    hip_lib.destroy(h2)
