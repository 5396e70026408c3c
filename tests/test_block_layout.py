"""Layout of the workgroup-per-QP kernel (csrc/tmpc_device.hpp: BlockQP): rows of G as functionals.

CPU part: the layout a host-only handle keeps (tmpc_debug_dump_block_layout -- the very arrays tests/wavesim feeds to the
kernel source) against the condensed QP of the same handle (tmpc_get_condensed): every constraint row, its right-hand side
and its dependence on x_k must be found on exactly one row side of the layout.
GPU part: the kernel on the paired layout against the kernel with a row of G per constraint row (TMPC_BLOCK_PAIRS=0)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import common
from LinearMPCOverNetworks import _native

_BQ_INTS = ("ncp", "nz4", "zx0", "znx", "mir", "ng", "ngp")
_ARRAYS = ("Hs", "Hinv", "F1s", "F2s", "gp0", "Ep", "Dv", "Tzs", "Txf", "Mth", "A", "B", "Grm", "Gcm", "GHrm", "g0", "Es", "ncols", "Gw", "ci")


def _layout(h, tmp_path):
    path = str(tmp_path / "block_layout.bin")
    L = _native.lib()
    L.tmpc_debug_dump_block_layout.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    assert L.tmpc_debug_dump_block_layout(h.ptr, 0, path.encode()) == 0
    raw = open(path, "rb").read()
    tag, fmt = struct.unpack_from("ii", raw, 0)
    assert tag == 0x43504D54 and fmt == 3, (hex(tag), fmt)      # "TMPC", tmpc::DUMP_FORMAT (csrc/tmpc_device.hpp): the record list below
    tiles, ws_rows = struct.unpack_from("ii", raw, 8)
    sz_d, sz_bq = struct.unpack_from("QQ", raw, 16)
    off = 32 + sz_d
    bq = dict(zip(_BQ_INTS, struct.unpack_from("7i", raw, off)))
    off += sz_bq
    arrs = {}
    for name in _ARRAYS:
        (n,) = struct.unpack_from("Q", raw, off)
        off += 8
        arrs[name] = np.frombuffer(raw, dtype=np.int32 if name == "ncols" else np.float64, count=n // (4 if name == "ncols" else 8), offset=off).copy() if n else None
        off += n
    assert off == len(raw)
    return tiles, bq, arrs


@pytest.mark.parametrize("name,N,fixed", [("synthetic", 30, True), ("cartpole", 20, True), ("double_integrator", 10, False)])
def test_every_constraint_row_sits_on_one_side_of_a_functional(name, N, fixed, hip_lib, tmp_path):
    mpc, _ = common.make_mpc(name, N, fixed)
    h = _native.create(mpc._problem_dict(), -1)
    try:
        nv, nc, _ = _native.get_dims(h, 0)
        cond = _native.get_condensed(h, 0)
        tiles, bq, a = _layout(h, tmp_path)
    finally:
        _native.destroy(h)
    NVP, nx = 16 * tiles, mpc._nx
    ncp, mir, ng, ngp = bq["ncp"], bq["mir"], bq["ng"], bq["ngp"]
    assert mir == ngp and ncp == 2 * ngp and 2 * ng == nc and ngp % 64 == 0 and bq["nz4"] == 0         # all rows paired
    Grm = a["Grm"].reshape(ngp + NVP, NVP)
    assert np.array_equal(Grm[ngp:], a["Hs"].reshape(NVP, NVP))          # the rows of Hs ride along in the G'v pass
    Grm = Grm[:ngp]
    assert np.array_equal(a["Gcm"].reshape(NVP, ngp), Grm.T)
    Gw, GH = a["Gw"].reshape(ncp, NVP), a["GHrm"].reshape(ncp, NVP)
    assert np.array_equal(Gw[:ngp], Grm) and np.array_equal(Gw[mir:], -Grm) and np.array_equal(GH[mir:], -GH[:ngp])
    assert np.all(Grm[ng:] == 0.0) and np.all(Grm[:, nv:] == 0.0)
    Hinv = a["Hinv"].reshape(NVP, NVP)
    assert np.abs(GH[:ng] - Grm[:ng] @ Hinv).max() < 1e-10 * max(1.0, np.abs(GH).max())
    # staircase bookkeeping: a row is zero beyond the columns it is said to reach
    for r in range(ng):
        assert np.all(Grm[r, a["ncols"][r]:] == 0.0)
    # the scaled rows of the condensed QP (Jacobi scaling Dv of z, unit rows), each with its right-hand side data
    Dv = a["Dv"][:nv]
    Gs = cond["G"] * Dv[None, :]
    rho = np.linalg.norm(Gs, axis=1)
    want = np.c_[Gs / rho[:, None], cond["g0"] / rho, cond["E"] / rho[:, None]]
    sides = np.r_[np.arange(ng), mir + np.arange(ng)]
    have = np.c_[Gw[sides, :nv], a["g0"][sides], a["Es"].reshape(ncp, nx)[sides]]
    assert have.shape == want.shape
    # one-to-one up to the 1e-13 by which a mirror row may differ from its negated partner
    key = lambda M: [tuple(np.round(row, 9) + 0.0) for row in M]
    assert sorted(key(have)) == sorted(key(want))
    pad = np.setdiff1d(np.arange(ncp), sides)
    assert np.all(a["g0"][pad] == 1.0) and np.all(a["Es"].reshape(ncp, nx)[pad] == 0.0)


def test_developer_knob_keeps_a_row_of_g_per_constraint_row(hip_lib, tmp_path, monkeypatch):
    mpc, _ = common.make_mpc("cartpole", 20, True)
    monkeypatch.setenv("TMPC_BLOCK_PAIRS", "0")
    h = _native.create(mpc._problem_dict(), -1)
    try:
        nv, nc, _ = _native.get_dims(h, 0)
        tiles, bq, a = _layout(h, tmp_path)
    finally:
        _native.destroy(h)
    assert bq["mir"] == 0 and bq["ng"] == nc and bq["ngp"] == bq["ncp"] and a["Gw"] is None


@pytest.mark.gpu
def test_paired_and_unpaired_layouts_give_the_same_answers(hip_lib, monkeypatch):
    """Block kernel, cartpole N = 20 (604 rows = 302 functionals) on closed-loop states and config 5 (1236 rows = 618 functionals):
    the two layouts are the same algorithm on the same numbers up to the order of summation."""
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))[::3]
    rng = np.random.default_rng(7)
    for name, N, X, R in (("cartpole", 20, S[:, :4], S[:, 4:]), ("synthetic", 30, None, None)):
        outs = []
        for pairs in ("1", "0"):
            monkeypatch.setenv("TMPC_BLOCK_PAIRS", pairs)
            mpc, w = common.make_mpc(name, N, True, create=True)
            if X is None:
                X = rng.uniform(-0.5, 0.5, (96, 12)) * mpc._Xc.b[:12]
                R = np.zeros((96, 12)); R[:, 0] = rng.uniform(-2, 2, 96)
            mpc.set_kernel_path("block")
            outs.append(mpc._solve(X, R))
        a, b = outs
        assert np.array_equal(a["status"], b["status"]) and (a["status"] == 0).mean() > 0.9
        assert np.abs(a["iters"].astype(int) - b["iters"]).max() <= 1
        good = a["status"] == 0
        assert (a["iters"][good] > 0).sum() > 20
        scale = max(1.0, np.abs(a["u_nom"][good]).max())
        assert np.abs(a["u_nom"][good] - b["u_nom"][good]).max() <= 1e-9 * scale


@pytest.mark.gpu
def test_work_counter_with_two_variants_on_the_block_path(hip_lib):
    """More instances than resident workgroups, both problems of the extended controller in one call, every instance through the
    workgroup-per-QP kernel: each variant's launch draws the whole batch from its own counter word and skips the other
    variant's instances.  Against the wave kernels on the same batch."""
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
    rng = np.random.default_rng(11)
    idx = rng.integers(0, len(S), 3072)
    mpc, w = common.make_mpc("cartpole", 20, True, extended=True, create=True)
    X = S[idx, :4] + rng.uniform(-1, 1, (len(idx), 4)) * w["w_bound"] * 2.0
    R = S[idx, 4:]
    gam = (rng.uniform(size=len(idx)) < 0.6).astype(np.uint8)
    ref = mpc._solve(X, R, variant=gam)
    assert mpc.get_kernel_path(0) == "wave" and mpc.get_kernel_path(1) == "wave"
    mpc.set_kernel_path("block")
    try:
        assert mpc.get_kernel_path(0) == "block" and mpc.get_kernel_path(1) == "block"
        out = mpc._solve(X, R, variant=gam)
        again = mpc._solve(X, R, variant=gam)
    finally:
        mpc.set_kernel_path("auto")
    assert np.array_equal(out["status"], again["status"]) and np.array_equal(out["u_nom"], again["u_nom"], equal_nan=True)
    both = (ref["status"] == 0) & (out["status"] == 0)
    assert both.sum() > 2000 and (gam[both] == 1).sum() > 800 and (gam[both] == 0).sum() > 800
    # infeasibility verdicts agree; the few instances one path certifies and the other returns uncertified may differ in status 0 / 1
    assert np.array_equal(ref["status"] == 2, out["status"] == 2)
    assert np.abs(out["u_nom"][both, 0] - ref["u_nom"][both, 0]).max() <= 1e-8
