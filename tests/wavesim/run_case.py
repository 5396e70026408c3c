"""TEST INFRASTRUCTURE: has the product library dump the wave kernel's input layout for a problem (host-only handle,
tmpc_debug_dump_layout), writes the batch, runs a wavesim binary on both, reads its output
(tests/wavesim/wavesim_main.cpp has the formats)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")


def build_all(extra="-DTMPC_SIM_SHAPES"):
    """every binary of the Makefile in one parallel make (what __graft_entry__.build() runs as well); up to date -> no-op"""
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "all", f"EXTRA={extra}"])
    return {t: os.path.join(BUILD, t) for t in ("wavesim", "wavesim_asan", "wavesim_msan", "wavesim_ext_asan", "blocksim", "blocksim_asan", "blocksim_msan",
                                                 "lpsim_asan", "lpsim_msan")}


def run(binary, d, X, R, variant=None, env=None, timeout=1800, block=False):
    """block: the workgroup-per-QP kernel (blocksim binaries, variant 0 only)"""
    from LinearMPCOverNetworks import _native
    L = _native.lib()
    dump = L.tmpc_debug_dump_block_layout if block else L.tmpc_debug_dump_layout
    dump.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    h = _native.create(d, -1)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            lays = []
            for k in range(2 if variant is not None else 1):
                path = os.path.join(tmp, f"layout{k}.bin")
                rc = dump(h.ptr, k, path.encode())
                if rc != 0:
                    raise RuntimeError(f"tmpc_debug_dump_layout({k}) failed: {rc}")
                lays.append(path)
            X = np.ascontiguousarray(X, dtype=np.float64)
            R = np.ascontiguousarray(R, dtype=np.float64)
            B, nx = X.shape
            batch, out = os.path.join(tmp, "batch.bin"), os.path.join(tmp, "out.bin")
            with open(batch, "wb") as f:
                np.array([B, nx], dtype=np.int64).tofile(f)
                X.tofile(f)
                R.tofile(f)
                np.array([0 if variant is None else 1], dtype=np.int64).tofile(f)
                if variant is not None:
                    np.ascontiguousarray(np.broadcast_to(np.asarray(variant, dtype=np.uint8), (B,))).tofile(f)
            res = subprocess.run([binary] + lays + [batch, out], capture_output=True, text=True, timeout=timeout,
                                 env=dict(os.environ, **(env or {})))
            if res.returncode != 0:
                raise RuntimeError(f"{os.path.basename(binary)} failed ({res.returncode}):\n{res.stderr[-8000:]}")
            nu, N = h.nu, h.N
            raw = open(out, "rb").read()
            off = 0

            def take(n, dt):
                nonlocal off
                a = np.frombuffer(raw, dtype=dt, count=n, offset=off)
                off += a.nbytes
                return a
            o = dict(u_nom=take(B * N * nu, np.float64).reshape(B, N, nu), x_nom0=take(B * nx, np.float64).reshape(B, nx),
                     xu_ss=take(B * (nx + nu), np.float64).reshape(B, nx + nu), status=take(B, np.int32), iters=take(B, np.int32))
            o["rendezvous"] = int(take(1, np.uint64)[0])
            o["stderr"] = res.stderr
            o["stdout"] = res.stdout
            return o
    finally:
        _native.destroy(h)


def run_lp(binary, H, h, Cmat, relax=None, relax_by=1.0, env=None, timeout=1800):
    """the batched LP kernel (lpsim binaries) on the layout tmpc_debug_dump_lp_layout writes for the polytope (H, h)"""
    from LinearMPCOverNetworks import _native
    L = _native.lib()
    L.tmpc_debug_dump_lp_layout.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_char_p]
    L.tmpc_debug_dump_lp_layout.restype = C.c_int
    H = np.ascontiguousarray(H, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64).reshape(-1)
    Cm = np.ascontiguousarray(np.atleast_2d(Cmat), dtype=np.float64)
    nr, d = H.shape
    B = Cm.shape[0]
    with tempfile.TemporaryDirectory() as tmp:
        lay, batch, out = (os.path.join(tmp, n) for n in ("layout.bin", "batch.bin", "out.bin"))
        rc = L.tmpc_debug_dump_lp_layout(d, nr, H.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), float(relax_by), lay.encode())
        if rc != 0:
            raise RuntimeError(f"tmpc_debug_dump_lp_layout failed ({rc}): {L.tmpc_last_error(None).decode()}")
        with open(batch, "wb") as f:
            np.array([B, 0 if relax is None else 1], dtype=np.int64).tofile(f)
            Cm.tofile(f)
            if relax is not None:
                np.ascontiguousarray(relax, dtype=np.int32).reshape(B).tofile(f)
        res = subprocess.run([binary, lay, batch, out], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
        if res.returncode != 0:
            raise RuntimeError(f"{os.path.basename(binary)} failed ({res.returncode}):\n{res.stderr[-8000:]}")
        raw = open(out, "rb").read()
    val = np.frombuffer(raw, np.float64, B, 0)
    x = np.frombuffer(raw, np.float64, B * d, 8 * B).reshape(B, d)
    st = np.frombuffer(raw, np.int32, B, 8 * B * (d + 1))
    it = np.frombuffer(raw, np.int32, B, 8 * B * (d + 1) + 4 * B)
    return dict(val=val, x=x, status=st, iters=it, stderr=res.stderr, stdout=res.stdout)


def run_loop(binary, d, K, K_anc, Z, p_loss, ref, th_u, ga_u, w, x0=None, smart=False, warm=False, env=None, timeout=3600, extended=False):
    """the closed loop with the state machines inside the solve kernels (wavesim --loop) for the controller of problem dict `d`: plain ->
    closed_loop_kernel, a wave keeps its trajectory for all T steps; extended (d carries the packet-received problem) -> one
    closed_loop_step_kernel launch per problem and time step.  Gains K, K_anc (nu, nx), tube cross-section Z (object with .A, .b) or None"""
    from LinearMPCOverNetworks import _native
    L = _native.lib()
    L.tmpc_debug_dump_layout.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    h = _native.create(d, -1)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            loop, out = (os.path.join(tmp, n) for n in ("loop.bin", "out.bin"))
            lays = []
            for k in range(2 if extended else 1):
                lays.append(os.path.join(tmp, f"layout{k}.bin"))
                rc = L.tmpc_debug_dump_layout(h.ptr, k, lays[-1].encode())
                if rc != 0:
                    raise RuntimeError(f"tmpc_debug_dump_layout({k}) failed: {rc}")
            c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
            th_u, ga_u, w = c(th_u), c(ga_u), c(w)
            B, T = th_u.shape
            nx, nu = h.nx, h.nu
            HZ = np.zeros((0, nx)) if Z is None else c(Z.A)
            hZ = np.zeros(0) if Z is None else c(Z.b)
            with open(loop, "wb") as f:
                np.array([B, T, nx, nu, HZ.shape[0], int(bool(extended)), int(bool(smart)), int(bool(warm))], dtype=np.int64).tofile(f)
                for a in (np.asarray(d["A"]), np.asarray(d["B"]), K, K_anc, HZ, hZ, p_loss, ref, th_u, ga_u, w,
                          np.zeros((B, nx)) if x0 is None else x0):
                    c(a).tofile(f)
            res = subprocess.run([binary, "--loop"] + lays + [loop, out], capture_output=True, text=True, timeout=timeout,
                                 env=dict(os.environ, **(env or {})))
            if res.returncode != 0:
                raise RuntimeError(f"{os.path.basename(binary)} --loop failed ({res.returncode}):\n{res.stderr[-8000:]}")
            raw = open(out, "rb").read()
    finally:
        _native.destroy(h)
    o = dict(err2=np.frombuffer(raw, np.float64, B, 0), x_final=np.frombuffer(raw, np.float64, B * nx, 8 * B).reshape(B, nx),
             consistent=np.frombuffer(raw, np.float64, B, 8 * B * (1 + nx)))
    off = 8 * B * (2 + nx)
    for k in ("tube_violations", "not_optimal", "iters_sum"):
        o[k] = np.frombuffer(raw, np.int32, B, off)
        off += 4 * B
    o["tracking_error"] = np.sqrt(o["err2"]) / T
    o["stderr"], o["stdout"] = res.stderr, res.stdout
    return o
