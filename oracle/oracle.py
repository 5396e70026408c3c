"""ORACLE (test infrastructure): ctypes binding of oracle/libtmpc_oracle.so.

Mirrors the product binding's call shape so that the parity tests read
`hip.solve(...)` vs `oracle.solve(...)` on the same inputs.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_PTR_FIELDS = ["A", "B", "Q", "R", "P", "T", "K", "K_anc",
               "Hx", "hx", "Hu", "hu", "HT", "hT", "HZ", "hZ", "HZW", "hZW", "HTP", "hTP"]


class Problem(C.Structure):
    """Field-for-field include/tmpc.h: tmpc_problem."""
    _fields_ = ([(n, C.c_int32) for n in ("nx", "nu", "N", "rx", "ru", "rT", "rZ", "rZW",
                                          "fixed_x0", "extended", "literal_terminal_row", "max_iter")]
                + [("tol", C.c_double)]
                + [(n, C.POINTER(C.c_double)) for n in _PTR_FIELDS] + [("rTP", C.c_int32), ("terminal_equality", C.c_int32)])


def pack_problem(d: dict):
    """dict from TubeTrackingMPC._problem_dict() -> (Problem, keepalive list)."""
    p = Problem()
    keep = []
    nx, nu = int(d["nx"]), int(d["nu"])
    p.nx, p.nu, p.N = nx, nu, int(d["N"])
    p.fixed_x0 = int(d.get("fixed_x0", 0))
    p.extended = int(d.get("extended", 0))
    p.literal_terminal_row = int(d.get("literal_terminal_row", 1))
    p.max_iter = int(d.get("max_iter", 0))
    p.tol = float(d.get("tol", 0.0))
    shapes = {"A": (nx, nx), "B": (nx, nu), "Q": (nx, nx), "R": (nu, nu), "P": (nx, nx), "T": (nx, nx),
              "K": (nu, nx), "K_anc": (nu, nx)}
    for name in _PTR_FIELDS:
        v = d.get(name)
        if v is None:
            setattr(p, name, C.POINTER(C.c_double)())
            continue
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
        if name in shapes:
            a = np.ascontiguousarray(a.reshape(shapes[name]))
        keep.append(a)
        setattr(p, name, a.ctypes.data_as(C.POINTER(C.c_double)))

    def rows(key, width):
        v = d.get(key)
        if v is None:
            return 0
        a = np.asarray(v)
        if a.ndim != 2 or a.shape[1] != width:
            raise ValueError(f"{key} must have {width} columns, got shape {a.shape}")
        return a.shape[0]

    p.rx, p.ru = rows("Hx", nx), rows("Hu", nu)
    p.rT = rows("HT", 2 * nx + nu)
    p.rZ, p.rZW = rows("HZ", nx), rows("HZW", nx)
    p.rTP = rows("HTP", nx + nu)
    p.terminal_equality = int(d.get("terminal_equality", 0))
    for hk, Hk in (("hx", "Hx"), ("hu", "Hu"), ("hT", "HT"), ("hZ", "HZ"), ("hZW", "HZW"), ("hTP", "HTP")):
        if d.get(Hk) is not None and np.asarray(d[hk]).size != np.asarray(d[Hk]).shape[0]:
            raise ValueError(f"{hk} / {Hk} row mismatch")
    return p, keep


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libtmpc_oracle.so")
    src = os.path.join(_HERE, "tmpc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libtmpc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.oracle_create.argtypes = [C.POINTER(Problem), C.POINTER(C.c_void_p)]
        L.oracle_create.restype = C.c_int
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_destroy.restype = None
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_get_dims.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.oracle_solve_batch.argtypes = [C.c_void_p, C.c_int64, dp, dp, up, dp, dp, dp, dp, ip, ip, C.c_int]
        L.oracle_solve_batch.restype = C.c_int
        L.oracle_get_reduced.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp]
        _LIB = L
    return _LIB


class Oracle:
    def __init__(self, problem: dict):
        L = lib()
        p, self._keep = pack_problem(problem)
        h = C.c_void_p()
        rc = L.oracle_create(C.byref(p), C.byref(h))
        if rc != 0:
            raise RuntimeError(L.oracle_last_error().decode())
        self._h = h
        self.nx, self.nu, self.N = p.nx, p.nu, p.N

    def dims(self, variant: int = 0):
        nv, nc, npar = C.c_int32(), C.c_int32(), C.c_int32()
        lib().oracle_get_dims(self._h, variant, C.byref(nv), C.byref(nc), C.byref(npar))
        return nv.value, nc.value, npar.value

    def solve(self, x_k, ref, variant=None, nthreads: int = 0):
        nx, nu, N = self.nx, self.nu, self.N
        x = np.ascontiguousarray(np.asarray(x_k, dtype=np.float64).reshape(-1, nx))
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(ref, dtype=np.float64).reshape(-1, nx), x.shape))
        B = x.shape[0]
        out = dict(u_nom=np.empty((B, N, nu)), x_nom0=np.empty((B, nx)), xu_ss=np.empty((B, nx + nu)),
                   x_nom=np.empty((B, N + 1, nx)), status=np.empty(B, np.int32), iters=np.empty(B, np.int32))
        dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        vp = up()
        if variant is not None:
            var = np.ascontiguousarray(np.broadcast_to(np.asarray(variant, dtype=np.uint8).reshape(-1), (B,)))
            vp = var.ctypes.data_as(up)
        rc = lib().oracle_solve_batch(self._h, B, x.ctypes.data_as(dp), r.ctypes.data_as(dp), vp,
                                      out["u_nom"].ctypes.data_as(dp), out["x_nom0"].ctypes.data_as(dp),
                                      out["xu_ss"].ctypes.data_as(dp), out["x_nom"].ctypes.data_as(dp),
                                      out["status"].ctypes.data_as(ip), out["iters"].ctypes.data_as(ip), int(nthreads))
        if rc != 0:
            raise RuntimeError(f"oracle_solve_batch failed ({rc})")
        out["x_ss"] = out["xu_ss"][:, :nx]
        out["u_ss"] = out["xu_ss"][:, nx:]
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
