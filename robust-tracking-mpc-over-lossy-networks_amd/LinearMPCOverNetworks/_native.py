"""ctypes binding of lib/libtmpc_hip.so (C ABI: include/tmpc.h).

This is the only route from the Python classes to the solver: there is no CPU
fall-back.  A missing library raises at import of the first solve/setup call.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (TMPC_LIB: another build of the same library, e.g. a diagnostic variant -- developers' A/B runs; default: the in-tree build)
LIB_PATH = os.environ.get("TMPC_LIB") or os.path.join(_PKG, "lib", "libtmpc_hip.so")
ABI_VERSION = 5

_PTR_FIELDS = ["A", "B", "Q", "R", "P", "T", "K", "K_anc",
               "Hx", "hx", "Hu", "hu", "HT", "hT", "HZ", "hZ", "HZW", "hZW", "HTP", "hTP"]
_INT_FIELDS = ["nx", "nu", "N", "rx", "ru", "rT", "rZ", "rZW",
               "fixed_x0", "extended", "literal_terminal_row", "max_iter"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint8)


class TmpcProblem(C.Structure):
    """Field-for-field include/tmpc.h: tmpc_problem."""
    _fields_ = ([(n, C.c_int32) for n in _INT_FIELDS] + [("tol", C.c_double)]
                + [(n, _dp) for n in _PTR_FIELDS] + [("rTP", C.c_int32), ("terminal_equality", C.c_int32)])


_lib = None


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so with the same SONAME as the system one.
    Two HIP runtimes cannot both own the GPU in one process, and whichever is loaded first
    wins the SONAME.  When torch is installed (bench.py and the multi-GPU driver use it for
    device buffers and RCCL) bind this library to torch's copy, so that tensors, RCCL and the
    solve kernels share one runtime regardless of import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_PKG, 'csrc')}` "
                "(or __graft_entry__.build()).  There is no CPU solve path.")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.tmpc_abi_version.restype = C.c_int
        if L.tmpc_abi_version() != ABI_VERSION:
            raise RuntimeError("libtmpc_hip.so ABI version mismatch")
        L.tmpc_last_error.argtypes = [C.c_void_p]
        L.tmpc_last_error.restype = C.c_char_p
        L.tmpc_create.argtypes = [C.POINTER(TmpcProblem), C.c_int, C.POINTER(C.c_void_p)]
        L.tmpc_create.restype = C.c_int
        L.tmpc_destroy.argtypes = [C.c_void_p]
        L.tmpc_destroy.restype = None
        sig = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tmpc_solve_batch.argtypes = sig
        L.tmpc_solve_batch.restype = C.c_int
        L.tmpc_solve_batch_device.argtypes = sig
        L.tmpc_solve_batch_device.restype = C.c_int
        L.tmpc_set_kernel_path.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_set_kernel_path.restype = C.c_int
        L.tmpc_get_kernel_path.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_get_kernel_path.restype = C.c_int
        L.tmpc_kernel_name.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_kernel_name.restype = C.c_char_p
        L.tmpc_mc_run.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int] + [C.c_void_p] * 8 + [C.c_int32] + [C.c_void_p] * 6
        L.tmpc_mc_set_capture.argtypes = [C.c_void_p, C.c_int64]
        L.tmpc_mc_set_capture.restype = C.c_int
        L.tmpc_mc_get_capture.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tmpc_mc_get_capture.restype = C.c_int
        L.tmpc_set_solve_timing.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_set_solve_timing.restype = C.c_int
        L.tmpc_get_solve_ticks.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.tmpc_get_solve_ticks.restype = C.c_int
        L.tmpc_mc_get_solve_ticks.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.tmpc_mc_get_solve_ticks.restype = C.c_int
        L.tmpc_mc_set_device_rng.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_void_p]
        L.tmpc_mc_set_device_rng.restype = C.c_int
        L.tmpc_mc_get_physics_error.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.tmpc_mc_get_physics_error.restype = C.c_int
        L.tmpc_mc_set_warm_start.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_mc_set_warm_start.restype = C.c_int
        L.tmpc_mc_set_fused.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_mc_set_fused.restype = C.c_int
        L.tmpc_mc_last_fused.argtypes = [C.c_void_p]
        L.tmpc_mc_last_fused.restype = C.c_int
        L.tmpc_mc_run.restype = C.c_int
        L.tmpc_mc_replay.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int] + [C.c_void_p] * 8
        L.tmpc_mc_replay.restype = C.c_int
        L.tmpc_mc_set_actuator.argtypes = [C.c_void_p, C.c_int]
        L.tmpc_mc_set_actuator.restype = C.c_int
        L.tmpc_mc_set_plant.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]
        L.tmpc_mc_set_plant.restype = C.c_int
        L.tmpc_lp_batch.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tmpc_lp_batch.restype = C.c_int
        L.tmpc_synchronize.argtypes = [C.c_void_p]
        L.tmpc_synchronize.restype = C.c_int
        L.tmpc_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.tmpc_last_kernel_ms.restype = C.c_int
        L.tmpc_kernel_ms_total.argtypes = [C.c_void_p, C.POINTER(C.c_float), _ip, C.c_int]
        L.tmpc_kernel_ms_total.restype = C.c_int
        L.tmpc_get_dims.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _ip]
        L.tmpc_get_dims.restype = C.c_int
        L.tmpc_get_factoring.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _ip]
        L.tmpc_get_factoring.restype = C.c_int
        L.tmpc_get_condensed.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.tmpc_get_condensed.restype = C.c_int
        _lib = L
    return _lib


def pack_problem(d: dict):
    """dict (TubeTrackingMPC._problem_dict) -> (TmpcProblem, keep-alive list)."""
    p = TmpcProblem()
    keep = []
    nx, nu = int(d["nx"]), int(d["nu"])
    p.nx, p.nu, p.N = nx, nu, int(d["N"])
    p.fixed_x0 = int(d.get("fixed_x0", 0))
    p.extended = int(d.get("extended", 0))
    p.literal_terminal_row = int(d.get("literal_terminal_row", 1))
    p.max_iter = int(d.get("max_iter", 0))
    p.tol = float(d.get("tol", 0.0))
    square = {"A": (nx, nx), "B": (nx, nu), "Q": (nx, nx), "R": (nu, nu), "P": (nx, nx), "T": (nx, nx),
              "K": (nu, nx), "K_anc": (nu, nx)}
    widths = {"Hx": nx, "Hu": nu, "HT": 2 * nx + nu, "HZ": nx, "HZW": nx, "HTP": nx + nu}
    rows = {}
    for name in _PTR_FIELDS:
        v = d.get(name)
        if v is None:
            setattr(p, name, _dp())
            continue
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
        if name in square:
            if a.size != square[name][0] * square[name][1]:
                raise ValueError(f"{name} has {a.size} entries, expected shape {square[name]}")
            a = np.ascontiguousarray(a.reshape(square[name]))
        elif name in widths:
            if a.ndim != 2 or a.shape[1] != widths[name]:
                raise ValueError(f"{name} must have {widths[name]} columns, got shape {a.shape}")
            rows[name] = a.shape[0]
        else:                                   # right-hand sides
            a = np.ascontiguousarray(a.reshape(-1))
            rows[name] = a.shape[0]
        keep.append(a)
        setattr(p, name, a.ctypes.data_as(_dp))
    for hk, Hk in (("hx", "Hx"), ("hu", "Hu"), ("hT", "HT"), ("hZ", "HZ"), ("hZW", "HZW"), ("hTP", "HTP")):
        if rows.get(hk, 0) != rows.get(Hk, 0):
            raise ValueError(f"{Hk} has {rows.get(Hk, 0)} rows but {hk} has {rows.get(hk, 0)} entries")
    p.rx, p.ru, p.rT = rows.get("Hx", 0), rows.get("Hu", 0), rows.get("HT", 0)
    p.rZ, p.rZW = rows.get("HZ", 0), rows.get("HZW", 0)
    p.rTP = rows.get("HTP", 0)
    p.terminal_equality = int(d.get("terminal_equality", 0))
    return p, keep


class Handle:
    def __init__(self, ptr, nx, nu, N, nvariants):
        self.ptr, self.nx, self.nu, self.N, self.nvariants = ptr, nx, nu, N, nvariants

    def error(self) -> str:
        return lib().tmpc_last_error(self.ptr).decode()


def create(problem: dict, device: int = 0) -> Handle:
    L = lib()
    p, _keep = pack_problem(problem)
    h = C.c_void_p()
    rc = L.tmpc_create(C.byref(p), int(device), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"tmpc_create failed ({rc}): {L.tmpc_last_error(None).decode()}")
    return Handle(h, p.nx, p.nu, p.N, 2 if p.extended else 1)


def destroy(h: Handle):
    if h is not None and h.ptr:
        lib().tmpc_destroy(h.ptr)
        h.ptr = None


def get_dims(h: Handle, variant: int = 0):
    nv, nc, npar = C.c_int32(), C.c_int32(), C.c_int32()
    if lib().tmpc_get_dims(h.ptr, variant, C.byref(nv), C.byref(nc), C.byref(npar)) != 0:
        raise RuntimeError("tmpc_get_dims failed")
    return nv.value, nc.value, npar.value


def get_factoring(h: Handle, variant: int = 0):
    """include/tmpc.h: tmpc_get_factoring -> (general rows, rows of the factored block, its rank)."""
    nd, ncc, kc = C.c_int32(), C.c_int32(), C.c_int32()
    if lib().tmpc_get_factoring(h.ptr, variant, C.byref(nd), C.byref(ncc), C.byref(kc)) != 0:
        raise RuntimeError("tmpc_get_factoring failed")
    return nd.value, ncc.value, kc.value


def get_condensed(h: Handle, variant: int = 0) -> dict:
    nv, nc, _ = get_dims(h, variant)
    out = dict(H=np.empty((nv, nv)), F1=np.empty((nv, h.nx)), F2=np.empty((nv, h.nx)),
               G=np.empty((nc, nv)), g0=np.empty(nc), E=np.empty((nc, h.nx)))
    rc = lib().tmpc_get_condensed(h.ptr, variant, *[out[k].ctypes.data_as(_dp) for k in ("H", "F1", "F2", "G", "g0", "E")])
    if rc != 0:
        raise RuntimeError("tmpc_get_condensed failed")
    return out


TICK_SECONDS = 1e-8      # s_memrealtime: constant 100 MHz (include/tmpc.h, tmpc_set_solve_timing)


def solve_batch(h: Handle, x, r, variant=None, want_traj: bool = True, timing: bool = False) -> dict:
    """Host-pointer entry (tmpc_solve_batch): numpy in, numpy out.  timing: also `solve_time` (B,), the seconds every
    instance spent in its wavefront / workgroup (tmpc_set_solve_timing)."""
    B = x.shape[0]
    if lib().tmpc_set_solve_timing(h.ptr, int(bool(timing))) != 0:
        raise RuntimeError(h.error())
    nx, nu, N = h.nx, h.nu, h.N
    out = dict(u_nom=np.empty((B, N, nu)), x_nom0=np.empty((B, nx)), xu_ss=np.empty((B, nx + nu)),
               x_nom=np.empty((B, N + 1, nx)) if want_traj else None,
               status=np.empty(B, np.int32), iters=np.empty(B, np.int32))
    vptr = None
    if variant is not None:
        var = np.ascontiguousarray(np.broadcast_to(np.asarray(variant, dtype=np.uint8).reshape(-1), (B,)))
        vptr = var.ctypes.data
    rc = lib().tmpc_solve_batch(h.ptr, B, x.ctypes.data, r.ctypes.data, vptr,
                                out["u_nom"].ctypes.data, out["x_nom0"].ctypes.data, out["xu_ss"].ctypes.data,
                                out["x_nom"].ctypes.data if want_traj else None,
                                out["status"].ctypes.data, out["iters"].ctypes.data)
    if rc != 0:
        raise RuntimeError(f"tmpc_solve_batch failed ({rc}): {h.error()}")
    out["x_ss"] = out["xu_ss"][:, :nx]
    out["u_ss"] = out["xu_ss"][:, nx:]
    if timing:
        ticks = np.empty(B, np.int64)
        if lib().tmpc_get_solve_ticks(h.ptr, B, ticks.ctypes.data) != 0:
            raise RuntimeError(h.error())
        out["solve_time"] = ticks * TICK_SECONDS
    return out


def solve_batch_device(h: Handle, B: int, x_ptr, r_ptr, var_ptr, u_ptr, x0_ptr, ss_ptr, xn_ptr, st_ptr, it_ptr):
    """Device-pointer entry (tmpc_solve_batch_device): raw addresses (e.g. tensor.data_ptr())."""
    rc = lib().tmpc_solve_batch_device(h.ptr, int(B), x_ptr, r_ptr, var_ptr, u_ptr, x0_ptr, ss_ptr, xn_ptr, st_ptr, it_ptr)
    if rc != 0:
        raise RuntimeError(f"tmpc_solve_batch_device failed ({rc}): {h.error()}")


def synchronize(h: Handle):
    if lib().tmpc_synchronize(h.ptr) != 0:
        raise RuntimeError(h.error())


def last_kernel_ms(h: Handle) -> float:
    ms = C.c_float()
    if lib().tmpc_last_kernel_ms(h.ptr, C.byref(ms)) != 0:
        raise RuntimeError(h.error())
    return float(ms.value)


def kernel_ms_total(h: Handle, reset: bool = True):
    """(sum of per-call device ms, number of calls) since the last reset."""
    ms, cnt = C.c_float(), C.c_int32()
    if lib().tmpc_kernel_ms_total(h.ptr, C.byref(ms), C.byref(cnt), int(reset)) != 0:
        raise RuntimeError(h.error())
    return float(ms.value), int(cnt.value)


def kernel_name(h: Handle, variant: int = 0) -> str:
    """Name of the kernel instantiation that solves `variant` (as in a rocprofv3 kernel trace)."""
    return lib().tmpc_kernel_name(h.ptr, int(variant)).decode()


KERNEL_PATHS = {"auto": 0, "wave": 1, "block": 2}


def set_kernel_path(h: Handle, path):
    """include/tmpc.h: tmpc_set_kernel_path ('auto' | 'wave' | 'block')."""
    code = KERNEL_PATHS[path] if isinstance(path, str) else int(path)
    if lib().tmpc_set_kernel_path(h.ptr, code) != 0:
        raise RuntimeError(h.error())


def get_kernel_path(h: Handle, variant: int = 0) -> str:
    code = lib().tmpc_get_kernel_path(h.ptr, int(variant))
    if code < 0:
        raise RuntimeError("tmpc_get_kernel_path failed")
    return {v: k for k, v in KERNEL_PATHS.items()}[code]


def mc_set_actuator(h: Handle, smart: bool):
    """include/tmpc.h: tmpc_mc_set_actuator (False: consistent actuator, True: plain smart actuator of the R-MPC loop)."""
    if lib().tmpc_mc_set_actuator(h.ptr, 1 if smart else 0) != 0:
        raise RuntimeError(h.error())


def mc_set_plant(h: Handle, plant=None, Th: float = 0.02, substeps: int = 10):
    """include/tmpc.h: tmpc_mc_set_plant.  plant: None / 'linear', or 'cartpole' (workloads.CARTPOLE_PARAMS)."""
    if plant in (None, "linear"):
        rc = lib().tmpc_mc_set_plant(h.ptr, 0, None, 0)
    elif plant == "cartpole":
        from .workloads import CARTPOLE_PARAMS as P
        par = (C.c_double * 7)(P["M"], P["m"], P["b"], P["I"], P["g"], P["l"], float(Th))
        rc = lib().tmpc_mc_set_plant(h.ptr, 1, par, int(substeps))
    else:
        raise ValueError(f"unknown plant {plant!r}")
    if rc != 0:
        raise RuntimeError(h.error())


MC_FUSED = {"off": 0, "on": 1, "auto": 2, False: 0, True: 1, None: 2}      # include/tmpc.h: TMPC_MC_FUSED_*


def mc_run(h: Handle, p_loss, ref, th_u, ga_u, w, x0=None, Z=None, extended: bool = False, warm_start: bool = False,
           capture=None, timing: bool = False, physics_substeps: int = 0, device_rng=None, fused=None) -> dict:
    """include/tmpc.h: tmpc_mc_run -- the closed loop over the lossy network, resident on the device.
    warm_start: tmpc_mc_set_warm_start for this call; capture: index of a trajectory to record (tmpc_mc_set_capture) ->
    x_traj (T, nx), x_nom_traj (T, nx), u_traj (T, nu) in the result; timing: per trajectory the mean and the maximum
    time of its T solves in seconds (solve_time_mean, solve_time_max; tmpc_set_solve_timing); physics_substeps > 0 (a
    nonlinear plant was set with that many steps per sampling period): tracking_error_physics, the scripts' tracking error
    over the physics-rate trajectory (tmpc_mc_get_physics_error, results_nonlinear_system.py:361).
    device_rng = (seed, first_trajectory, w_bound): the realisations are drawn on the device (tmpc_mc_set_device_rng;
    montecarlo.draw_realisations_philox gives the same numbers on the host); th_u, ga_u, w are then ignored and may be None,
    the batch is len(p_loss) x len(ref).
    fused: "on" / "off" / "auto" (None) -- tmpc_mc_set_fused: one launch for all T steps, a launch pair per step, or the library's
    choice; the result's "fused" says what ran."""
    if lib().tmpc_set_solve_timing(h.ptr, int(bool(timing))) != 0:
        raise RuntimeError(h.error())
    if lib().tmpc_mc_set_fused(h.ptr, MC_FUSED[fused]) != 0:
        raise RuntimeError(h.error())
    if lib().tmpc_mc_set_warm_start(h.ptr, int(bool(warm_start))) != 0:
        raise RuntimeError(h.error())
    if lib().tmpc_mc_set_capture(h.ptr, -1 if capture is None else int(capture)) != 0:
        raise RuntimeError(h.error())
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    p_loss, ref = c(p_loss), c(ref)
    if device_rng is not None:
        seed, first, w_bound = device_rng
        wb = None if w_bound is None else c(w_bound).reshape(h.nx)
        if lib().tmpc_mc_set_device_rng(h.ptr, 1, int(seed), int(first), None if wb is None else wb.ctypes.data) != 0:
            raise RuntimeError(h.error())
        th_u = ga_u = w = None
        B, T = p_loss.shape[0], ref.shape[0]
    else:
        if lib().tmpc_mc_set_device_rng(h.ptr, 0, 0, 0, None) != 0:
            raise RuntimeError(h.error())
        th_u, ga_u, w = c(th_u), c(ga_u), c(w)
        B, T = th_u.shape
        if ga_u.shape != (B, T) or w.shape != (B, T, h.nx) or p_loss.shape != (B,) or ref.shape != (T,):
            raise ValueError("mc_run: inconsistent shapes")
    x0c = None if x0 is None else c(x0).reshape(B, h.nx)
    HZ = hZ = None
    rZ = 0
    if Z is not None:
        HZ, hZ = c(Z.A), c(Z.b)
        rZ = HZ.shape[0]
    out = dict(err2=np.empty(B), tube_violations=np.empty(B, np.int32), not_optimal=np.empty(B, np.int32),
               x_final=np.empty((B, h.nx)), consistent=np.empty(B), iters_sum=np.empty(B, np.int32))
    ptr = lambda a: None if a is None else a.ctypes.data
    rc = lib().tmpc_mc_run(h.ptr, B, T, int(bool(extended)), ptr(p_loss), ptr(ref), ptr(th_u), ptr(ga_u), ptr(w), ptr(x0c),
                           ptr(HZ), ptr(hZ), rZ, ptr(out["err2"]), ptr(out["tube_violations"]), ptr(out["not_optimal"]),
                           ptr(out["x_final"]), ptr(out["consistent"]), ptr(out["iters_sum"]))
    if rc != 0:
        raise RuntimeError(f"tmpc_mc_run failed ({rc}): {h.error()}")
    if capture is not None:
        out["x_traj"], out["x_nom_traj"], out["u_traj"] = np.empty((T, h.nx)), np.empty((T, h.nx)), np.empty((T, h.nu))
        if lib().tmpc_mc_get_capture(h.ptr, T, ptr(out["x_traj"]), ptr(out["x_nom_traj"]), ptr(out["u_traj"])) != 0:
            raise RuntimeError(h.error())
    if timing:
        tsum, tmax = np.empty(B, np.int64), np.empty(B, np.int64)
        if lib().tmpc_mc_get_solve_ticks(h.ptr, B, tsum.ctypes.data, tmax.ctypes.data) != 0:
            raise RuntimeError(h.error())
        out["solve_time_mean"], out["solve_time_max"] = tsum * (TICK_SECONDS / max(T, 1)), tmax * TICK_SECONDS
    if physics_substeps > 0:
        out["err2_physics"] = np.empty(B)
        if lib().tmpc_mc_get_physics_error(h.ptr, B, ptr(out["err2_physics"])) != 0:
            raise RuntimeError(h.error())
        out["tracking_error_physics"] = np.sqrt(out["err2_physics"]) / (T * physics_substeps)
    out["loop_mode"] = int(lib().tmpc_mc_last_fused(h.ptr))       # 1: one launch per sweep; 2: one launch per problem and step; 0: + a state-machine launch
    out["fused"] = out["loop_mode"] == 1
    out["tracking_error"] = np.sqrt(out["err2"]) / T
    out["consistent_estimate_error"] = float(out["consistent"].max()) if B else 0.0
    out["iters_mean"] = float(out["iters_sum"].sum()) / max(B * T, 1)            # interior-point iterations per solve
    return out


def mc_replay(h: Handle, U, theta, gamma, w, xn0=None, x0=None, extended: bool = False, smart: bool = False) -> dict:
    """include/tmpc.h: tmpc_mc_replay -- the device-side estimator / actuator state machines driven by GIVEN controller
    packets (no QP is solved).  U (B, T, N+1, nu): packets, terminal column last; theta, gamma (B, T): arrival flags;
    w (B, T, nx); xn0 (B, T, nx) for the extended controller.  Returns per step x (state after the step), x_hat (estimate
    after the step), x_nom (nominal state in the plant's packet), u (applied input), s, Theta, q."""
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    U, w = c(U), c(w)
    B, T = U.shape[:2]
    if U.shape != (B, T, h.N + 1, h.nu) or w.shape != (B, T, h.nx):
        raise ValueError("mc_replay: inconsistent shapes")
    th = np.ascontiguousarray(np.asarray(theta).reshape(B, T) != 0, dtype=np.uint8)
    ga = np.ascontiguousarray(np.asarray(gamma).reshape(B, T) != 0, dtype=np.uint8)
    xn0c = None if xn0 is None else c(xn0).reshape(B, T, h.nx)
    x0c = None if x0 is None else c(x0).reshape(B, h.nx)
    tf = np.empty((B, T, 3 * h.nx + h.nu))
    ti = np.empty((B, T, 3), np.int32)
    mc_set_actuator(h, smart)
    ptr = lambda a: None if a is None else a.ctypes.data
    try:
        rc = lib().tmpc_mc_replay(h.ptr, B, T, int(bool(extended)), ptr(U), ptr(xn0c), ptr(th), ptr(ga), ptr(w), ptr(x0c), ptr(tf), ptr(ti))
    finally:
        mc_set_actuator(h, False)
    if rc != 0:
        raise RuntimeError(f"tmpc_mc_replay failed ({rc}): {h.error()}")
    nx = h.nx
    return dict(x=tf[:, :, :nx], x_hat=tf[:, :, nx:2 * nx], x_nom=tf[:, :, 2 * nx:3 * nx], u=tf[:, :, 3 * nx:],
                s=ti[:, :, 0], Theta=ti[:, :, 1], q=ti[:, :, 2])


def lp_batch(H, h, Cmat, relax=None, relax_by: float = 1.0, device: int = 0, want_x: bool = False) -> dict:
    """Batch of support-function LPs over one polytope (include/tmpc.h: tmpc_lp_batch):
    val[b] = max Cmat[b] . x  s.t.  H x <= h, row relax[b] of h raised by relax_by."""
    L = lib()
    H = np.ascontiguousarray(H, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64).reshape(-1)
    Cm = np.ascontiguousarray(np.atleast_2d(Cmat), dtype=np.float64)
    nr, d = H.shape
    if h.size != nr or Cm.shape[1] != d:
        raise ValueError("lp_batch: shapes of H (nr x d), h (nr), C (B x d) do not agree")
    B = Cm.shape[0]
    rel = None if relax is None else np.ascontiguousarray(relax, dtype=np.int32).reshape(-1)
    if rel is not None and rel.size != B:
        raise ValueError("lp_batch: relax needs one row index per objective")
    val = np.empty(B)
    x = np.empty((B, d)) if want_x else None
    st = np.empty(B, dtype=np.int32)
    it = np.empty(B, dtype=np.int32)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = L.tmpc_lp_batch(int(device), d, nr, ptr(H), ptr(h), B, ptr(Cm), ptr(rel), float(relax_by),
                         ptr(val), ptr(x), ptr(st), ptr(it))
    if rc != 0:
        raise RuntimeError(f"tmpc_lp_batch failed ({rc}): {L.tmpc_last_error(None).decode()}")
    out = {"val": val, "status": st, "iters": it}
    if want_x:
        out["x"] = x
    return out
