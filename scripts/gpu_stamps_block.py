"""Developer script: per-phase cycle shares of the workgroup-per-QP kernel on BASELINE config 5 from the diagnostic
build (make OBJDIR=../build_stamps OUT=../lib/libtmpc_stamps.so EXTRA=-DTMPC_STAMPS).  Not part of the product or tests."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from LinearMPCOverNetworks import _native, workloads
_native.LIB_PATH = os.path.join(os.path.dirname(_native.__file__), "..", "lib", "libtmpc_stamps.so")
names = ["setup", "P1 rows", "P2 G'v", "M init", "G'DG mfma", "chol", "solve1", "P5 rows", "P6 G'v+solve", "P7/8 rows", "refine", "outputs", "inv I", "inv II", "inv III", "-"]
mpc, w = workloads.make_controller("synthetic", 30, True, device=0)
rng = np.random.default_rng(50)
B = 16384
X = rng.uniform(-0.5, 0.5, (B, 12)) * mpc._Xc.b[:12]
R = np.zeros((B, 12)); R[:, 0] = rng.uniform(-2, 2, B)
L = _native.lib()
L.tmpc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for nb in (256, 16384):
    for _ in range(2):
        o = mpc._solve(X[:nb], R[:nb], want_traj=False)
    buf = (C.c_longlong * 16)()
    L.tmpc_debug_stamps(mpc._handle.ptr, 0, buf)
    t = np.array(buf[:16], dtype=float)
    it = o["iters"]
    print(f"B={nb}: kernel {_native.last_kernel_ms(mpc._handle):.2f} ms; iters hist {np.bincount(it)}; last instance of workgroup 0 with iterations: {t.sum()/100:.1f} us")
    for n_, v in zip(names, t):
        print(f"   {n_:14s} {v/100:9.1f} us  {100*v/t.sum():5.1f}%")
