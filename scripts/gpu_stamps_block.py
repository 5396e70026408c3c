"""Developer script: per-phase times of the block kernel from the diagnostic build (lib/libtmpc_stamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
_native.LIB_PATH = os.path.join(common.PKG, "lib", "libtmpc_stamps.so")
names = ["setup", "P1 rows", "P2 grad+G'v", "P3 init M", "MFMA G'DG", "Cholesky", "solve1", "P5 rows", "P6 G'v+solve2", "P7+P8 rows", "refinement", "outputs"]
def run(tag, mpc, X, R, var, B, variant):
    h = mpc._handle
    L = _native.lib()
    L.tmpc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
    for _ in range(2):
        o = mpc._solve(X[:B], R[:B], None if var is None else var[:B], want_traj=False)
    buf = (C.c_longlong * 12)()
    L.tmpc_debug_stamps(h.ptr, variant, buf)
    t = np.array(buf[:12], dtype=float)
    it = max(int(o["iters"][0]), 1)
    print(f"{tag} B={B}: instance 0 iters {it}, total {t.sum()/100:.1f} us (100 MHz ticks), kernel {_native.last_kernel_ms(h)*1e3:.1f} us")
    for n_, v in zip(names, t):
        print(f"   {n_:16s} {v/100:9.1f} us  {100*v/t.sum():5.1f}%   per-iter {v/it/100:8.2f} us")
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
idx = np.r_[40, np.random.default_rng(0).integers(0, len(S), 4095)]
mpc, w = common.make_mpc("cartpole", 10, True, create=True)
mpc.set_kernel_path("block")
for B in (1, 512, 4096):
    run("cartpole N=10 block T=1", mpc, S[idx, :4].copy(), S[idx, 4:].copy(), None, B, 0)
mpc, w = common.make_mpc("cartpole", 20, True, extended=True, create=True)
SX = common.harvest_states("cartpole", 20, True, [[0.5], [-0.4, 0.3]], 40, seed=4, disturb=False, extended=True)
idx = np.random.default_rng(0).integers(0, len(SX), 4096)
idx[0] = 5
one = np.ones(4096, dtype=np.uint8)
for B in (1, 512, 4096):
    run("ext N=20 variant 1 block T=2", mpc, SX[idx, :4].copy(), SX[idx, 4:].copy(), one, B, 1)
mpc, w = common.make_mpc("synthetic", 30, True, create=True)
rng = np.random.default_rng(0)
X = rng.uniform(-0.95, 0.95, (1024, 12)) * mpc._Xc.b[:12]
R = np.zeros((1024, 12)); R[:, 0] = rng.uniform(-2, 2, 1024)
for B in (1, 256, 1024):
    run("synthetic T=8", mpc, X, R, None, B, 0)
