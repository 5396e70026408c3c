"""Developer script: per-phase cycle shares from the diagnostic build (lib/libtmpc_stamps.so,
built with -DTMPC_STAMPS).  Not part of the product or the tests."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native
_native.LIB_PATH = os.path.join(common.PKG, "lib", "libtmpc_stamps.so")
names = ["setup", "sweepA+reduce", "grad/conv", "factor+solve1", "sweepB+reduce", "solve2", "sweepD+update", "polish rest", "outputs", "loop-top",
         "ref: compaction", "ref: expand+T", "ref: S+factor", "ref: steps", "ref: verify", "-"]
def run(name, N, fixed, X, R, B):
    mpc, w = common.make_mpc(name, N, fixed, create=True)
    h = mpc._handle
    L = _native.lib()
    L.tmpc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
    for _ in range(2):
        o = mpc._solve(X[:B], R[:B], want_traj=False)
    buf = (C.c_longlong * 16)()
    L.tmpc_debug_stamps(h.ptr, 0, buf)
    t = np.array(buf[:16], dtype=float)
    it = max(int(o["iters"][0]), 1)
    print(f"{name} N={N} B={B}: instance 0 iters {it}, total {t.sum():.0f} ticks (100 MHz => {t.sum()/100:.1f} us), kernel {_native.last_kernel_ms(h)*1e3:.1f} us")
    for n_, v in zip(names, t):
        print(f"   {n_:16s} {v:10.0f} ticks  {100*v/t.sum():5.1f}%   per-iter {v/it:8.1f}")
S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
idx = np.r_[40, np.random.default_rng(0).integers(0, len(S), 4095)]
for B in (4096,):
    run("cartpole", 10, True, S[idx, :4].copy(), S[idx, 4:].copy(), B)
rng = np.random.default_rng(0)
X = rng.uniform(-1, 1, (1024, 2)) * [3.0, 0.5]; R = np.c_[rng.uniform(-9, 9, 1024), np.zeros(1024)]
for B in ():
    run("double_integrator", 5, False, X, R, B)
