"""Developer script: where a WARM-STARTED solve of the closed loop spends its cycles (the diagnostic build lib/libtmpc_stamps.so,
-DTMPC_STAMPS, all shapes): the stamps are those of trajectory 0 at the last time step of a per-step loop.  N from the command line."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common
from LinearMPCOverNetworks import _native, montecarlo
_native.LIB_PATH = os.path.join(common.PKG, "lib", "libtmpc_stamps.so")
names = ["setup", "sweepA+reduce", "grad/conv", "factor+solve1", "sweepB+reduce", "solve2", "sweepD+update", "polish rest", "outputs", "loop-top",
         "ref: compaction", "ref: expand+T", "ref: S+factor", "ref: steps", "ref: verify", "-"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, T = 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 40
mpc, w = common.make_mpc("cartpole", N, True, create=True)
th, ga, wd = montecarlo.draw_realisations(B, T, w["w_bound"], seed=99)
L = _native.lib()
L.tmpc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for warm in (True, False):
    cl = mpc.run_closed_loop(np.full(B, 0.3), 0.5 * np.ones(T), th, ga, wd, warm_start=warm, fused="off")
    buf = (C.c_longlong * 16)()
    L.tmpc_debug_stamps(mpc._handle.ptr, 0, buf)
    t = np.array(buf[:16], dtype=float)
    print(f"cartpole N={N} warm={warm}: iters/solve {cl['iters_mean']:.2f}; trajectory 0, last step: total {t.sum():.0f} cycles")
    for n_, v in zip(names, t):
        if v: print(f"   {n_:16s} {v:10.0f}  {100*v/t.sum():5.1f}%")
