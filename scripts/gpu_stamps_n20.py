"""Developer script: per-phase cycle shares of the N = 20 shapes of the wave kernel (base problem and packet-received problem) from the
diagnostic build (lib/libtmpc_stamps.so built with -DTMPC_STAMPS, all shapes).  Not part of the product or the tests."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
import numpy as np
from LinearMPCOverNetworks import _native, workloads
_native.LIB_PATH = os.path.join(os.path.dirname(_native.__file__), "..", "lib", "libtmpc_stamps.so")
names = ["setup", "sweepA+reduce", "grad/conv", "factor+solve1", "sweepB+reduce", "solve2", "sweepD+update", "polish rest", "outputs", "loop-top",
         "ref: compaction", "ref: expand+T", "ref: S+factor", "ref: steps", "ref: verify", "-"]
mpc, w = workloads.make_controller("cartpole", 20, True, extended=True, device=0)
X, R, G = workloads.harvest_closed_loop_states(mpc, w, 128, 32, seed=300, extended=True)
L = _native.lib()
L.tmpc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for var in (0, 1):
    sel = np.flatnonzero(G == var)[:4096]
    # a typical instance first: the stamps are those of instance 0
    o = mpc._solve(X[sel], R[sel], variant=np.full(len(sel), var, np.uint8), want_traj=False)
    med = int(np.argsort(o["iters"])[len(sel) // 2])
    order = np.r_[med, np.delete(np.arange(len(sel)), med)]
    for _ in range(2):
        o = mpc._solve(X[sel][order], R[sel][order], variant=np.full(len(sel), var, np.uint8), want_traj=False)
    buf = (C.c_longlong * 16)()
    L.tmpc_debug_stamps(mpc._handle.ptr, var, buf)
    t = np.array(buf[:16], dtype=float)
    it = max(int(o["iters"][0]), 1)
    print(f"variant {var} ({_native.kernel_name(mpc._handle, var) if hasattr(_native, 'kernel_name') else ''}): {len(sel)} instances, instance 0 iters {it} (mean {o['iters'].mean():.1f}), total {t.sum():.0f} cycles, kernel {_native.last_kernel_ms(mpc._handle)*1e3:.1f} us")
    for n_, v in zip(names, t):
        print(f"   {n_:16s} {v:10.0f}  {100*v/max(t.sum(),1):5.1f}%   per-iter {v/it:8.1f}")
