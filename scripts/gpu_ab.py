"""Developer script: time several builds of the library on the bench batch (one subprocess per build)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for lib in sys.argv[1:]:
    env = dict(os.environ, TMPC_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(here, "gpu_check.py")], env=env, capture_output=True, text=True, timeout=300).stdout
    keep = [l for l in out.splitlines() if l.startswith(("max |u_nom", "B=4096", "status"))]
    print(lib, "\n   " + "\n   ".join(keep), flush=True)
