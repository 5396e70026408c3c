"""Generates tests/golden/*_sets.npz: the offline sets (Z, Xc, Uc, Xf, Z(-)W) of the
workloads named in BASELINE.json, computed by THIS repository's host-side set-up
stage (LinearMPCOverNetworks/utils_polytope.py, scipy LPs).

These are not reference outputs (the reference's `polytope`/`control` stack is not
installable here); they are cached because the cartpole set-up takes ~1 minute of
LPs, so that tests, smoke() and bench.py all start from the same sets and never
need /root/reference.  Row counts (854 / 420 for the cartpole) match the sizing
probe recorded in SURVEY.md appendix D.

    python tests/golden/make_sets.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "robust-tracking-mpc-over-lossy-networks_amd"))

from LinearMPCOverNetworks import polytope_lite  # noqa: E402
from LinearMPCOverNetworks import utils_polytope as up  # noqa: E402
from LinearMPCOverNetworks import workloads  # noqa: E402
from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC  # noqa: E402


def make(name, w, rpi_method):
    mpc = TubeTrackingMPC(w["A"], w["B"], w["Q"], w["R"], 10)
    mpc.set_input_constraints(w["U"])
    mpc.set_state_constraints(w["X"])
    mpc.determine_mRPI(w["W"], rpi_method=rpi_method)
    mpc.tighten_constraints()
    mpc.determine_Xf()
    mpc._ZmW = up.pont_diff(mpc._Z, w["W"])
    sets = mpc.export_sets()
    path = os.path.join(HERE, f"{name}_sets.npz")
    np.savez_compressed(path, **sets)
    print(name, {k: v.shape for k, v in sets.items()}, "->", path)


if __name__ == "__main__":
    polytope_lite.set_lp_backend("scipy")      # the fixtures come from the solver the reference calls
    make("double_integrator_rakovic", workloads.double_integrator(), 0)
    make("double_integrator_darup", workloads.double_integrator(), 1)
    make("cartpole", workloads.cartpole(), 1)
    make("synthetic", workloads.synthetic(), 1)
