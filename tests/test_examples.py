"""The example scripts (the reference's example scenarios through this package) run and report sane numbers."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script", ["tube_tracking_mpc.py", "tube_tracking_mpc_over_lossy_network.py"])
def test_example_runs(hip_lib, capsys, script, monkeypatch):
    from LinearMPCOverNetworks import polytope_lite as pl
    old = pl.set_lp_backend("hip")           # the examples use the package defaults
    monkeypatch.setattr(sys, "argv", [script])
    try:
        runpy.run_path(os.path.join(ROOT, "examples", script), run_name="__main__")
    finally:
        pl.set_lp_backend(old)
    out = capsys.readouterr().out
    assert "120 of 120 steps" in out                       # the tube guarantee held at every step
    assert "Input constraints violated" not in out
    assert "reference +4: x1 at the end of the segment = +4." in out
