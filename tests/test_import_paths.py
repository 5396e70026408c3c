"""Every module path the reference's scripts and modules import binds to this package (CPU, no GPU needed)."""
import importlib

import numpy as np
import pytest

import common  # noqa: F401  (puts the package on sys.path)

# reference: Results/results_linear_system.py:12-15, Results/results_linear_system_with_extendedMPC.py (same four),
# src/LinearMPCOverNetworks/TrackingMPC.py:15-16, TubeRegulatorMPC.py:11-12, TubeTrackingMPC.py:16-17
MODULES = {
    "LinearMPCOverNetworks.TubeTrackingMPC": ["TubeTrackingMPC", "ExtendedTubeTrackingMPC"],
    "LinearMPCOverNetworks.TrackingMPC": ["TrackingMPC"],
    "LinearMPCOverNetworks.SmartActuator": ["SmartActuator", "ConsistentActuator"],
    "LinearMPCOverNetworks.Estimator": ["Estimator", "RobustEstimator"],
    "LinearMPCOverNetworks.RegulatorMPC": ["RegulatorMPC"],
    "LinearMPCOverNetworks.TubeRegulatorMPC": ["TubeRegulatorMPC"],
    "LinearMPCOverNetworks.utils_polytope": ["support", "pont_diff", "calculate_RPI",
                                             "calculate_minimal_robust_positively_invariant_set",
                                             "calculate_maximum_admissible_output_set"],
}


@pytest.mark.parametrize("path", sorted(MODULES))
def test_reference_import_path(path):
    mod = importlib.import_module(path)
    for name in MODULES[path]:
        assert hasattr(mod, name), f"{path}.{name}"


def test_regulator_is_the_root_of_the_class_chain():
    """`from LinearMPCOverNetworks.RegulatorMPC import RegulatorMPC` (reference TubeRegulatorMPC.py:12) yields the class the
    tube controllers derive from; its constructor and setters behave as RegulatorMPC.py:11-43, :93-94."""
    from LinearMPCOverNetworks.RegulatorMPC import RegulatorMPC
    import LinearMPCOverNetworks.RegulatorMPC as RegulatorMPCModule          # reference TrackingMPC.py:16
    from LinearMPCOverNetworks.TubeRegulatorMPC import TubeRegulatorMPC
    from LinearMPCOverNetworks.TubeTrackingMPC import TubeTrackingMPC
    from LinearMPCOverNetworks.TrackingMPC import TrackingMPC
    assert RegulatorMPCModule.RegulatorMPC is RegulatorMPC
    assert RegulatorMPC.__module__ == "LinearMPCOverNetworks.RegulatorMPC"
    assert issubclass(TubeRegulatorMPC, RegulatorMPC) and issubclass(TubeTrackingMPC, TubeRegulatorMPC)
    assert issubclass(TrackingMPC, RegulatorMPC)
    w = common.workload("double_integrator")
    reg = RegulatorMPC(w["A"], w["B"], w["Q"], w["R"], 7.0)
    assert (reg._nx, reg._nu, reg._N) == (2, 1, 7) and isinstance(reg._N, int)
    assert reg._X is None and reg._U is None
    reg.set_state_constraints(w["X"])
    reg.set_input_constraints(w["U"])
    assert reg._X.A.shape[1] == 2 and reg._U.A.shape[1] == 1
    reg.set_solver("hip")
    with pytest.raises(ValueError):
        reg.set_solver("osqp")
    assert np.array_equal(reg._A, w["A"])
