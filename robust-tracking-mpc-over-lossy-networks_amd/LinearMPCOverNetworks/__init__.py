"""MI355X-native drop-in for the tube-tracking MPC hot path of
EricssonResearch/Robust-Tracking-MPC-over-Lossy-Networks.

The package keeps the reference's import path (`LinearMPCOverNetworks.TubeTrackingMPC`)
so that the reference's driver scripts bind to it unchanged; only the
per-timestep QP solve (`solve_optimization_problem` / `determine_packet`) and
what it needs is provided.
"""
