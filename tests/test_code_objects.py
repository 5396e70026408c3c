"""Resource notes of the gfx950 code objects inside lib/libtmpc_hip.so (no GPU needed: the notes are read from the
library's .hip_fatbin section, scripts/code_object_notes.py).  The wave-per-QP shapes must not touch scratch: builds of that
kernel with > 120 spilled registers once returned wrong statuses."""
import importlib.util
import os

import common

ROOT = os.path.dirname(common.PKG)
spec = importlib.util.spec_from_file_location("code_object_notes", os.path.join(ROOT, "scripts", "code_object_notes.py"))
notes = importlib.util.module_from_spec(spec)
spec.loader.exec_module(notes)


def _kernels():
    ks = notes.kernels(os.path.join(common.PKG, "lib", "libtmpc_hip.so"))
    dm = notes.demangle(list(ks))
    return {dm[n]: k for n, k in ks.items()}


def test_wave_shapes_have_no_private_segment():
    ks = _kernels()
    wave = {n: k for n, k in ks.items() if "::solve_kernel<" in n}
    assert len(wave) == 13, sorted(wave)
    for n, k in wave.items():
        assert k[".private_segment_fixed_size"] == 0, (n, k[".private_segment_fixed_size"], k[".vgpr_spill_count"])
        # (a spill count without a private segment is the accumulation-register file used as spill space: one wave per SIMD)
        assert k[".vgpr_spill_count"] <= 16, (n, k[".vgpr_spill_count"])
    bench = [k for n, k in wave.items() if "solve_kernel<11, 1, 0, 5, 4, 0, 8>" in n]
    assert len(bench) == 1 and bench[0][".vgpr_count"] <= 256 and bench[0][".vgpr_spill_count"] == 0      # two waves per SIMD


def test_fused_closed_loop_kernels_match_the_solve_kernels():
    """closed_loop_kernel<shape> is solve_kernel<shape>'s body with the trajectory's state machines between two solves (the record of
    the loop read through the constant address space, field by field): the same register budget, no scratch."""
    ks = _kernels()
    fused = {n: k for n, k in ks.items() if "::closed_loop_kernel<" in n}
    wave = {n.split("::solve_kernel")[1]: k for n, k in ks.items() if "::solve_kernel<" in n}
    assert len(fused) == 13, sorted(fused)
    for n, k in fused.items():
        shape = n.split("::closed_loop_kernel")[1].split("(")[0]
        twin = [v for m, v in wave.items() if m.split("(")[0] == shape]
        assert len(twin) == 1, (n, sorted(wave))
        assert k[".vgpr_count"] <= twin[0][".vgpr_count"] + 24, (n, k[".vgpr_count"], twin[0][".vgpr_count"])
        assert k[".private_segment_fixed_size"] == 0, (n, k[".private_segment_fixed_size"])
    bench = [k for n, k in fused.items() if "closed_loop_kernel<11, 1, 0, 5, 4, 0, 8>" in n]
    assert len(bench) == 1 and bench[0][".vgpr_count"] <= 256 and bench[0][".vgpr_spill_count"] == 0
    # the extended controller's form: one problem at one time step, the state machines of its trajectories inside
    step = {n: k for n, k in ks.items() if "::closed_loop_step_kernel<" in n}
    assert len(step) == 13, sorted(step)
    for n, k in step.items():
        assert k[".private_segment_fixed_size"] == 0, (n, k[".private_segment_fixed_size"])


def test_other_kernels_stay_within_their_known_footprint():
    """The block kernel carries no scratch since round 3 (model record and launch record behind one pointer each, read through
    the constant address space; lane- and record-derived values re-derived per phase: DESIGN.md 5.2).  The LP kernels and the
    closed-loop kernels: the LP shapes up to d = 16 carry none."""
    ks = _kernels()
    block = {n: k for n, k in ks.items() if "::solve_block_kernel<" in n}
    assert len(block) == 4
    for n, k in block.items():
        assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, (n, k[".vgpr_spill_count"], k[".private_segment_fixed_size"])
        assert k[".vgpr_count"] <= 256, (n, k[".vgpr_count"])        # T = 8: two waves per SIMD
    # the closed loop's state machines: one wave per trajectory, state vectors on the lanes -- no private arrays (round 3: a
    # thread per trajectory with 1168 B of them)
    step = [k for n, k in ks.items() if "mc_step_kernel" in n]
    assert len(step) == 1 and step[0][".private_segment_fixed_size"] == 0 and step[0][".vgpr_spill_count"] == 0
    assert not any("mc_post_kernel" in n or "mc_tube_kernel" in n for n in ks)
    for n, k in ks.items():
        if "::lp_kernel<" in n:
            # d <= 16: no scratch.  D = 32 (one wave per SIMD, all 512 registers: the normal matrix's column blocks) keeps
            # nine dwords of kernel-prologue values in scratch since the round-3 hand-over code (five stores and six loads
            # in the whole kernel, none inside a loop: `scratch_` lines of the -S output sit at the prologue and at two
            # phase boundaries)
            limit = 64 if "lp_kernel<32" in n else 0
            assert k[".private_segment_fixed_size"] <= limit, (n, k[".private_segment_fixed_size"])
