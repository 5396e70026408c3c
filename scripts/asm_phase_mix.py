#!/usr/bin/env python3
"""Static instruction mix of the wave kernel between its phase marks (the `; MARK p` comments STAMP() leaves in a
non-diagnostic build).  The sweeps are fully unrolled, so the static count of a phase is what one interior-point iteration
issues.  Usage: asm_phase_mix.py kernel.s [first_mark last_mark]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rcp_f64", "v_rsq_f64", "v_div_scale_f64",
              "v_div_fmas_f64", "v_div_fixup_f64", "v_fmac_f64", "v_ldexp_f64", "v_frexp_mant_f64", "v_cmp_f64"): return "fp64"
    if op.startswith("v_cmp") and op.endswith("f64"): return "cmp64"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"): return "readlane"
    if op.startswith("v_writelane"): return "writelane"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("ds_"): return "ds_other"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op == "s_waitcnt": return "s_waitcnt"
    if op == "s_nop": return "s_nop"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_cbranch") or op == "s_branch": return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_"): return "valu_other"
    return "other"


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    marks = [(i, int(m.group(1))) for i, l in enumerate(lines) if (m := re.search(r"; MARK (\d+)", l))]
    rows = []
    for (i0, p0), (i1, p1) in zip(marks, marks[1:]):
        mix = collections.Counter()
        dpp = 0
        for l in lines[i0:i1]:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            op = t.split()[0]
            mix[classify(op)] += 1
            if "dpp" in t or "quad_perm" in t or "row_" in t: dpp += 1
        rows.append((p0, p1, mix, dpp))
    keys = ["fp64", "cmp64", "mfma", "readlane", "v_mov", "v_cndmask", "valu_other", "ds_read", "ds_write", "vmem", "salu", "s_nop", "s_waitcnt", "branch"]
    print("%-9s %6s " % ("phase", "total") + " ".join("%9s" % k for k in keys) + "   dpp")
    for p0, p1, mix, dpp in rows:
        print("%3d->%-3d  %6d " % (p0, p1, sum(mix.values())) + " ".join("%9d" % mix[k] for k in keys) + "  %4d" % dpp)


if __name__ == "__main__":
    main()
