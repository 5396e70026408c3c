"""Summarise rocprofv3 --kernel-trace output (kernel_trace.csv): per kernel, the launches at its largest grid size (the
set-up phase of bench.py launches the solve kernel on small batches as well) and within a factor of two of the longest -- calls, average / min / max duration.
Usage: python scripts/trace_summary.py gpurun_out/prof_xxx [more dirs] > profiles/xxx_kernel_trace_summary.txt"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = collections.defaultdict(list)
        for r in rows:
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            per[r["Kernel_Name"]].append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        print(f"# {f}")
        print(f"{'kernel':70s} {'grid':>8s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
        for name, v in sorted(per.items(), key=lambda kv: -sum(t for _, t in kv[1])):
            gmax = max(g for g, _ in v)
            t = [x for g, x in v if g == gmax]
            # (round 4: the first work items of a launch are dealt wave-major, so every batch of at least one item per CU has the full grid;
            # the full-size launches are then the ones within a factor of two of the longest)
            t = [x for x in t if 2 * x >= max(t)]
            short = name.split("(tmpc::")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
            print(f"{short:70s} {gmax:8d} {len(t):6d} {sum(t) / len(t) / 1e3:10.2f} {min(t) / 1e3:10.2f} {max(t) / 1e3:10.2f}")
