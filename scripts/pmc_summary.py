"""Summarise rocprofv3 --pmc counter_collection.csv files: per-dispatch mean of each counter, per kernel whose name contains
the given substring.  Usage: python scripts/pmc_summary.py <kernel-substring> gpurun_out/pmc_*/ > profiles/xxx.txt"""
import collections
import csv
import glob
import sys

sub = sys.argv[1]
res = collections.OrderedDict()
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
        # only the full-size launches of each kernel (the set-up phase of bench.py launches the same kernel on small batches)
        gmax = collections.defaultdict(int)
        for r in rows:
            gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size"]))
        acc = collections.defaultdict(list)
        for r in rows:
            if int(r["Grid_Size"]) != gmax[r["Kernel_Name"]]:
                continue
            name = r["Kernel_Name"].split("(tmpc::DeviceQP")[0].replace("void ", "").replace("(anonymous namespace)::", "")
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            # (smaller batches have the full grid as well since the first work items are dealt wave-major: a full-size dispatch is one whose
            # counter is within a factor of two of the largest -- every counter collected here grows with the work of a launch)
            v = [x for x in v if 2 * x >= max(v)] if max(v) > 0 else v
            res[k] = (sum(v) / len(v), len(v))
last = None
for (name, ctr), (v, n) in res.items():
    if name != last:
        print(f"== {name}")
        last = name
    print(f"   {ctr:30s} {v:18.1f}   (mean over {n} dispatches)")
