"""Summarise rocprofv3 --pmc counter_collection.csv files: per-dispatch mean of each counter for
the solve kernel.  Usage: python scripts/pmc_summary.py gpurun_out/pmc_*/ > profiles/xxx.txt"""
import csv, glob, sys, collections, json
res = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "solve_kernel" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = (sum(v) / len(v), len(v))
for k, (v, n) in res.items():
    print(f"{k:28s} {v:18.1f}   (mean over {n} dispatches)")
