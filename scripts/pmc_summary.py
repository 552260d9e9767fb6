"""Summarise rocprofv3 --pmc counter_collection.csv files for the step kernel. Usage: python scripts/pmc_summary.py <dir> [...]"""
import csv, glob, statistics, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
        by = collections.defaultdict(list)
        for r in rows:
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(by.items()):
            print(f"{k:28s} n={len(v):4d} median={statistics.median(v):16.1f}")
