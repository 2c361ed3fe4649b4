"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean of each counter."""
import csv, glob, sys, collections
tag = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else "fit_"
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if match in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:62s} {c:26s} n={len(v)} mean={sum(v)/len(v):.6g}")
