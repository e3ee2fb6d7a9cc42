"""Summarise rocprofv3 --pmc CSVs per kernel of the pipeline (helper for profiles/, not a test):
sum of every counter over all launches of each kernel, number of dispatches."""
import collections, csv, glob, sys
acc = collections.defaultdict(float); calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("bmpc_"):
                acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("kernel,counter,sum,dispatches")
for k in sorted(acc):
    print(f"{k[0]},{k[1]},{acc[k]:.0f},{len(calls[k])}")
