"""Summarise rocprofv3 --pmc CSVs for the solve kernel (helper for profiles/, not a test)."""
import collections, csv, glob, sys
acc = collections.defaultdict(float); calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bmpc_solve" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(f"{k},{acc[k]:.0f},{len(calls[k])}")
