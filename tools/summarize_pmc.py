"""Summarise rocprofv3 --pmc CSVs per kernel of the pipeline (helper for profiles/, not a test):
sum of every counter over all launches of each kernel, number of dispatches."""
import collections, csv, glob, sys
acc = collections.defaultdict(float); calls = collections.defaultdict(set)
for d in [a for a in sys.argv[1:] if not a.startswith('--') and not a.endswith('.json')]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("bmpc_"):
                acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("kernel,counter,sum,dispatches")
for k in sorted(acc):
    print(f"{k[0]},{k[1]},{acc[k]:.0f},{len(calls[k])}")

# per-batch HBM traffic of the whole hot path (bench.py reports it as roofline.traffic): FETCH_SIZE and
# WRITE_SIZE are in KB; the passes profile ONE synchronous batch (bench.py --steps 1 --warmup 0 --depth 1)
if "--traffic-json" in sys.argv:
    import json
    fetch = sum(v for (k, c), v in acc.items() if c == "FETCH_SIZE") * 1024
    write = sum(v for (k, c), v in acc.items() if c == "WRITE_SIZE") * 1024
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"fetch_bytes_raw": fetch, "write_bytes": write, "unit": "bytes per batch (B=8192, N=20)", "kernel_src_sha16": bench.kernel_src_sha(),
               "per_kernel_GB": {k: {"fetch_raw": acc.get((k, "FETCH_SIZE"), 0) * 1024 / 1e9, "write": acc.get((k, "WRITE_SIZE"), 0) * 1024 / 1e9}
                                 for k in sorted({k for k, _ in acc})}},
              open(sys.argv[sys.argv.index("--traffic-json") + 1], "w"), indent=1)
