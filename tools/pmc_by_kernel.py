"""Sum the counters of a rocprofv3 --pmc CSV (counter_collection.csv) per kernel (with --grid: per kernel and grid size)."""
import collections, csv, sys
by_grid = "--grid" in sys.argv
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if by_grid: k = (k, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k, " ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())), "launches", max(n[(k, c)] for c in acc[k]))
