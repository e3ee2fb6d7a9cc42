import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("bmpc_k_ric")]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
per=collections.defaultdict(list)
for r in rows: per[r["Kernel_Name"].split("(")[0]].append(((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r.get("Grid_Size_X",0) or 0)))
for k,v in per.items():
    print(k, len(v), "durations us (grid):", " ".join("%.0f(%d)"%(d,g//128) for d,g in v))
