"""Summarise a rocprofv3 --kernel-trace CSV of ONE synchronous batch: per super-step (k_points ... k_rotate) the kernel
durations and the gaps between kernels, separately for the bulk and the straggler tail (helper for profiles/)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("bmpc_k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if name in ("bmpc_k_init_inst", "bmpc_k_init", "bmpc_k_init_fin", "bmpc_k_out", "bmpc_k_fin", "bmpc_k_mult", "bmpc_k_mult_sweep"):
        continue
    cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if name == "bmpc_k_rotate":
        steps.append(cur); cur = []
print("super-steps", len(steps))
def summarize(sel, tag):
    dur = collections.defaultdict(float); gap = 0.0; wall = 0.0
    for st in sel:
        for i, (n, s, e) in enumerate(st):
            dur[n] += (e - s) / 1e3
            if i: gap += max(0, s - st[i - 1][2]) / 1e3
        wall += (st[-1][2] - st[0][1]) / 1e3
    n = max(1, len(sel))
    print(f"{tag}: {len(sel)} super-steps, mean wall {wall / n:.1f} us (kernels {sum(dur.values()) / n:.1f} us, gaps inside {gap / n:.1f} us)")
    for k, v in sorted(dur.items(), key=lambda kv: -kv[1]):
        print(f"    {k:18s} {v / n:8.1f} us")
    if len(sel) > 1:
        between = sum(max(0, sel[i + 1][0][1] - sel[i][-1][2]) for i in range(len(sel) - 1)) / 1e3 / (len(sel) - 1)
        print(f"    gap between super-steps {between:.1f} us")
summarize(steps[:25], "bulk (first 25)")
summarize(steps[60:], "tail (from 60)")
total = (steps[-1][-1][2] - steps[0][0][1]) / 1e6
print(f"total {total:.1f} ms; bulk 25 steps {(steps[24][-1][2] - steps[0][0][1]) / 1e6:.1f} ms; steps 25-60 {(steps[59][-1][2] - steps[25][0][1]) / 1e6:.1f} ms; tail {(steps[-1][-1][2] - steps[60][0][1]) / 1e6:.1f} ms")
