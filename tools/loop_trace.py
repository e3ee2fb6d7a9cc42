"""Summarise a rocprofv3 --kernel-trace CSV of a closed-loop run with ONE group (tools/closed_loop_device.py --groups 1): per MPC step
(bmpc_loop_k_prepare ... bmpc_loop_k_finish) the number of super-steps, where the time goes by super-step index and by kernel."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("bmpc_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# MPC steps: from one bmpc_loop_k_prepare to the next
starts = [i for i, e in enumerate(ev) if e[0] == "bmpc_loop_k_prepare"]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print("MPC steps in the trace:", len(starts), "(the first", skip, "are skipped)")
tot = collections.defaultdict(float); n_steps = 0; wall_sum = 0.0; ss_count = []
bins = [(0, 10), (10, 25), (25, 50), (50, 100), (100, 200), (200, 400), (400, 10**9)]
bin_ms = collections.defaultdict(float); busy_sum = 0.0
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    seg = ev[a:b]
    wall = (seg[-1][2] - seg[0][1]) / 1e6
    wall_sum += wall; n_steps += 1
    ss = 0; t_prev = seg[0][1]
    for (n, s, e) in seg:
        tot[n] += (e - s) / 1e6; busy_sum += (e - s) / 1e6
        if n == "bmpc_k_rotate":
            for lo, hi in bins:
                if lo <= ss < hi: bin_ms[(lo, hi)] += (e - t_prev) / 1e6
            ss += 1; t_prev = e
    ss_count.append(ss)
print(f"{n_steps} steps: mean wall {wall_sum / n_steps:.1f} ms, kernel busy {busy_sum / n_steps:.1f} ms; super-steps per MPC step: mean {sum(ss_count) / n_steps:.0f}, min {min(ss_count)}, max {max(ss_count)}")
print("wall by super-step index:")
for lo, hi in bins:
    print(f"    [{lo:4d}, {hi if hi < 10**9 else 'inf'}): {bin_ms[(lo, hi)] / n_steps:7.1f} ms")
print("kernel time per MPC step:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]:
    print(f"    {k:24s} {v / n_steps:8.2f} ms")
