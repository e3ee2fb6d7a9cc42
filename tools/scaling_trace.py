"""Per-kernel duration against the number of live instances: solves the first B instances of one configs[2] batch for a few
interior-point iterations (max_iter 4: every instance is alive in every super-step), B = 256 ... 8192.  Run under
`rocprofv3 --kernel-trace`, then `python tools/scaling_trace.py --parse <kernel_trace.csv>` prints, per B, the mean duration
of each pipeline kernel in the FIRST super-step of that solve (all B instances alive; later launches of the same solve work on
fewer and fewer instances).  One wavefront per SIMD = 1024 wavefronts = 3072 instances at N=20."""
import argparse
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIZES = [256, 1024, 2048, 3072, 4096, 6144, 8192]


def run():
    import numpy as np
    import torch  # noqa: F401  (HIP runtime first)
    from boundplanner_amd import scenes, solver
    if os.environ.get("BMPC_LIB"):          # A/B runs against another build of the library
        solver.LIB_PATH = os.path.abspath(os.environ["BMPC_LIB"])
    from boundplanner_amd.solver import HipBoundMPC
    N = 20
    be = HipBoundMPC(N, max_iter=4)
    d = scenes.make_batch(max(SIZES), N, 8192, be.fk, randomize_sets=True)
    x0, lbx, ubx, p = d["x0"], d["lbx"], d["ubx"], d["p"]
    for B in SIZES:
        r = be.solve_batch(x0[:B], lbx[:B], ubx[:B], p[:B])
        print(B, int(r["iters"].max()), flush=True)


def parse(path):
    rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith("bmpc_k_")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    solves, cur = [], []
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if name == "bmpc_k_eval_curv_split": name = "bmpc_k_eval_curv"      # (the two-wavefront k_eval beside k_curv: same column)
        if name == "bmpc_k_trial_spec": name = "bmpc_k_trial"
        if name == "bmpc_k_init_inst" and cur:
            solves.append(cur); cur = []
        cur.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    solves.append(cur)
    names = ["bmpc_k_points_pose", "bmpc_k_eval_curv", "bmpc_k_points", "bmpc_k_pose", "bmpc_k_eval", "bmpc_k_curv", "bmpc_k_ric", "bmpc_k_ric_lat", "bmpc_k_fwd", "bmpc_k_step",
             "bmpc_k_trial", "bmpc_k_accept"]      # (k_accept: builds before round 4)
    print("B      " + " ".join(f"{n[7:]:>11s}" for n in names) + "   (us, first super-step of the solve; k_eval workgroups)")
    for B, sv in zip(SIZES, solves):
        d = collections.defaultdict(list)
        for n, us, wg in sv:
            if not d[n]: d[n].append(us)          # the first launch of each kernel = the first super-step
        wg = max([w for n, _, w in sv if n in ("bmpc_k_eval", "bmpc_k_eval_curv")] or [0])
        print(f"{B:6d} " + " ".join(f"{(sum(d[n]) / len(d[n]) if d[n] else 0):11.1f}" for n in names) + f"   {wg}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--parse")
    a = ap.parse_args()
    parse(a.parse) if a.parse else run()
