"""GPU-box diagnostic (not a test): BASELINE configs[2] at full size, HIP path vs oracle, per block of the decision
vector, with the outliers' details.  Writes gpurun_out/full_parity.json.
    python tests/diag/diag_full_parity.py [B]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from boundplanner_amd import scenes  # noqa: E402
from boundplanner_amd.solver import HipBoundMPC  # noqa: E402

N = 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
be = HipBoundMPC(N)
t0 = time.time()
batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
print("gen", time.time() - t0, flush=True)
r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
t0 = time.time()
ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=0)
print("oracle", time.time() - t0, flush=True)
blocks = {"q": (0, 7 * N), "dq": (7 * N, 14 * N), "ddq": (14 * N, 21 * N), "u": (21 * N, 28 * N),
          "p": (28 * N, 34 * N), "v": (34 * N, 40 * N), "slacks": (40 * N, 44 * N + 6)}
out = {"B": B, "N": N}
out["status_pairs"] = {f"{a}{b}": int(((r["status"] == a) & (ro["status"] == b)).sum()) for a in range(4) for b in range(4)}
conv = (r["status"] == 0) & (ro["status"] == 0)
dit = r["iters"].astype(int) - ro["iters"].astype(int)
out["dit_hist"] = {str(k): int((dit[conv] == k).sum()) for k in np.unique(dit[conv])}
eq = conv & (dit == 0)
d = np.abs(r["x"] - ro["x"])
per = {k: d[:, a:b].max(axis=1) for k, (a, b) in blocks.items()}
df = np.abs(r["f"] - ro["f"]) / np.maximum(1.0, np.abs(ro["f"]))
q = lambda v: {"max": float(v.max()), "p999": float(np.quantile(v, 0.999)), "p99": float(np.quantile(v, 0.99)), "p50": float(np.median(v))}
out["same_iters"] = {k: q(v[eq]) for k, v in per.items()}
out["both_converged"] = {k: q(v[conv]) for k, v in per.items()}
out["df_rel"] = {"same_iters": q(df[eq]), "both_converged": q(df[conv])}
bars = {"q": 2e-3, "dq": 2e-3, "ddq": 2e-3, "u": 2e-2, "p": 2e-5, "v": 2e-5}
out["count_over_bar_same_iters"] = {k: int((per[k][eq] > bars[k]).sum()) for k in bars}
out["count_over_bar_both_converged"] = {k: int((per[k][conv] > bars[k]).sum()) for k in bars}
# outliers: the worst instances per block among the both-converged ones
worst = set()
for k in bars:
    idx = np.nonzero(conv)[0]
    worst |= set(idx[np.argsort(per[k][idx])[-4:]].tolist())
worst = sorted(worst)
# tight-tolerance solves of the outliers on both sides: same local solution or a branch?
bt = HipBoundMPC(N, tol=1e-8)
sel = np.array(worst)
rt = bt.solve_batch(batch["x0"][sel], batch["lbx"][sel], batch["ubx"][sel], batch["p"][sel])
rot = O.solve_batch(N, batch["x0"][sel], batch["lbx"][sel], batch["ubx"][sel], batch["p"][sel], tol=1e-8)
out["outliers"] = []
for j, i in enumerate(worst):
    dt = np.abs(rt["x"][j] - rot["x"][j])
    dg = np.abs(r["x"][i] - rt["x"][j]); do = np.abs(ro["x"][i] - rot["x"][j])
    out["outliers"].append({
        "i": int(i), "iters_gpu": int(r["iters"][i]), "iters_oracle": int(ro["iters"][i]), "f_gpu": float(r["f"][i]), "f_oracle": float(ro["f"][i]),
        "d": {k: float(per[k][i]) for k in per},
        "tight": {"status_gpu": int(rt["status"][j]), "status_oracle": int(rot["status"][j]), "iters_gpu": int(rt["iters"][j]),
                  "iters_oracle": int(rot["iters"][j]), "f_gpu": float(rt["f"][j]), "f_oracle": float(rot["f"][j]),
                  "d_gpu_vs_oracle": {k: float(dt[a:b].max()) for k, (a, b) in blocks.items()},
                  "d_gpu_tol5_vs_gpu_tol8": {k: float(dg[a:b].max()) for k, (a, b) in blocks.items()},
                  "d_or_tol5_vs_or_tol8": {k: float(do[a:b].max()) for k, (a, b) in blocks.items()}}})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_parity.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("status_pairs", "dit_hist", "same_iters", "both_converged", "df_rel", "count_over_bar_same_iters", "count_over_bar_both_converged")}, indent=1))
