"""Diagnostic (not a test): per-instance differences HIP vs oracle."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from boundplanner_amd import scenes
from boundplanner_amd.solver import HipBoundMPC
N, seed, rnd, B = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), int(sys.argv[4])
tol = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-5
be = HipBoundMPC(N, tol=tol)
batch = scenes.make_batch(B, N, seed, be.fk, randomize_sets=rnd)
r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
r_again = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
print("GPU run-to-run identical:", np.array_equal(r["x"], r_again["x"]), np.abs(r["x"] - r_again["x"]).max())
ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], tol=tol)
blocks = {"q": (0, 7 * N), "dq": (7 * N, 14 * N), "ddq": (14 * N, 21 * N), "u": (21 * N, 28 * N), "p": (28 * N, 34 * N), "v": (34 * N, 40 * N), "sl": (40 * N, 44 * N + 6)}
for i in range(B):
    d = np.abs(r["x"][i] - ro["x"][i])
    if d[: 21 * N].max() > 1e-6 or r["iters"][i] != ro["iters"][i]:
        print(i, "it", r["iters"][i], ro["iters"][i], "st", r["status"][i], ro["status"][i], "f", r["f"][i], ro["f"][i],
              " ".join(f"{k}:{d[a:b].max():.1e}" for k, (a, b) in blocks.items()))
np.savez(os.path.join(ROOT, "gpurun_out", f"diag_{N}_{seed}.npz"), xg=r["x"], xo=ro["x"], itg=r["iters"], ito=ro["iters"])
