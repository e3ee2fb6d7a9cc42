"""Generates the per-instance files behind tests/golden/bridge_N{10,15,20,30}.npz: solutions of BASELINE configs[1]/[2]-style
instances by INDEPENDENT third-party NLP methods (scipy SLSQP; scipy trust-constr where it converges) from the
reference's cold start, on the pinned full-space NLP (tests/independent_nlp.py).  SURVEY.md 8(c) bridge (ii).

SLSQP is a dense active-set SQP: one instance takes ~2 min at N=10, ~8 min at N=15, ~35 min at N=20 and hours at N=30 on one core, so
the instances are farmed out to worker processes and every finished instance is written at once:
    python tests/golden/gen/gen_bridge.py N first count [method]      # one worker: instances first..first+count-1
    python tests/golden/gen/gen_bridge.py verify N                    # where SLSQP and the interior-point oracle ended in DIFFERENT local
                                                                      # solutions: SLSQP restarted from the interior-point solution (does the
                                                                      # independent method confirm it as a local solution?) -> *_polish.npz
    python tests/golden/gen/gen_bridge.py collect                     # merge what is there into tests/golden/bridge_N*.npz
Instance i of horizon N: even i -> configs[1] generator (fixed sets, seed 1024), odd i -> configs[2] generator (randomized
sets, seed 8192); both take row i // 2 of a 16-instance batch."""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from boundplanner_amd import scenes  # noqa: E402
from independent_nlp import slsqp_solve, trust_constr_solve  # noqa: E402

OUT = os.environ.get("BRIDGE_DIR", "/tmp/w/bridge")


def instance(N, i):
    rnd = bool(i % 2)
    b = scenes.make_batch(16, N, 8192 if rnd else 1024, O.fk_batch, randomize_sets=rnd)
    j = i // 2
    big = lambda a: np.nan_to_num(a, posinf=1e20, neginf=-1e20)
    return b["x0"][j], big(b["lbx"][j]), big(b["ubx"][j]), b["p"][j]


def main():
    if sys.argv[1] == "verify":
        N = int(sys.argv[2])
        for f in sorted(glob.glob(os.path.join(OUT, f"N{N}_*_slsqp.npz"))):
            d = np.load(f)
            out = f.replace("_slsqp.npz", "_polish.npz")
            if os.path.exists(out):
                continue
            r = O.solve(N, d["x0"], d["lbx"], d["ubx"], d["p"], tol=1e-8)
            if r["status"] != 0 or np.abs(r["x"] - d["x"])[28 * N:40 * N].max() < 1e-3:
                continue                       # same solution (or no interior-point solution to confirm)
            s = slsqp_solve(N, r["x"], d["lbx"], d["ubx"], d["p"], maxiter=100)
            np.savez(out, idx=int(d["idx"]), x_ip=r["x"], f_ip=r["f"], x_polish=s.x, f_polish=s.fun, status_polish=s.status, nit_polish=s.nit)
            print(N, int(d["idx"]), "SLSQP from the interior-point solution: status", s.status, "nit", s.nit, "f", s.fun, "vs", r["f"],
                  "moved q %.1e task %.1e" % (np.abs(s.x - r["x"])[:7 * N].max(), np.abs(s.x - r["x"])[28 * N:40 * N].max()), flush=True)
        return
    if sys.argv[1] == "collect":
        for N in (10, 15, 20, 30):
            for method in ("slsqp", "trust-constr"):
                files = sorted(glob.glob(os.path.join(OUT, f"N{N}_*_{method}.npz")))
                if not files:
                    continue
                recs = [dict(np.load(f)) for f in files]
                for r, f in zip(recs, files):
                    pf = f.replace(f"_{method}.npz", "_polish.npz")
                    pol = dict(np.load(pf)) if (method == "slsqp" and os.path.exists(pf)) else None
                    r["has_polish"] = pol is not None
                    r["x_polish"] = pol["x_polish"] if pol else np.full_like(r["x"], np.nan)
                    r["f_polish"] = float(pol["f_polish"]) if pol else np.nan
                    r["x_ip"] = pol["x_ip"] if pol else np.full_like(r["x"], np.nan)
                keys = [k for k in recs[0].keys() if k != "method"]
                tag = "" if method == "slsqp" else "_tc"
                np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"bridge_N{N}{tag}.npz"),
                                    **{k: np.array([r[k] for r in recs]) for k in keys})
                print(N, method, len(recs), "instances")
        return
    N, first, count = int(sys.argv[2 - 1]), int(sys.argv[2]), int(sys.argv[3])
    method = sys.argv[4] if len(sys.argv) > 4 else "slsqp"
    os.makedirs(OUT, exist_ok=True)
    for i in range(first, first + count):
        x0, lbx, ubx, p = instance(N, i)
        t0 = time.time()
        s = slsqp_solve(N, x0, lbx, ubx, p, maxiter=600) if method == "slsqp" else trust_constr_solve(N, x0, lbx, ubx, p)
        dt = time.time() - t0
        f, g, gr, J = O.nlp_eval(N, s.x, p, jac=False)
        lbg, ubg = O.gbounds(N)
        viol = max(0.0, (lbg - g).max(), (g - ubg).max(), (lbx - s.x).max(), (s.x - ubx).max())
        np.savez(os.path.join(OUT, f"N{N}_{i:02d}_{method}.npz"), N=N, idx=i, method=method, x0=x0, lbx=lbx, ubx=ubx, p=p, x=s.x,
                 f=s.fun, nit=s.nit, status=s.status, seconds=dt, max_viol=viol)
        print(N, i, method, "status", s.status, "nit", s.nit, "f", s.fun, "viol %.1e" % viol, "%.0f s" % dt, flush=True)


if __name__ == "__main__":
    main()
