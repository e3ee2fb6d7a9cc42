"""Generates the per-instance files behind tests/golden/bridge_N{10,15,20}.npz: solutions of BASELINE configs[1]/[2]-style
instances by INDEPENDENT third-party NLP methods (scipy SLSQP; scipy trust-constr where it converges) from the
reference's cold start, on the pinned full-space NLP (tests/independent_nlp.py).  SURVEY.md 8(c) bridge (ii).

SLSQP is a dense active-set SQP: one instance takes ~2 min at N=10, ~8 min at N=15 and ~35 min at N=20 on one core, so
the instances are farmed out to worker processes and every finished instance is written at once:
    python tests/golden/gen/gen_bridge.py N first count [method]      # one worker: instances first..first+count-1
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
    if sys.argv[1] == "collect":
        for N in (10, 15, 20):
            files = sorted(glob.glob(os.path.join(OUT, f"N{N}_*.npz")))
            if not files:
                continue
            recs = [dict(np.load(f)) for f in files]
            keys = recs[0].keys()
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"bridge_N{N}.npz"),
                                **{k: np.array([r[k] for r in recs]) for k in keys})
            print(N, len(recs), "instances")
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
