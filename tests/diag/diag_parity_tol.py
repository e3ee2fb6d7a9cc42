"""GPU-box diagnostic (not a test): how tight must both sides be solved for the per-instance bars of tests/parity_lib.py (P1)?
    python tests/diag/diag_parity_tol.py [B] [N]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import oracle_lib as O  # noqa: E402
import parity_lib as PL  # noqa: E402
from boundplanner_amd import scenes  # noqa: E402
from boundplanner_amd.solver import HipBoundMPC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
be = HipBoundMPC(N)
batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
a = (batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
hip5 = be.solve_batch(*a); or5 = O.solve_batch(N, *a, nthreads=0)
for tol, mi in ((1e-8, 100), (1e-9, 100), (1e-10, 100), (1e-10, 200)):
    h = HipBoundMPC(N, tol=tol, max_iter=mi)
    hip8 = h.solve_batch(*a); h.close()
    or8 = O.solve_batch(N, *a, tol=tol, max_iter=mi, nthreads=0)
    rep = PL.compare(N, hip5, or5, hip8, or8)
    s = PL.summary(rep)
    c = rep["conv8"]
    q = lambda v: [float(np.quantile(v[c], x)) for x in (0.5, 0.9, 0.99, 1.0)]
    print(json.dumps({"tol": tol, "max_iter": mi, "tol_tight": s["tol1e-8"], "d_quantiles_50_90_99_100": {k: q(v) for k, v in rep["d8"].items()},
                      "df": q(rep["df8"]), "iters_mean": [float(hip8["iters"].mean()), float(or8["iters"].mean())]}), flush=True)
