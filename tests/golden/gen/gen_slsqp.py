"""Generates tests/golden/slsqp_N10.npz: SLSQP solutions (tests/independent_nlp.py) of two
configs[2]-style instances (N=10, randomized sets, seed 8192) from the reference's cold start.
Takes ~4 min; the N=6 cases are solved live in tests/test_independent_solver.py.
Run from the repo root:  python tests/golden/gen/gen_slsqp.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from boundplanner_amd import scenes  # noqa: E402
from independent_nlp import slsqp_solve  # noqa: E402

N, seed, B = 10, 8192, 2
b = scenes.make_batch(B, N, seed, O.fk_batch, randomize_sets=True)
xs, fs, nit = [], [], []
for i in range(B):
    s = slsqp_solve(N, b["x0"][i], b["lbx"][i], b["ubx"][i], b["p"][i])
    assert s.status == 0, s.message
    xs.append(s.x); fs.append(s.fun); nit.append(s.nit)
    print(i, s.nit, s.fun)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "slsqp_N10.npz"), N=N, seed=seed, x0=b["x0"], lbx=b["lbx"],
                    ubx=b["ubx"], p=b["p"], x=np.array(xs), f=np.array(fs), nit=np.array(nit))
