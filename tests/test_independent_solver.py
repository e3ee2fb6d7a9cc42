"""SURVEY.md 8(c) bridge (ii) for the unpinned IPOPT boundary: an independent third-party NLP method
(scipy SLSQP, analytic derivatives from the golden-pinned NLP functions) started from the reference's
cold start reaches the same local solution as the interior-point oracle.  Tolerances are the survey's:
|dq| <= 1e-4 rad, |dp| <= 1e-4 m, df/f <= 1e-5."""
import os

import numpy as np
import pytest

import oracle_lib as O
from boundplanner_amd import scenes
from independent_nlp import slsqp_solve


def _agree(N, xo, fo, xs, fs, q_tol=1e-4, pv_tol=2e-5):
    d = np.abs(xo - xs)
    assert d[: 7 * N].max() < q_tol                      # q
    assert d[28 * N: 40 * N].max() < pv_tol               # task space p, v
    assert d[7 * N: 21 * N].max() < 10 * q_tol           # dq, ddq
    assert abs(fo - fs) <= 1e-5 * abs(fs)


@pytest.mark.parametrize("seed,rnd", [(6, False), (8192, True)])
def test_slsqp_reaches_the_oracle_solution_N6(seed, rnd):
    N = 6
    b = scenes.make_batch(1, N, seed, O.fk_batch, randomize_sets=rnd)
    x0, lbx, ubx, p = b["x0"][0], b["lbx"][0], b["ubx"][0], b["p"][0]
    r = O.solve(N, x0, lbx, ubx, p, tol=1e-8)
    assert r["status"] == 0
    s = slsqp_solve(N, x0, lbx, ubx, p)
    assert s.status == 0, s.message
    _agree(N, r["x"], r["f"], s.x, s.fun)


def test_oracle_matches_committed_slsqp_solutions_N10(golden_dir):
    d = np.load(os.path.join(golden_dir, "slsqp_N10.npz"))
    N = int(d["N"])
    for i in range(d["x"].shape[0]):
        r = O.solve(N, d["x0"][i], d["lbx"][i], d["ubx"][i], d["p"][i], tol=1e-8)
        assert r["status"] == 0
        _agree(N, r["x"], r["f"], d["x"][i], float(d["f"][i]))
        rd = O.solve(N, d["x0"][i], d["lbx"][i], d["ubx"][i], d["p"][i])     # the reference's tol 1e-5
        # at tol 1e-5 the null-space motion of the redundant arm is only resolved to the stated
        # joint-space bar of tests/test_gpu_parity.py (2e-3) and the early stop leaves ~1e-4 in task space
        # (distance of a tol-1e-5 interior-point iterate from the exact optimum, not a solver difference)
        _agree(N, rd["x"], rd["f"], d["x"][i], float(d["f"][i]), q_tol=2e-3, pv_tol=5e-4)
