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


# ---- the wider bridge: tests/golden/bridge_N*.npz (tests/golden/gen/gen_bridge.py) ----
BRIDGE = ["bridge_N10.npz", "bridge_N10_tc.npz", "bridge_N15.npz", "bridge_N20.npz", "bridge_N30.npz"]


def bridge_check(name, d, solve):
    """`solve(N, x0, lbx, ubx, p) -> (x, f, status)` at tol 1e-8 from the reference's cold start, against the independent
    scipy solutions of the fixture.  Per instance one of:
      same       the third-party method converged and both methods are in the same local solution: |dq| <= 1e-4 rad, task space
                 <= 2e-4 (SLSQP stalls at ~1e-4: its status-8 exits are within that of the interior-point point), df/f <= 1e-6;
      confirmed  they are in different local solutions of the non-convex NLP, and SLSQP restarted from the interior-point solution
                 stayed there (x_polish, moved < 1e-5): the independent method confirms the returned point as a local solution
                 (`confirmed_higher` counts those among them whose objective is the higher of the two: 1 of 12 at N=20);
      unconverged  the third-party method ran into its iteration limit (trust-constr, reported, not compared).
    Returns the counts."""
    N = int(d["N"][0])
    out = {"same": 0, "confirmed": 0, "confirmed_higher": 0, "unconverged": 0}
    tc = name.endswith("_tc.npz")
    for i in range(d["x"].shape[0]):
        if tc and int(d["status"][i]) == 0:           # trust-constr: 0 = iteration limit
            out["unconverged"] += 1
            continue
        x, f, st = solve(N, d["x0"][i], d["lbx"][i], d["ubx"][i], d["p"][i])
        assert st == 0, (name, i)
        dx = np.abs(x - d["x"][i])
        if dx[28 * N:40 * N].max() < 1e-3:
            assert dx[:7 * N].max() < 1e-4 and dx[28 * N:40 * N].max() < 2e-4, (name, i, dx[:7 * N].max(), dx[28 * N:40 * N].max())
            assert abs(f - float(d["f"][i])) <= 1e-6 * abs(f), (name, i)
            out["same"] += 1
        else:
            assert bool(d["has_polish"][i]), (name, i, "different local solutions and no confirmation run in the fixture")
            out["confirmed_higher"] += int(f >= float(d["f"][i]))                 # (which of the two local solutions is the better one)
            # the point the confirmation run started from (an interior-point solution at tol 1e-8 of an earlier build): two
            # tol-1e-8 points of one local solution are up to 3e-5 apart in the weakly determined joint-space directions
            # (tests/diag/diag_parity_tol.py), their objectives agree (next but one line)
            assert np.abs(x - d["x_ip"][i]).max() < 5e-5
            moved = np.abs(d["x_polish"][i] - d["x_ip"][i])
            assert moved[:7 * N].max() < 1e-5 and moved[28 * N:40 * N].max() < 1e-5, (name, i)
            assert abs(float(d["f_polish"][i]) - f) <= 1e-6 * abs(f)
            out["confirmed"] += 1
    return out


@pytest.mark.parametrize("name", BRIDGE)
def test_oracle_lands_on_the_independent_solutions(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    d = np.load(path)

    def solve(N, x0, lbx, ubx, p):
        r = O.solve(N, x0, lbx, ubx, p, tol=1e-8)
        return r["x"], r["f"], r["status"]
    counts = bridge_check(name, d, solve)
    print(name, counts)
    assert counts["same"] + counts["confirmed"] >= 0.75 * d["x"].shape[0]
