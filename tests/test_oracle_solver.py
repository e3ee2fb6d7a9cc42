"""The oracle's interior-point solve: convergence + KKT conditions of the PINNED full-space NLP
(the function values/derivatives used in the check are the golden-validated ones)."""
import numpy as np
import pytest
from scipy.optimize import lsq_linear

import oracle_lib as O
from boundplanner_amd import scenes


def kkt_residual(N, x, p, lbx, ubx, act_tol=1e-3):
    """min over multipliers (right signs, zero on inactive rows) of |grad f + J^T lam_g + lam_x|_inf."""
    f, g, gr, J = O.nlp_eval(N, x, p)
    lbg, ubg = O.gbounds(N)
    cols, lo, hi = [], [], []
    for i in range(g.size):
        eq = lbg[i] == ubg[i]
        up = (not eq) and ubg[i] < 1e19 and g[i] > ubg[i] - act_tol
        dn = (not eq) and lbg[i] > -1e19 and g[i] < lbg[i] + act_tol
        if eq or up or dn:
            cols.append(J[i])
            lo.append(-np.inf if (eq or dn) else 0.0)
            hi.append(np.inf if (eq or up) else 0.0)
    for j in range(x.size):
        fixed = lbx[j] == ubx[j]
        up = (not fixed) and x[j] > ubx[j] - act_tol
        dn = (not fixed) and x[j] < lbx[j] + act_tol
        if fixed or up or dn:
            e = np.zeros(x.size); e[j] = 1.0
            cols.append(e)
            lo.append(-np.inf if (fixed or dn) else 0.0)
            hi.append(np.inf if (fixed or up) else 0.0)
    A = np.array(cols).T
    # (bvls: the active-set method; trf's iterations stall on some of these 270 x 240 problems -- status 0 with a residual of 1e4)
    res = lsq_linear(A, -gr, bounds=(np.array(lo), np.array(hi)), tol=1e-14, max_iter=2000, method="bvls")
    return np.abs(A @ res.x + gr).max(), g, lbg, ubg


@pytest.mark.parametrize("N,seed,rnd", [(6, 6, False), (10, 1024, False), (10, 8192, True)])
def test_solution_is_kkt_point(N, seed, rnd):
    b = scenes.make_batch(3, N, seed, O.fk_batch, randomize_sets=rnd)
    for i in range(3):
        r = O.solve(N, b["x0"][i], b["lbx"][i], b["ubx"][i], b["p"][i])
        assert r["status"] == 0 and r["iters"] < 60
        assert r["viol"] < 1e-4                      # the reference's own acceptance test (Q8)
        res, g, lbg, ubg = kkt_residual(N, r["x"], b["p"][i], b["lbx"][i], b["ubx"][i])
        assert res < 1e-5, res   # barrier multipliers mu/t live on rows within act_tol of their bound
        assert (g <= ubg + 1e-5).all() and (g >= lbg - 1e-5).all()
        assert (r["x"] <= b["ubx"][i] + 1e-6).all() and (r["x"] >= b["lbx"][i] - 1e-6).all()


def test_batch_statistics_N20():
    b = scenes.make_batch(16, 20, 8192, O.fk_batch, randomize_sets=True)
    r = O.solve_batch(20, b["x0"], b["lbx"], b["ubx"], b["p"], nthreads=4)
    assert (r["status"] == 0).all()
    assert r["iters"].mean() < 40
    assert r["viol"].max() < 1e-4


def test_second_order_correction_reaches_the_same_points():
    """The oracle-only second-order correction of the filter line search (Waechter & Biegler 2006, Sec. 2.4; measured without gain,
    DESIGN.md 2.2, so the kernels do not mirror it) changes the path, not the destination: same minimisers at a tight tolerance."""
    N = 10
    b = scenes.make_batch(12, N, 8192, O.fk_batch, randomize_sets=True)
    r0 = O.solve_batch(N, b["x0"], b["lbx"], b["ubx"], b["p"], nthreads=4, tol=1e-9, max_iter=200)
    r1 = O.solve_batch(N, b["x0"], b["lbx"], b["ubx"], b["p"], nthreads=4, tol=1e-9, max_iter=200, soc=4)
    ok = (r0["status"] == 0) & (r1["status"] == 0)
    assert ok.sum() >= 11
    assert (r0["iters"] != r1["iters"]).any()        # the correction was taken somewhere
    assert np.abs(r0["f"][ok] - r1["f"][ok]).max() < 1e-8 * max(1.0, np.abs(r0["f"][ok]).max())
    assert np.abs(r0["x"][ok][:, : 21 * N] - r1["x"][ok][:, : 21 * N]).max() < 1e-5


def test_warm_start_is_cheaper():
    N = 10
    b = scenes.make_batch(2, N, 1024, O.fk_batch)
    r0 = O.solve(N, b["x0"][0], b["lbx"][0], b["ubx"][0], b["p"][0])
    r1 = O.solve(N, r0["x"], b["lbx"][0], b["ubx"][0], b["p"][0])
    assert r1["status"] == 0 and r1["iters"] <= r0["iters"]
    assert np.abs(r1["x"][: 28 * N] - r0["x"][: 28 * N]).max() < 1e-3


def check_multipliers(N, x, p, lbx, ubx, lam_g, lam_x, res_tol, compl_tol):
    """The returned multipliers close the stationarity condition of the PINNED full-space NLP in CasADi's convention
    (grad f + J_g^T lam_g + lam_x = 0, BoundMPC.py:638-645), have the signs of their bounds and are complementary."""
    f, g, gr, J = O.nlp_eval(N, x, p)
    lbg, ubg = O.gbounds(N)
    res = gr + J.T @ lam_g + lam_x
    assert np.abs(res).max() < res_tol, (np.abs(res).max(), int(np.argmax(np.abs(res))))
    up = (ubg < 1e19) & (lbg < -1e19); lo = (lbg > -1e19) & (ubg > 1e19)
    assert (lam_g[up] >= 0).all() and (lam_g[lo] <= 0).all()
    assert np.abs(lam_g[up] * (g[up] - ubg[up])).max() < compl_tol and np.abs(lam_g[lo] * (g[lo] - lbg[lo])).max() < compl_tol
    free = lbx < ubx                         # not pinned
    xu = free & (ubx < 1e19); xl = free & (lbx > -1e19)
    both = xu & xl
    assert (lam_x[xu & ~xl] >= 0).all() and (lam_x[xl & ~xu] <= 0).all()
    cu = np.where(xu, lam_x * (x - np.where(xu, ubx, 0)), 0); cl = np.where(xl, lam_x * (x - np.where(xl, lbx, 0)), 0)
    # two-sided boxes: lam_x = z_ub - z_lb, each complementary with its own bound
    one = ~both
    assert np.abs(cu[one]).max() < compl_tol and np.abs(cl[one]).max() < compl_tol
    assert (lam_x[free & ~xu & ~xl] == 0).all()
    return np.abs(res).max()


@pytest.mark.parametrize("N,seed,rnd,tol,res_tol", [(6, 6, True, 1e-5, 1e-4), (10, 1024, False, 1e-5, 1e-4), (20, 8192, True, 1e-5, 1e-4),
                                                    (20, 8192, True, 1e-8, 1e-8), (15, 15, True, 1e-8, 1e-8)])
def test_multipliers_close_the_full_space_kkt_conditions(N, seed, rnd, tol, res_tol):
    """KKT residual of the pinned full-space NLP with the solver's own multipliers, up to the reference's horizons."""
    b = scenes.make_batch(3, N, seed, O.fk_batch, randomize_sets=rnd)
    for i in range(3):
        r = O.solve(N, b["x0"][i], b["lbx"][i], b["ubx"][i], b["p"][i], tol=tol)
        assert r["status"] == 0
        lbx = np.where(np.isinf(b["lbx"][i]), -1e20, b["lbx"][i]); ubx = np.where(np.isinf(b["ubx"][i]), 1e20, b["ubx"][i])
        check_multipliers(N, r["x"], b["p"][i], lbx, ubx, r["lam_g"], r["lam_x"], res_tol, 20 * tol)
