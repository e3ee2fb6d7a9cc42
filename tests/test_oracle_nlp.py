"""Oracle NLP functions vs golden vectors produced by the REFERENCE's own formulation code
(casadi_ocp_formulation.setup_optimization_problem run numerically, complex-step derivatives)."""
import os

import numpy as np
import pytest

import oracle_lib as O


def test_full_jacobian_N6(golden_dir):
    g = np.load(os.path.join(golden_dir, "nlp_N6.npz"))
    N = int(g["N"])
    for i in range(g["w"].shape[0]):
        f, gv, gr, J = O.nlp_eval(N, g["w"][i], g["p"][i])
        assert abs(f - g["f"][i]) <= 1e-12 * max(1, abs(g["f"][i]))
        assert np.abs(gv - g["g"][i]).max() < 1e-12
        assert np.abs(gr - g["grad_f"][i]).max() < 1e-11
        assert np.abs(J - g["jac_g"][i]).max() < 1e-12


@pytest.mark.parametrize("N", [10, 15, 20, 30])       # 15 = the reference's default horizon, 30 = BASELINE configs[4]
def test_directional_derivatives(golden_dir, N):
    g = np.load(os.path.join(golden_dir, f"nlp_N{N}.npz"))
    for i in range(g["w"].shape[0]):
        f, gv, gr, J = O.nlp_eval(N, g["w"][i], g["p"][i])
        assert abs(f - g["f"][i]) <= 1e-12 * max(1, abs(g["f"][i]))
        assert np.abs(gv - g["g"][i]).max() < 1e-12
        for r, df, dg in zip(g["r"][i], g["df"][i], g["dg"][i]):
            assert abs(gr @ r - df) < 1e-10 * max(1, abs(df))
            assert np.abs(J @ r - dg).max() < 1e-11


def test_constraint_bounds(golden_dir):
    g = np.load(os.path.join(golden_dir, "nlp_N6.npz"))
    lb, ub = O.gbounds(6)
    ref_lb = np.where(np.isinf(g["lbg"]), -1e20, g["lbg"])
    ref_ub = np.where(np.isinf(g["ubg"]), 1e20, g["ubg"])
    assert np.array_equal(lb, ref_lb) and np.array_equal(ub, ref_ub)
