"""Kernel logic of the HIP pipeline engine on the CPU: tests/emu/emu_pipe.cpp compiles the identical
device source (bmpc_stage.hpp, bmpc_pair_kernels.hpp, bmpc_ric_kernel.hpp) with 64/128 host threads per
workgroup standing in for the lanes.  The emulated pipeline must reproduce the oracle iterate for
iterate (same iteration counts, iterates to 1e-5) -- this pins the state machine, the record layout /
scatter table and every kernel body without a GPU.  (The -m gpu tests check the real kernels.)"""
import numpy as np

import emu_pipe_lib as E
import oracle_lib as O
from boundplanner_amd import scenes


def test_emulated_pipeline_matches_oracle():
    N, B = 6, 5
    batch = scenes.make_batch(B, N, 6, O.fk_batch, randomize_sets=True)
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=1)
    r = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True, want_lam=True)
    assert np.array_equal(r["status"], ro["status"]) and (r["status"] == 0).all()
    assert np.array_equal(r["iters"], ro["iters"])
    assert np.abs(r["x"] - ro["x"]).max() < 1e-5
    assert np.abs(r["f"] - ro["f"]).max() < 1e-9 * max(1.0, np.abs(ro["f"]).max())
    for i in range(B):
        _, g, _, _ = O.nlp_eval(N, r["x"][i], batch["p"][i], jac=False)
        assert np.abs(g - r["g"][i]).max() < 1e-9
        # multipliers (k_mult, k_mult_sweep): every entry written, equal to the oracle's, closing the full-space KKT conditions
        rs = O.solve(N, batch["x0"][i], batch["lbx"][i], batch["ubx"][i], batch["p"][i])
        assert np.isfinite(r["lam_g"][i]).all() and np.isfinite(r["lam_x"][i]).all()
        assert np.abs(r["lam_g"][i] - rs["lam_g"]).max() < 1e-5 and np.abs(r["lam_x"][i] - rs["lam_x"]).max() < 1e-5
        _, _, gr, J = O.nlp_eval(N, r["x"][i], batch["p"][i])
        assert np.abs(gr + J.T @ r["lam_g"][i] + r["lam_x"][i]).max() < 1e-4


def test_emulated_pool_streams_more_instances_than_slots():
    """bmpc_opts.pool_slots: 7 instances through a pool of 3 slots (a slot whose instance has finished is retired and takes
    the next input row) return bitwise what they return with a slot each."""
    N, B = 6, 7
    batch = scenes.make_batch(B, N, 9, O.fk_batch, randomize_sets=True)
    full = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    pool = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True, slots=3)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(full[k], pool[k]), k
    assert (full["status"] == 0).all() and len(set(full["iters"].tolist())) > 1      # they finish at different times


def test_emulated_pipeline_several_wavefronts_per_kernel():
    """N = 18: three instances per wavefront, 5 instances = one full and one ragged wavefront per thread-per-pair kernel (the
    emulator runs the workgroups of a launch one after the other on one LDS buffer, with a barrier between them), and the
    same through a pool of 2 slots."""
    N, B = 18, 5
    batch = scenes.make_batch(B, N, 18, O.fk_batch, randomize_sets=True)
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=1)
    r = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    assert np.array_equal(r["status"], ro["status"]) and np.array_equal(r["iters"], ro["iters"])
    assert np.abs(r["x"] - ro["x"]).max() < 1e-5
    pool = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True, slots=2)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(r[k], pool[k]), k


def test_emulated_field_major_layout_is_bitwise_the_same(monkeypatch):
    """BMPC_LAYOUT=0 (the field-major workspace of round 1, kept for A/B runs) against the slot-major default: the layouts
    only move data, so every output is bitwise identical."""
    N, B = 6, 5
    batch = scenes.make_batch(B, N, 6, O.fk_batch, randomize_sets=True)
    a = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    monkeypatch.setenv("BMPC_LAYOUT", "0")
    b = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(a[k], b[k]), k


def test_emulated_backtracking_inside_k_trial_is_scheduling_only():
    """bmpc_opts.trial_repeats: k_trial repeats a rejected trial itself (half the step length, up to nine times: the whole line
    search in one super-step) instead of handing the instance to the next super-step's trial list (0, rounds 1-2).  Same
    trials, same results bitwise, fewer super-steps."""
    N, B = 18, 5
    batch = scenes.make_batch(B, N, 18, O.fk_batch, randomize_sets=True)
    r0 = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True, trial_repeats=0)
    r9 = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True, trial_repeats=9)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(r0[k], r9[k]), k
    assert r9["steps"] < r0["steps"]          # some trial was rejected: the backtracking instance waited for nobody


def test_emulated_iterates_equal_the_oracles_iterate_for_iterate():
    """The CPU-side counterpart of tests/test_iterate_parity.py (which runs the real kernels): the emulated device source and the
    oracle stopped after k = 1, 2, 4, 8, 12 iterations are at the same point -- 1e-9 relative, per block -- and the oracle reports the
    decisions of its last iteration consistently (bmpc_oracle_solve_batch_info).  Exact-Hessian iterations (from k ~ 8 on) included:
    this is the comparison that exposed stale multipliers in the oracle's Hessian assembly in round 4."""
    import iterate_parity_lib as IP
    N, B = 10, 4
    batch = scenes.make_batch(B, N, 1024, O.fk_batch, randomize_sets=True)
    a = (batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    hess_seen = False
    for k in (1, 2, 4, 8, 12):
        r = E.solve_batch(N, *a, max_iter=k)
        o = O.solve_batch_info(N, *a, nthreads=1, max_iter=k)
        assert np.array_equal(r["iters"], o["iters"]) and np.array_equal(r["status"], o["status"])
        assert np.array_equal(o["info"][:, 0], o["iters"]) and np.array_equal(o["info"][:, 1], o["status"])
        running = o["iters"] == k
        rd = IP.rel_diff(N, r["x"], o["x"])
        assert rd[running].max() <= (1e-12 if k == 1 else 1e-9), (k, rd)
        hess_seen = hess_seen or bool((o["info"][running, 7] == 1).any())
    assert hess_seen          # some instance has switched to the exact Hessian within twelve iterations


def test_emulated_four_wavefront_trial_kernel_agrees_with_the_default():
    """-DBMPC_TRIAL_NW=4 (k_trial as a workgroup of four wavefronts, each one part of the row walk, partial sums combined through
    LDS in part order): kept behind a build knob (EXPERIMENTS.md, slower on the GPU).  Same iteration counts, iterates equal to
    rounding (the sums of theta and of log t are taken per part first)."""
    N, B = 6, 5
    batch = scenes.make_batch(B, N, 6, O.fk_batch, randomize_sets=True)
    a = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    b = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], variant="trial4")
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"])
    assert np.abs(a["x"] - b["x"]).max() < 1e-8


def test_emulated_speculative_factorisation_attempts_are_bitwise_the_sequential_ones(monkeypatch):
    """bmpc_k_ric_att + bmpc_k_ric_sel (the deep tail's Riccati kernels on the GPU): the attempts of an iteration -- delta_w = 0, then
    the Gauss-Newton fallback or escalating delta_w -- run as workgroups of their own, the selection kernel takes the first one that
    succeeded and only sweeps itself beyond the speculated ones.  Every attempt is the arithmetic the sequential loop would have done:
    the results are bitwise the same, here on instances that need inertia corrections (iteration counts up to the limit)."""
    N = 15
    batch = scenes.make_batch(128, N, 7, O.fk_batch, randomize_sets=True)
    pick = [49, 108, 86]                         # the instances of this batch with the most factorisation retries
    a = tuple(batch[k][pick] for k in ("x0", "lbx", "ubx", "p"))
    seq = E.solve_batch(N, *a, want_g=True, max_iter=30)
    monkeypatch.setenv("BMPC_EMU_RIC_SPEC", "1")
    spec = E.solve_batch(N, *a, want_g=True, max_iter=30)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(seq[k], spec[k]), k


def test_emulated_speculative_line_search_is_bitwise_the_sequential_one(monkeypatch):
    """bmpc_k_trial_spec (the tail regime's line search on the GPU): the step lengths of a search -- alpha, alpha / 2, ... -- are
    tried four at a time by a workgroup of four wavefronts, the tests run in the order of the sequential search and the first
    accepted candidate is copied over.  Same trial points, same tests: bitwise the results of k_trial's own backtracking, here on
    instances whose searches do backtrack (the three above and one more: up to the iteration limit)."""
    N = 15
    batch = scenes.make_batch(128, N, 7, O.fk_batch, randomize_sets=True)
    pick = [49, 108, 86, 74]
    a = tuple(batch[k][pick] for k in ("x0", "lbx", "ubx", "p"))
    seq = E.solve_batch(N, *a, want_g=True, max_iter=30)
    monkeypatch.setenv("BMPC_EMU_TRIAL_SPEC", "1")
    spec = E.solve_batch(N, *a, want_g=True, max_iter=30)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(seq[k], spec[k]), k
    assert seq["iters"].max() >= 30          # (the searches of these instances do backtrack: they run into the limit)


def test_emulated_two_wavefront_k_eval_is_bitwise_the_one_wavefront_kernel(monkeypatch):
    """bmpc_k_eval_curv_split (the tail regime's evaluation on the GPU): k_eval as two bodies -- everything but the chained (q, dq, pi)
    block, with a hole in its record stores / that block alone, straight into the record.  Same expressions entry for entry: bitwise
    the results of the one-wavefront kernel (here the chain part runs first: the other part must not touch its fields)."""
    N, B = 6, 5
    batch = scenes.make_batch(B, N, 6, O.fk_batch, randomize_sets=True)
    a = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    monkeypatch.setenv("BMPC_EMU_EVAL_SPLIT", "1")
    b = E.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    for k in ("x", "g", "f", "iters", "status", "viol"):
        assert np.array_equal(a[k], b[k]), k
