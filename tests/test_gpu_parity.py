"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backends():
    from boundplanner_amd.solver import HipBoundMPC
    cache = {}

    def get(N, **kw):
        key = (N, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = HipBoundMPC(N, **kw)
        return cache[key]
    return get


def test_fk_matches_oracle(backends):
    be = backends(10)
    rng = np.random.default_rng(0)
    q = rng.uniform(-2, 2, size=(257, 7)); dq = rng.normal(size=(257, 7))
    a, b = be.fk(q, dq), O.fk_batch(q, dq)
    for k in a:
        assert np.abs(a[k] - b[k]).max() < 1e-12, k


def test_fk_matches_reference_tapes(backends, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "kin.npz"))
    out = backends(10).fk(g["q"], g["dq"])
    assert np.abs(out["ee_pos"] - g["fk_pos"]).max() < 1e-12
    assert np.abs(out["col_pts"] - g["fk_pos_col"]).max() < 1e-12
    assert np.abs(out["jac"] - g["jacobian"]).max() < 1e-12
    assert np.abs(out["ee_rot"] - g["hom_trans"][:, :3, :3]).max() < 1e-12
    assert np.abs(out["dvdq"] - g["dvdq"]).max() < 1e-11


# Tolerance.  Both sides run the same algorithm in FP64; differences come from summation order and
# libm only.  The NLP itself is ill-conditioned in two ways that are properties of the reference
# formulation, not of either implementation: the 7-DOF arm is redundant for the 6-D pose task
# (joint-space motion in the null space costs only the 1e-3 / 1e-4 velocity and jerk weights), and
# the jerk weight 2e-4 sits against barrier terms up to 1e8.  Hence the stated bars at tol = 1e-5:
#   task space  p (m, rad), v        |d|_inf <= 2e-5        objective |df| <= 1e-6 relative
#   joint space q, dq, ddq           |d|_inf <= 2e-3        jerk u   |d|_inf <= 2e-2 (bound 35)
# with identical iteration counts (+-1 on a few instances), and at tol = 1e-8 agreement to 1e-5 in
# joint space / 1e-7 in task space.
@pytest.mark.parametrize("N,seed,rnd,B", [(6, 6, True, 16), (10, 1024, False, 64), (20, 8192, True, 48),
                                          (15, 15, True, 24),      # the reference's default horizon (util_functions.py:49)
                                          (30, 4096, False, 24),   # configs[4]
                                          (3, 3, True, 8)])        # shortest horizon the handle accepts
def test_solve_matches_oracle(backends, N, seed, rnd, B):
    from boundplanner_amd import scenes
    be = backends(N)
    batch = scenes.make_batch(B, N, seed, be.fk, randomize_sets=rnd)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=0)
    same = (r["status"] == ro["status"])
    conv = (r["status"] == 0) & (ro["status"] == 0)
    dit = np.abs(r["iters"] - ro["iters"])
    eq = conv & (dit == 0)
    print(f"N={N}: status equal {same.sum()}/{B}, both converged {conv.sum()}, same iterations {eq.sum()}, max |d iters| {dit[conv].max()}")
    assert same.all()                              # every instance ends in the same status
    assert conv.sum() >= B - 1                     # (N=30: one instance of the 24 runs into max_iter on both sides)
    assert dit[conv].max() <= 1
    assert eq.sum() >= conv.sum() - 2              # at most two instances (N=30: 2 of 24) stop an iteration apart
    blk = lambda a, lo, hi: np.abs(a["x"][:, lo * N:hi * N])
    d = lambda lo, hi: np.abs(r["x"][:, lo * N:hi * N] - ro["x"][:, lo * N:hi * N]).max(axis=1)
    d_task, d_joint, d_u = d(28, 40), d(0, 21), d(21, 28)
    print(f"N={N}: same iters {eq.sum()}/{conv.sum()} task {d_task[eq].max():.1e} joint {d_joint[eq].max():.1e} u {d_u[eq].max():.1e}")
    assert d_task[eq].max() < 2e-5 and d_joint[eq].max() < 2e-3 and d_u[eq].max() < 2e-2
    # one side stopped one Newton iteration earlier (KKT error within rounding of tol): both points
    # pass the same optimality test and differ by the size of that last step
    assert d_task[conv].max() < 1e-3
    assert np.abs(r["f"][conv] - ro["f"][conv]).max() < 1e-6 * max(1.0, np.abs(ro["f"][conv]).max())
    assert np.abs(r["viol"][conv] - ro["viol"][conv]).max() < 1e-8
    # g returned by the kernel == the pinned full-space g evaluated at the returned x
    for i in np.nonzero(conv)[0][:8]:
        _, g, _, _ = O.nlp_eval(N, r["x"][i], batch["p"][i], jac=False)
        assert np.abs(g - r["g"][i]).max() < 1e-9


def test_split_index_variants_and_slacks0(backends):
    from boundplanner_amd import scenes
    N, B = 10, 12
    be = backends(N)
    batch = scenes.make_batch(B, N, 77, be.fk, randomize_sets=True)
    p = batch["p"].copy()
    p[0::3, 0:5] = [0, 3, N, N, N]
    p[1::3, 0:5] = [0, 2, 5, N, N]
    p[:, 5:11] = [0.01, 0, 0.02, 0, 0, 0.005]
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], p)
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], p)
    conv = (r["status"] == 0) & (ro["status"] == 0)
    assert conv.sum() >= B - 2
    eq = conv & (r["iters"] == ro["iters"])
    assert eq.sum() >= conv.sum() - 2
    assert np.abs(r["x"][eq][:, :21 * N] - ro["x"][eq][:, :21 * N]).max() < 2e-3
    assert np.abs(r["x"][eq][:, 28 * N:40 * N] - ro["x"][eq][:, 28 * N:40 * N]).max() < 2e-5


def test_tight_tolerance_agreement(backends):
    """At tol = 1e-8 (barrier floor 1e-9) the HIP path and the oracle land on the same point."""
    from boundplanner_amd import scenes
    N, B = 10, 24
    be = backends(N, tol=1e-8)
    batch = scenes.make_batch(B, N, 31, be.fk, randomize_sets=True)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_lam=True)
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], tol=1e-8)
    conv = (r["status"] == 0) & (ro["status"] == 0)
    assert conv.mean() > 0.9
    same = conv & (r["iters"] == ro["iters"])
    dj = np.abs(r["x"][:, :21 * N] - ro["x"][:, :21 * N]).max(axis=1)
    dt_ = np.abs(r["x"][:, 28 * N:40 * N] - ro["x"][:, 28 * N:40 * N]).max(axis=1)
    df = np.abs(r["f"] - ro["f"]) / np.maximum(1.0, np.abs(ro["f"]))
    w = int(np.argmax(np.where(conv, dj, 0)))
    print(f"tol 1e-8: same iterations {same.sum()}/{conv.sum()}, joint {dj[same].max():.1e} / {dj[conv].max():.1e}, task {dt_[same].max():.1e} / {dt_[conv].max():.1e}, "
          f"df {df[conv].max():.1e}; worst instance {w}: iters {r['iters'][w]} / {ro['iters'][w]}, df {df[w]:.1e}, "
          f"second worst joint {np.sort(dj[conv])[-2]:.1e}")
    assert same.sum() >= conv.sum() - 2
    close = conv & (dj < 1e-5) & (dt_ < 1e-7)
    assert close.sum() >= conv.sum() - 2          # observed: 23 of 24 to <= 6e-9 in joint space
    # the others (one side a Newton step further, or a borderline inertia decision taken differently on rounding): the
    # null-space motion of the redundant arm is priced by weights of 1e-3 / 1e-4 only, so both points can pass the 1e-8
    # test ~1e-5 apart in joint space; they must be the same solution by objective and the HIP point a KKT point of the
    # pinned NLP with the multipliers it returns
    assert dj[conv].max() < 1e-4 and dt_[conv].max() < 1e-5 and df[conv].max() < 1e-7
    for i in np.nonzero(conv & ~close)[0]:
        _, _, gr, J = O.nlp_eval(N, r["x"][i], batch["p"][i])
        assert np.abs(gr + J.T @ r["lam_g"][i] + r["lam_x"][i]).max() < 1e-7, i


def test_round_trip_properties_full_size(backends):
    """Size-independent properties at a BASELINE size (N=20): returned points satisfy the pinned
    constraints, re-solving from the solution is a fixed point, and a batch equals its halves."""
    from boundplanner_amd import scenes
    N, B = 20, 256
    be = backends(N)
    batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    ok = r["status"] == 0
    assert ok.mean() > 0.95
    lbg, ubg = be.lbg, be.ubg
    assert (r["g"][ok] <= ubg + 1e-4).all() and (r["g"][ok] >= lbg - 1e-4).all()
    assert r["viol"][ok].max() < 1e-4
    r2 = be.solve_batch(r["x"], batch["lbx"], batch["ubx"], batch["p"])
    ok2 = ok & (r2["status"] == 0)
    assert (r2["iters"][ok2] <= r["iters"][ok2]).mean() > 0.8
    # re-solving from a solution returns to it (task space; a few instances may leave for another
    # local solution because slacks/multipliers are re-initialised, Q11)
    dtask = np.abs(r2["x"][:, 28 * N:40 * N] - r["x"][:, 28 * N:40 * N]).max(axis=1)
    assert (dtask[ok2] < 1e-3).mean() > 0.9
    h = B // 2
    ra = be.solve_batch(batch["x0"][:h], batch["lbx"][:h], batch["ubx"][:h], batch["p"][:h])
    assert np.array_equal(ra["x"], r["x"][:h])      # instance results do not depend on the batch


def test_nlpsolver_object_matches_reference_call_convention(backends):
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipNlpSolver
    N = 10
    be = backends(N)
    batch = scenes.make_batch(1, N, 5, be.fk)
    s = HipNlpSolver(N, backend=be)
    sol = s(x0=batch["x0"][0], lbx=batch["lbx"][0], ubx=batch["ubx"][0], lbg=s.lbg, ubg=s.ubg, p=batch["p"][0])
    st = s.stats()
    assert st["success"] and st["iter_count"] > 0 and st["return_status"] == "Solve_Succeeded"
    assert sol["x"].full().shape == (44 * N + 6, 1) and sol["g"].full().shape == (147 * (N - 1) + 21, 1)
    lam_g, lam_x = sol["lam_g"].full().ravel(), sol["lam_x"].full().ravel()
    assert lam_g.shape == (147 * (N - 1) + 21,) and lam_x.shape == (44 * N + 6,) and np.abs(lam_g).max() > 0
    _, _, gr, J = O.nlp_eval(N, sol["x"].full().ravel(), batch["p"][0])
    assert np.abs(gr + J.T @ lam_g + lam_x).max() < 1e-4


def test_engines_and_async_entry_agree(backends):
    """The pipeline engine (default), the persistent one-wavefront-per-instance engine and the asynchronous
    device-pointer entry solve the same batch: identical results between the synchronous and the asynchronous
    entry of one engine (bitwise), same iteration counts and iterates within the stated tolerance between engines."""
    import torch
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 10, 96
    be0 = backends(N)
    be1 = backends(N, engine=1)
    batch = scenes.make_batch(B, N, 77, be0.fk, randomize_sets=True)
    r0 = be0.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    r1 = be1.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    conv = (r0["status"] == 0) & (r1["status"] == 0)
    assert conv.mean() > 0.95
    assert np.abs(r0["iters"][conv] - r1["iters"][conv]).max() <= 1
    same = conv & (r0["iters"] == r1["iters"])
    assert np.abs(r0["x"][same][:, 28 * N:40 * N] - r1["x"][same][:, 28 * N:40 * N]).max() < 2e-5
    # asynchronous entry, device pointers
    dev = torch.device("cuda", 0)
    big = lambda a: np.nan_to_num(a, posinf=1e20, neginf=-1e20)
    d = {k: torch.from_numpy(big(batch[k])).to(dev) for k in ("x0", "lbx", "ubx", "p")}
    x = torch.empty((B, be0.n_w), dtype=torch.float64, device=dev)
    f = torch.empty(B, dtype=torch.float64, device=dev); viol = torch.empty(B, dtype=torch.float64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    be2 = HipBoundMPC(N, max_batch=B)
    be2.solve_dev_async(B, d["x0"].data_ptr(), d["lbx"].data_ptr(), d["ubx"].data_ptr(), d["p"].data_ptr(), x.data_ptr(),
                        f.data_ptr(), it.data_ptr(), st.data_ptr(), viol.data_ptr())
    assert be2.active() >= 0
    be2.wait()
    assert be2.active() == 0
    assert np.array_equal(x.cpu().numpy(), r0["x"]) and np.array_equal(it.cpu().numpy(), r0["iters"])
    assert np.array_equal(st.cpu().numpy(), r0["status"]) and np.array_equal(viol.cpu().numpy(), r0["viol"])


def test_full_size_batch_properties(backends):
    """BASELINE configs[2] at full size (8192 instances, N=20, randomized convex sets): properties that do
    not need the oracle.  (1) an instance's result does not depend on the batch it is solved in: the first
    512 instances solved alone are bitwise equal to their rows in the full batch (what makes sharding over
    GPUs exact); (2) every accepted solution satisfies the reference's acceptance test and its returned g
    matches the box structure of lbg/ubg; (3) repeated solves are bitwise reproducible."""
    from boundplanner_amd import scenes
    N, B = 20, 8192
    be = backends(N)
    batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    r2 = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    for k in ("x", "f", "iters", "status", "viol"):
        assert np.array_equal(r[k], r2[k]), k
    sub = be.solve_batch(batch["x0"][:512], batch["lbx"][:512], batch["ubx"][:512], batch["p"][:512], want_g=True)
    for k in ("x", "f", "iters", "status", "viol"):
        assert np.array_equal(sub[k], r[k][:512]), k
    ok = (r["status"] == 0) | (r["viol"] < 1e-4)
    assert ok.mean() > 0.99 and (r["status"] == 0).mean() > 0.99
    assert r["iters"][r["status"] == 0].max() <= 100 and r["iters"].mean() < 30
    conv = sub["status"] == 0
    g = sub["g"][conv]
    assert (g <= be.ubg + 1e-4).all() and (g >= be.lbg - 1e-4).all()
    # stage-0 pins are returned exactly
    x = r["x"]
    for blk in range(4):
        assert np.array_equal(x[:, blk * 7 * N:(blk + 1) * 7 * N:N], batch["lbx"][:, blk * 7 * N:(blk + 1) * 7 * N:N])


BLOCKS = lambda N: {"q": (0, 7 * N), "dq": (7 * N, 14 * N), "ddq": (14 * N, 21 * N), "u": (21 * N, 28 * N),
                    "p": (28 * N, 34 * N), "v": (34 * N, 40 * N), "slacks": (40 * N, 44 * N + 6)}


def test_config2_full_batch_against_oracle(backends):
    """BASELINE configs[2] at its full size (8192 instances, N=20, randomized convex sets), EVERY instance against the oracle,
    per block of the decision vector.  Both sides run the same algorithm in FP64 and differ by summation order only; what
    that rounding noise turns into depends on the instance:
      * 97.9 % of the instances that converge on both sides take the same number of iterations; they agree to the bars
        below (worst observed: q 3.7e-4, u 2.9e-3, task 2.8e-4, objective 2.1e-7 relative).  The task-space bar of
        DESIGN.md (2e-5) holds for 99.7 % of them; the rest are long, ill-conditioned runs (40-90 iterations) in which
        the noise is amplified by every line-search decision.
      * the others stop an iteration (or, for the stragglers, up to 30 iterations) apart: both points pass the same
        optimality test and differ by the size of the last Newton steps.  At tol = 1e-5 an accepted iterate is itself
        up to 1e-3 rad / 9e-2 (jerk) / 2e-3 (task) away from the exact KKT point -- measured as the distance between the
        tol = 1e-5 and tol = 1e-8 solutions of the same solver, tests/diag/diag_full_parity.py -- so that is the size of
        the differences here (worst observed: jerk 2.6e-2 at instance 7445, 66 vs 67 iterations; at tol = 1e-8 the two
        sides agree on it to 2.4e-5).  Objective values agree to 4.4e-6 relative on every instance.
    Bars are the observed maxima with a factor ~2 of head room."""
    from boundplanner_amd import scenes
    N, B = 20, 8192
    be = backends(N)
    batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=0)
    n_status_diff = int((r["status"] != ro["status"]).sum())
    conv = (r["status"] == 0) & (ro["status"] == 0)
    dit = np.abs(r["iters"].astype(int) - ro["iters"].astype(int))
    eq = conv & (dit == 0)
    d = np.abs(r["x"] - ro["x"])
    per = {k: d[:, a:b].max(axis=1) for k, (a, b) in BLOCKS(N).items()}
    df = np.abs(r["f"] - ro["f"]) / np.maximum(1.0, np.abs(ro["f"]))
    worst = sorted(set(int(np.nonzero(conv)[0][np.argmax(v[conv])]) for v in per.values()))
    print(f"configs[2]: status differs on {n_status_diff}, both converged {conv.sum()}/{B}, same iterations {eq.sum()}, "
          f"|d iters| <= 1: {(dit[conv] <= 1).sum()}, max {dit[conv].max()}")
    print("  same-iters max   ", {k: f"{v[eq].max():.1e}" for k, v in per.items()}, f"df {df[eq].max():.1e}")
    print("  both-conv  max   ", {k: f"{v[conv].max():.1e}" for k, v in per.items()}, f"df {df[conv].max():.1e}")
    print("  worst instances  ", [(i, int(r["iters"][i]), int(ro["iters"][i])) for i in worst])
    assert n_status_diff <= 16                                   # observed 8 (4 + 4 at the max_iter edge)
    assert conv.sum() >= 0.995 * B                               # observed 8168
    # the largest gap is a single straggler that wanders differently on the two sides (45 and 52 iterations apart in two
    # builds of round 2 that differ in a summation order): bounded by a quantile, not by its maximum
    print(f"  |d iters| <= 5: {(dit[conv] <= 5).sum()}, <= 20: {(dit[conv] <= 20).sum()}")
    assert eq.sum() >= 0.97 * conv.sum() and (dit[conv] <= 1).sum() >= 0.985 * conv.sum()
    assert (dit[conv] <= 5).sum() >= 0.993 * conv.sum() and (dit[conv] <= 20).sum() >= 0.998 * conv.sum()
    # u (jerk, weight 2e-4 against barrier terms up to 1e8) is the weakly determined block: the largest same-iteration gap moved
    # between 3.4e-3, 9.6e-3 and 2.5e-2 (one instance) over builds of round 2 that differ only in summation order
    bars_same = {"q": 1e-3, "dq": 1e-3, "ddq": 2e-3, "u": 4e-2, "p": 5e-4, "v": 5e-4, "slacks": 1e-5}
    bars_conv = {"q": 2e-3, "dq": 3e-3, "ddq": 8e-3, "u": 6e-2, "p": 2e-3, "v": 2e-3, "slacks": 2e-4}
    for k in per:
        assert per[k][eq].max() < bars_same[k], (k, int(np.argmax(np.where(eq, per[k], 0))))
        assert per[k][conv].max() < bars_conv[k], (k, int(np.argmax(np.where(conv, per[k], 0))))
    assert df[eq].max() < 1e-6 and df[conv].max() < 1e-5
    # the stated tight bars hold for all but a few per mille of the same-iteration instances
    tight = {"q": 2e-3, "dq": 2e-3, "ddq": 2e-3, "u": 2e-2, "p": 2e-5, "v": 2e-5}
    for k, bar in tight.items():
        assert (per[k][eq] > bar).sum() <= (0 if k in ("q", "dq", "ddq") else 3 if k == "u" else 0.006 * eq.sum()), k
        assert np.quantile(per[k][eq], 0.99) < 0.5 * bar, k
    # sum of violations beyond the reference's 1e-6 dead band (BoundMPC.py:613-615): observed 5.3e-5 apart at most
    assert np.abs(r["viol"][conv] - ro["viol"][conv]).max() < 2e-4


@pytest.mark.parametrize("N,seed,rnd,tol,res_tol", [(10, 1024, False, 1e-5, 1e-4), (20, 8192, True, 1e-5, 1e-4), (15, 15, True, 1e-8, 1e-7)])
def test_multipliers(backends, N, seed, rnd, tol, res_tol):
    """sol["lam_g"], sol["lam_x"] (BoundMPC.py:638-645) through the C ABI: they close the KKT conditions of the pinned
    full-space NLP at the returned point, carry CasADi's signs, are complementary, and equal the oracle's."""
    from boundplanner_amd import scenes
    from test_oracle_solver import check_multipliers
    B = 16
    be = backends(N, tol=tol)
    batch = scenes.make_batch(B, N, seed, be.fk, randomize_sets=rnd)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_lam=True)
    ro = [O.solve(N, batch["x0"][i], batch["lbx"][i], batch["ubx"][i], batch["p"][i], tol=tol) for i in range(B)]
    assert (r["status"] == 0).sum() >= B - 1
    worst = 0.0
    for i in np.nonzero(r["status"] == 0)[0]:
        lbx = np.where(np.isinf(batch["lbx"][i]), -1e20, batch["lbx"][i]); ubx = np.where(np.isinf(batch["ubx"][i]), 1e20, batch["ubx"][i])
        worst = max(worst, check_multipliers(N, r["x"][i], batch["p"][i], lbx, ubx, r["lam_g"][i], r["lam_x"][i], res_tol, 20 * tol))
        if ro[i]["status"] == 0 and ro[i]["iters"] == r["iters"][i]:
            sc = max(1.0, np.abs(ro[i]["lam_g"]).max())
            assert np.abs(r["lam_g"][i] - ro[i]["lam_g"]).max() < 1e-3 * sc and np.abs(r["lam_x"][i] - ro[i]["lam_x"]).max() < 1e-3 * sc
    print(f"N={N} tol={tol}: max stationarity residual {worst:.1e}")
    # device-pointer entry: multipliers of the most recent solve on the handle
    import torch
    dev = torch.device("cuda", 0)
    lg = torch.empty((B, be.n_g), dtype=torch.float64, device=dev); lx = torch.empty((B, be.n_w), dtype=torch.float64, device=dev)
    be.multipliers_dev(B, lg.data_ptr(), lx.data_ptr())
    assert np.array_equal(lg.cpu().numpy(), r["lam_g"]) and np.array_equal(lx.cpu().numpy(), r["lam_x"])
    with pytest.raises(RuntimeError):
        be.multipliers_dev(B + 1, lg.data_ptr(), lx.data_ptr())       # no solve of that size on the handle


def test_pool_streams_a_batch_through_fewer_slots(backends):
    """bmpc_opts.pool_slots: 3000 instances through a pool of 512 / 1000 slots (a slot whose instance has finished is
    retired and takes the next input row) return bitwise what they return with a slot each; multipliers, which need
    every final iterate in the workspace, are refused for a streamed call."""
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 10, 3000
    be = backends(N)
    batch = scenes.make_batch(B, N, 1024, be.fk, randomize_sets=True)
    full = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    for slots in (512, 1000):
        pool = HipBoundMPC(N, pool_slots=slots)
        r = pool.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
        for k in ("x", "g", "f", "iters", "status", "viol"):
            assert np.array_equal(full[k], r[k]), (slots, k)
        with pytest.raises(RuntimeError):
            pool.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_lam=True)
        small = pool.solve_batch(batch["x0"][:300], batch["lbx"][:300], batch["ubx"][:300], batch["p"][:300], want_lam=True)
        assert np.array_equal(small["x"], full["x"][:300]) and np.isfinite(small["lam_g"]).all()
        pool.close()


def test_pool_at_bench_horizon_is_bitwise_the_same(backends):
    """The bench schedule in small: N = 20, 6000 instances streamed through a 2048-slot pool (every slot is reused about three
    times, at different moments of its predecessors' solves) return bitwise what one slot per instance returns."""
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 20, 6000
    be = backends(N)
    batch = scenes.make_batch(B, N, [8192, 77], be.fk, randomize_sets=True)
    full = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    pool = HipBoundMPC(N, pool_slots=2048)
    r = pool.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    pool.close()
    for k in ("x", "f", "iters", "status", "viol"):
        assert np.array_equal(full[k], r[k]), k
    assert (full["status"] == 0).mean() > 0.99 and full["iters"].max() > 60      # stragglers included


def test_field_major_layout_is_bitwise_the_same(monkeypatch):
    """BMPC_LAYOUT=0 selects the field-major workspace of round 1 (kept for A/B measurements) when a handle is created; it
    only moves data: outputs are bitwise those of the slot-major default, also through a streaming pool."""
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 10, 700
    ref = HipBoundMPC(N)
    batch = scenes.make_batch(B, N, 2048, ref.fk, randomize_sets=True)
    a = ref.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
    monkeypatch.setenv("BMPC_LAYOUT", "0")
    for kw in ({}, {"pool_slots": 256}):
        other = HipBoundMPC(N, **kw)
        b = other.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"], want_g=True)
        for k in ("x", "g", "f", "iters", "status", "viol"):
            assert np.array_equal(a[k], b[k]), (kw, k)
        other.close()
    ref.close()


def test_hip_matches_committed_slsqp_solutions(backends, golden_dir):
    """SURVEY 8(c) bridge (ii) on the product path: the HIP solve (through the C ABI) of the committed
    N=10 instances lands on the solutions an independent SLSQP run found (tests/golden/gen/gen_slsqp.py)."""
    import os
    d = np.load(os.path.join(golden_dir, "slsqp_N10.npz"))
    N = int(d["N"])
    for tol, q_tol, pv_tol in ((1e-8, 1e-4, 2e-5), (1e-5, 2e-3, 5e-4)):
        r = backends(N, tol=tol).solve_batch(d["x0"], d["lbx"], d["ubx"], d["p"])
        assert (r["status"] == 0).all()
        dx = np.abs(r["x"] - d["x"])
        assert dx[:, : 7 * N].max() < q_tol
        assert dx[:, 28 * N: 40 * N].max() < pv_tol
        assert (np.abs(r["f"] - d["f"]) <= 1e-5 * np.abs(d["f"])).all()


@pytest.mark.parametrize("name", ["bridge_N10.npz", "bridge_N10_tc.npz", "bridge_N15.npz", "bridge_N20.npz"])
def test_hip_matches_bridge_solutions(backends, golden_dir, name):
    """SURVEY 8(c) bridge (ii) on the product path, widened (tests/golden/gen/gen_bridge.py): the HIP solve lands on the
    solutions independent scipy methods (SLSQP, trust-constr) found from the reference's cold start, or -- where the two
    families end in different local solutions of the non-convex NLP -- on the lower one, which SLSQP, restarted from it,
    confirms.  See tests/test_independent_solver.py::bridge_check for the per-instance criteria."""
    import os
    from test_independent_solver import bridge_check
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    d = np.load(path)
    N = int(d["N"][0])
    r = backends(N, tol=1e-8).solve_batch(d["x0"], d["lbx"], d["ubx"], d["p"])
    rows = iter(range(d["x"].shape[0]))
    tc = name.endswith("_tc.npz")

    def solve(N_, x0, lbx, ubx, p):
        i = next(i for i in rows if not (tc and int(d["status"][i]) == 0))
        assert np.array_equal(x0, d["x0"][i])
        return r["x"][i], float(r["f"][i]), int(r["status"][i])
    counts = bridge_check(name, d, solve)
    print(name, counts)
    assert counts["same"] + counts["confirmed"] >= 0.75 * d["x"].shape[0]


def test_ragged_batches_and_api_misuse(backends):
    """Batch sizes that do not fill a wavefront (3 instances of N=20 share one) or a list slot, and the error
    convention of the C ABI (nonzero return code + message, no exception from the device)."""
    from boundplanner_amd import scenes
    N = 20
    be = backends(N)
    batch = scenes.make_batch(67, N, 5, be.fk, randomize_sets=True)
    full = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    for B in (1, 2, 3, 4, 63, 65):
        r = be.solve_batch(batch["x0"][:B], batch["lbx"][:B], batch["ubx"][:B], batch["p"][:B])
        assert np.array_equal(r["x"], full["x"][:B]) and np.array_equal(r["iters"], full["iters"][:B]), B
    sel = [66, 3, 41]                                    # order / neighbours do not matter
    r = be.solve_batch(batch["x0"][sel], batch["lbx"][sel], batch["ubx"][sel], batch["p"][sel])
    assert np.array_equal(r["x"], full["x"][sel])
    rc = be.lib.bmpc_solve(be._h, 0, None, None, None, None, None, None, None, None, None, None, None, None)
    assert rc != 0                                       # empty batch / null pointers: refused, not a crash
    with pytest.raises(RuntimeError):
        from boundplanner_amd.solver import HipBoundMPC
        HipBoundMPC(2)                                   # horizon below the formulation's minimum


def test_config1_full_batch_against_oracle(backends):
    """BASELINE configs[1] at its full size (1024 instances, N=10, fixed sets), every instance against the oracle.  Over a
    batch this large a few instances take a different line-search decision on rounding-level differences and then need
    one to three iterations more or less; both sides still stop at KKT points of the same NLP, so the solutions agree to
    the size of the last Newton step."""
    from boundplanner_amd import scenes
    N, B = 10, 1024
    be = backends(N)
    batch = scenes.make_batch(B, N, 1024, be.fk, randomize_sets=False)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=0)
    assert (r["status"] == ro["status"]).all()
    conv = (r["status"] == 0) & (ro["status"] == 0)
    assert conv.all()                                   # observed 1024/1024
    dit = np.abs(r["iters"] - ro["iters"])
    # observed: 1017 identical iteration counts, the rest within 5
    assert (dit[conv] == 0).sum() >= 1010 and (dit[conv] <= 1).sum() >= 1018 and dit[conv].max() <= 5
    d_task = np.abs(r["x"][:, 28 * N:40 * N] - ro["x"][:, 28 * N:40 * N]).max(axis=1)
    eq = conv & (dit == 0)
    print(f"configs[1]: converged {conv.sum()}/{B}, same iterations {eq.sum()}, max |d task| same-iters {d_task[eq].max():.1e} all {d_task[conv].max():.1e}")
    assert (d_task[eq] < 2e-5).sum() >= eq.sum() - 3     # the stated bar, per instance
    assert d_task[conv].max() < 1e-3                     # observed 3.3e-4 (one long, ill-conditioned run)
    assert (np.abs(r["f"][conv] - ro["f"][conv]) <= 1e-5 * np.maximum(1.0, np.abs(ro["f"][conv]))).all()
    assert abs(r["iters"].mean() - ro["iters"].mean()) < 0.05


def test_longest_horizon(backends):
    """N = 64, the longest horizon the handle accepts (one instance per wavefront in the thread-per-pair kernels, 63
    Riccati stages): instances that converge on both sides agree with the oracle; one of the six wanders to max_iter on
    the oracle and may or may not do so on the GPU."""
    from boundplanner_amd import scenes
    N, B = 64, 6
    be = backends(N)
    batch = scenes.make_batch(B, N, 64, be.fk, randomize_sets=False)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=0)
    conv = (r["status"] == 0) & (ro["status"] == 0)
    assert conv.sum() >= B - 1
    # observed (round 2): all six converge on both sides, iteration counts 27/37/37/93/68/50 against 27/39/37/95/68/50 (long
    # runs drift by an iteration or two with the rounding of the two builds), task-space solutions within 1.2e-6
    eq = conv & (np.abs(r["iters"] - ro["iters"]) <= 3)
    assert eq.sum() >= conv.sum() - 1
    assert np.abs(r["x"][eq][:, 28 * N:40 * N] - ro["x"][eq][:, 28 * N:40 * N]).max() < 1e-4
    assert np.abs(r["f"][conv] - ro["f"][conv]).max() < 1e-5 * np.abs(ro["f"][conv]).max()
    assert (r["viol"][conv] < 1e-4).all()


def test_poisoned_instance_is_contained(backends):
    """A NaN / infinite parameter in one instance must not hang the batch or leak into its neighbours: the other
    instances return bitwise what they return without it, the poisoned one comes back with a nonzero status."""
    from boundplanner_amd import scenes
    N, B = 10, 9
    be = backends(N)
    batch = scenes.make_batch(B, N, 1024, be.fk)
    clean = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    p = batch["p"].copy()
    p[4, 100] = np.nan          # a reference-path entry of instance 4
    x0 = batch["x0"].copy()
    x0[7, 3] = np.inf           # a start-vector entry of instance 7
    r = be.solve_batch(x0, batch["lbx"], batch["ubx"], p)
    ok = [i for i in range(B) if i not in (4, 7)]
    assert np.array_equal(r["x"][ok], clean["x"][ok]) and np.array_equal(r["iters"][ok], clean["iters"][ok])
    assert r["status"][4] != 0 and r["status"][7] != 0
    assert (r["iters"][[4, 7]] <= be.opts.max_iter).all()


def test_two_handles_from_two_host_threads():
    """One handle per host thread (the C ABI's threading contract): concurrent solves on the same GPU return what the
    same solves return one after the other."""
    import threading
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N = 10
    hs = [HipBoundMPC(N), HipBoundMPC(N)]
    batches = [scenes.make_batch(96, N, 11 + i, hs[0].fk, randomize_sets=True) for i in range(2)]
    seq = [hs[i].solve_batch(b["x0"], b["lbx"], b["ubx"], b["p"]) for i, b in enumerate(batches)]
    par = [None, None]

    def work(i):
        b = batches[i]
        for _ in range(3):
            par[i] = hs[i].solve_batch(b["x0"], b["lbx"], b["ubx"], b["p"])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(2):
        assert np.array_equal(par[i]["x"], seq[i]["x"]) and np.array_equal(par[i]["status"], seq[i]["status"])


def test_one_handle_from_two_threads_is_refused():
    """A handle serves one host thread at a time (include/boundmpc.h): a second thread entering while a solve runs gets a
    nonzero return code and a message, not a corrupted workspace; the first solve is unaffected."""
    import threading
    import time
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N = 10
    be = HipBoundMPC(N)
    big = scenes.make_batch(2048, N, 5, be.fk, randomize_sets=True)
    ref = be.solve_batch(big["x0"], big["lbx"], big["ubx"], big["p"])
    out, errs = {}, []

    def long_solve():
        out["r"] = be.solve_batch(big["x0"], big["lbx"], big["ubx"], big["p"])
    t = threading.Thread(target=long_solve)
    t.start()
    time.sleep(0.02)
    for _ in range(50):                       # hammer the handle from this thread while the other solve is running
        if not t.is_alive():
            break
        try:
            be.solve_batch(big["x0"][:4], big["lbx"][:4], big["ubx"][:4], big["p"][:4])
        except RuntimeError as e:
            errs.append(str(e))
        time.sleep(0.005)
    t.join()
    assert errs and all("in use by another thread" in e for e in errs)
    assert np.array_equal(out["r"]["x"], ref["x"]) and np.array_equal(out["r"]["status"], ref["status"])
