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


# Parity statement (tests/parity_lib.py): (P1) solved tightly (tol = 1e-10) the HIP path and the oracle end at the same point --
# joint space <= 1e-5, task space <= 1e-7, objective <= 1e-8 relative, per instance; outliers are listed and must be KKT points by the
# multipliers the HIP path returns -- and (P2) at the reference's tol = 1e-5 the two accepted iterates are no further apart than the
# two sides' own truncation errors (distance between a side's tol-1e-5 and tol-1e-8 solutions) add up to, per instance.
def four_solves(backends, N, batch, want_lam=True):
    """HIP and oracle at the reference's tolerance (1e-5) and solved tightly (parity_lib.TIGHT) on the same inputs."""
    import parity_lib as PL
    a = (batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    hip5 = backends(N).solve_batch(*a, want_g=True)
    hip8 = backends(N, **PL.TIGHT).solve_batch(*a, want_lam=want_lam)
    or5 = O.solve_batch(N, *a, nthreads=0)
    or8 = O.solve_batch(N, *a, nthreads=0, **PL.TIGHT)
    return hip5, or5, hip8, or8


def stationarity(N, batch, res, rows):
    """max |grad f + J_g^T lam_g + lam_x| of the pinned full-space NLP at the returned point, for the given rows."""
    out = {}
    for i in rows:
        _, _, gr, J = O.nlp_eval(N, res["x"][i], batch["p"][i])
        out[int(i)] = float(np.abs(gr + J.T @ res["lam_g"][i] + res["lam_x"][i]).max())
    return out


def check_parity(N, batch, hip5, or5, hip8, or8, max_outliers, max_status_diff=1, dump=None):
    import json
    import os
    import parity_lib as PL
    rep = PL.compare(N, hip5, or5, hip8, or8)
    kkt = stationarity(N, batch, hip8, rep["outliers"]) if hip8.get("lam_g") is not None else None
    summ = PL.summary(rep)
    print(json.dumps(summ))
    recs = PL.outlier_records(rep, hip5, or5, hip8, or8, kkt)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if dump and os.path.isdir(out_dir):
        json.dump({"summary": summ, "tight": PL.TIGHT, "bars_tight": PL.BARS_TIGHT, "outliers": recs}, open(os.path.join(out_dir, dump), "w"), indent=1)
    assert (~rep["status_equal5"]).sum() <= max_status_diff and (~rep["status_equal8"]).sum() <= max_status_diff
    # (P1) per instance; the outliers are KKT points of the pinned NLP by the returned multipliers
    assert len(rep["outliers"]) <= max_outliers, recs
    if kkt is not None:
        for i, v in kkt.items():
            assert v < 1e-8, (i, v)
    # (P2) per instance
    PL.assert_triangle(rep)
    return rep, summ


@pytest.mark.parametrize("N,seed,rnd,B", [(6, 6, True, 16), (10, 1024, False, 64), (20, 8192, True, 48),
                                          (15, 15, True, 24),      # the reference's default horizon (util_functions.py:49)
                                          (30, 4096, False, 24),   # configs[4]
                                          (3, 3, True, 8)])        # shortest horizon the handle accepts
def test_solve_matches_oracle(backends, N, seed, rnd, B):
    from boundplanner_amd import scenes
    be = backends(N)
    batch = scenes.make_batch(B, N, seed, be.fk, randomize_sets=rnd)
    hip5, or5, hip8, or8 = four_solves(backends, N, batch)
    rep, summ = check_parity(N, batch, hip5, or5, hip8, or8, max_outliers=max(1, B // 12))
    conv = rep["conv5"]
    assert conv.sum() >= B - 1                     # (N=30: one instance of the 24 may run into max_iter on both sides)
    assert (rep["dit5"][conv] == 0).mean() >= 0.75 and np.median(rep["dit5"][conv]) == 0
    assert np.abs(hip5["viol"][conv] - or5["viol"][conv]).max() < 1e-4
    # g returned by the kernel == the pinned full-space g evaluated at the returned x
    for i in np.nonzero(conv)[0][:8]:
        _, g, _, _ = O.nlp_eval(N, hip5["x"][i], batch["p"][i], jac=False)
        assert np.abs(g - hip5["g"][i]).max() < 1e-9


def test_split_index_variants_and_slacks0(backends):
    from boundplanner_amd import scenes
    N, B = 10, 12
    be = backends(N)
    batch = dict(scenes.make_batch(B, N, 77, be.fk, randomize_sets=True))
    p = batch["p"].copy()
    p[0::3, 0:5] = [0, 3, N, N, N]
    p[1::3, 0:5] = [0, 2, 5, N, N]
    p[:, 5:11] = [0.01, 0, 0.02, 0, 0, 0.005]
    batch["p"] = p
    hip5, or5, hip8, or8 = four_solves(backends, N, batch)
    rep, _ = check_parity(N, batch, hip5, or5, hip8, or8, max_outliers=1, max_status_diff=2)
    assert rep["conv5"].sum() >= B - 2 and rep["conv8"].sum() >= B - 2


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


def test_sync_and_async_entry_agree(backends):
    """The synchronous host-pointer entry and the asynchronous device-pointer entry solve the same batch: bitwise identical
    results."""
    import torch
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 10, 96
    be0 = backends(N)
    batch = scenes.make_batch(B, N, 77, be0.fk, randomize_sets=True)
    r0 = be0.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    assert (r0["status"] == 0).mean() > 0.95
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
    """BASELINE configs[2] at its full size (8192 instances, N=20, randomized convex sets), EVERY instance against the oracle at
    the reference's tolerance and solved tightly (tests/parity_lib.py): (P1) tight per-instance bars with the outliers
    written to gpurun_out/r03_parity_config2.json (committed as profiles/r03_parity_config2.json), (P2) the triangle bound per
    instance and equal truncation-error distributions at 1e-5."""
    import parity_lib as PL
    from boundplanner_amd import scenes
    N, B = 20, 8192
    be = backends(N)
    batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
    hip5, or5, hip8, or8 = four_solves(backends, N, batch)
    rep, summ = check_parity(N, batch, hip5, or5, hip8, or8, max_outliers=int(0.05 * B), max_status_diff=16, dump="r03_parity_config2.json")
    assert rep["conv5"].sum() >= 0.995 * B and rep["conv8"].sum() >= 0.99 * B
    PL.assert_same_truncation(rep)
    conv = rep["conv5"]
    assert abs(hip5["iters"][conv].mean() - or5["iters"][conv].mean()) < 0.1
    assert (rep["dit5"][conv] == 0).mean() >= 0.9
    assert np.abs(hip5["viol"][conv] - or5["viol"][conv]).max() < 2e-4


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


@pytest.mark.parametrize("name", ["bridge_N10.npz", "bridge_N10_tc.npz", "bridge_N15.npz", "bridge_N20.npz", "bridge_N30.npz"])
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
    """BASELINE configs[1] at its full size (1024 instances, N=10, fixed sets), every instance against the oracle at both
    tolerances (tests/parity_lib.py)."""
    import parity_lib as PL
    from boundplanner_amd import scenes
    N, B = 10, 1024
    be = backends(N)
    batch = scenes.make_batch(B, N, 1024, be.fk, randomize_sets=False)
    hip5, or5, hip8, or8 = four_solves(backends, N, batch)
    rep, summ = check_parity(N, batch, hip5, or5, hip8, or8, max_outliers=int(0.05 * B), max_status_diff=0, dump="r03_parity_config1.json")
    assert rep["conv5"].all() and rep["conv8"].sum() >= B - 2
    PL.assert_same_truncation(rep)
    assert abs(hip5["iters"].mean() - or5["iters"].mean()) < 0.05
    assert (rep["dit5"] == 0).mean() >= 0.97


def test_config3_all_65536_rows_on_one_gpu(backends):
    """BASELINE configs[3] (65536 instances = 8 shards of 8192, N=20, the generator and seeds of bench.py --gpus 8) through ONE
    GPU: all rows in one call through a 16384-slot pool, then shards 0, 3 and 7 alone -- bitwise equal to their rows of the big
    call.  That equality is what makes the all-gather of a sharded run exact, here at the config's full size."""
    import multiprocessing as mp
    import os
    import bench
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, S, G = 20, 8192, 8
    be = backends(N)
    with mp.get_context("fork").Pool(min(16, len(os.sched_getaffinity(0)))) as pool:     # the workers never touch the GPU
        shards = [scenes.make_batch(S, N, bench.batch_seed(G, r, 0), be.fk, randomize_sets=True, pool=pool) for r in range(G)]
    cat = {k: np.concatenate([sh[k] for sh in shards]) for k in ("x0", "lbx", "ubx", "p")}
    big = HipBoundMPC(N, pool_slots=16384)
    rb = big.solve_batch(cat["x0"], cat["lbx"], cat["ubx"], cat["p"])
    big.close()
    assert rb["x"].shape == (G * S, 44 * N + 6)
    ok = (rb["status"] == 0) | (rb["viol"] < 1e-4)
    assert ok.mean() > 0.995 and (rb["status"] == 0).mean() > 0.995
    for r in (0, 3, 7):
        sh = shards[r]
        alone = be.solve_batch(sh["x0"], sh["lbx"], sh["ubx"], sh["p"])
        for k in ("x", "f", "iters", "status", "viol"):
            assert np.array_equal(alone[k], rb[k][r * S:(r + 1) * S]), (r, k)
    assert not np.array_equal(shards[0]["p"], shards[1]["p"])          # the shards are different problems (seeds [65536, 0, r])


def test_ric_variants_agree_bitwise(tmp_path):
    """bmpc_k_ric (throughput variant, three noinline sweeps), bmpc_k_ric_lat (one body, used below 512 active instances) and the
    speculative pair bmpc_k_ric_att + bmpc_k_ric_sel (below 160: the factorisation attempts of an iteration side by side, then
    the first successful one is taken) are compilations / schedules of the same arithmetic; results must not depend on which one
    ran.  BMPC_RIC_LAT_BELOW / BMPC_RIC_SPEC_BELOW are read once per process, so the settings run in child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import torch\n"
            "from boundplanner_amd import scenes; from boundplanner_amd.solver import HipBoundMPC\n"
            "be = HipBoundMPC(20); b = scenes.make_batch(700, 20, 8192, be.fk, randomize_sets=True)\n"
            "r = be.solve_batch(b['x0'], b['lbx'], b['ubx'], b['p'])\n"
            "np.savez(sys.argv[1], **{k: r[k] for k in ('x', 'f', 'iters', 'status', 'viol')})\n") % root
    out = []
    # (the last two: the tail regime's line search, bmpc_k_trial_spec -- four step lengths of a search side by side -- never / always)
    for lat, spec, tspec in (("0", "0", None), ("1000000000", "0", None), ("1000000000", "1000000000", None), ("0", "0", "0"), ("0", "0", "1000000000")):
        path = str(tmp_path / f"ric_{lat}_{spec}_{tspec}.npz")
        env = dict(os.environ, BMPC_RIC_LAT_BELOW=lat, BMPC_RIC_SPEC_BELOW=spec)
        if tspec is not None:
            env["BMPC_TRIAL_SPEC_WGS"] = tspec
            env["BMPC_EVAL_SPLIT_WGS"] = tspec      # (and the two-wavefront k_eval of the tail regime, bmpc_k_eval_curv_split)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=600)
        out.append(np.load(path))
    for k in ("x", "f", "iters", "status", "viol"):
        assert np.array_equal(out[0][k], out[1][k]), k
        assert np.array_equal(out[0][k], out[2][k]), ("speculative pair", k)
        assert np.array_equal(out[0][k], out[3][k]), ("sequential line search everywhere", k)
        assert np.array_equal(out[0][k], out[4][k]), ("speculative line search everywhere", k)
    assert out[0]["iters"].max() > 40       # stragglers included


def test_config4_generator_against_oracle(backends):
    """The configs[4] generator at its horizon (N=30, seed 4096, fixed sets), 1024 cold-start instances -- the first solve of the
    closed loop -- every instance against the oracle at both tolerances (tests/parity_lib.py), outliers to
    gpurun_out/r03_parity_config4_N30.json."""
    import parity_lib as PL
    from boundplanner_amd import scenes
    N, B = 30, 1024
    be = backends(N)
    batch = scenes.make_batch(B, N, 4096, be.fk, randomize_sets=False)
    hip5, or5, hip8, or8 = four_solves(backends, N, batch)
    rep, summ = check_parity(N, batch, hip5, or5, hip8, or8, max_outliers=int(0.05 * B), max_status_diff=8, dump="r03_parity_config4_N30.json")
    assert rep["conv5"].sum() >= 0.99 * B and rep["conv8"].sum() >= 0.98 * B
    PL.assert_same_truncation(rep)


def test_trial_repeats_are_scheduling_only():
    """bmpc_opts.trial_repeats: rejected line-search trials are repeated inside the super-step that rejected them (0: one trial per
    super-step, as in rounds 1-2).  Every instance sees the same sequence of trials, so the results are bitwise the same -- with a
    slot per instance and with the rows streaming through a small pool."""
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 20, 600
    ref = None
    for kw in (dict(trial_repeats=0), dict(trial_repeats=4), dict(trial_repeats=9), dict(trial_repeats=3, pool_slots=128)):
        be = HipBoundMPC(N, **kw)
        if ref is None:
            batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
        r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
        if ref is None:
            ref = r
            assert r["iters"].max() > 40       # stragglers included
            continue
        for k in ("x", "f", "iters", "status", "viol"):
            assert np.array_equal(ref[k], r[k]), (kw, k)


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


def test_watchdog_turns_a_stuck_stream_into_an_error():
    """A kernel that does not return must cost the caller an error code, not a host thread stuck in hipStreamSynchronize
    (DESIGN.md section 7).  bmpc_debug_spin occupies the handle's stream for 3 s (a bounded spin: it ends by itself); with
    watchdog_ms = 300 the solve behind it returns rc 5 and a message after ~0.3 s, from the synchronous entry and from the
    worker thread of the asynchronous one; the handle then refuses further work, a fresh handle is unaffected."""
    import time
    import torch
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, B = 10, 8
    ok = HipBoundMPC(N)
    batch = scenes.make_batch(B, N, 3, ok.fk, randomize_sets=True)
    ref = ok.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    be = HipBoundMPC(N, watchdog_ms=300)
    be.debug_spin(3000)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="watchdog"):
        be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    assert 0.25 < time.perf_counter() - t0 < 2.0
    with pytest.raises(RuntimeError, match="unusable"):
        be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    # asynchronous entry: the worker thread gives up the same way, bmpc_wait reports it
    dev = torch.device("cuda", 0)
    big = lambda a: np.nan_to_num(a, posinf=1e20, neginf=-1e20)
    d = {k: torch.from_numpy(big(batch[k])).to(dev) for k in ("x0", "lbx", "ubx", "p")}
    x = torch.empty((B, ok.n_w), dtype=torch.float64, device=dev)
    f = torch.empty(B, dtype=torch.float64, device=dev); viol = torch.empty(B, dtype=torch.float64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    be2 = HipBoundMPC(N, watchdog_ms=300, max_batch=B)
    be2.debug_spin(3000)
    be2.solve_dev_async(B, d["x0"].data_ptr(), d["lbx"].data_ptr(), d["ubx"].data_ptr(), d["p"].data_ptr(), x.data_ptr(),
                        f.data_ptr(), it.data_ptr(), st.data_ptr(), viol.data_ptr())
    with pytest.raises(RuntimeError, match="watchdog"):
        be2.wait()
    time.sleep(3.2)                                   # the spin kernels end by themselves
    torch.cuda.synchronize(dev)
    again = ok.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    assert np.array_equal(again["x"], ref["x"])
    be.close(); be2.close()


def test_host_batch_stream_prefetches_and_returns_what_a_direct_solve_returns():
    """HostBatchStream (bench.py's leg with the transfers inside): inputs from pinned host memory on an upload stream, two staging
    sets per handle, a handle's next call fetched while its current one is solved, outputs through a download stream.  Five calls of
    two batches on two handles (the third call of handle 0 and the second of handle 1 are prefetched): the outputs left in the
    pinned buffers are bitwise those of direct solves of the same rows."""
    import torch
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_stream import HostBatchStream
    from boundplanner_amd.solver import HipBoundMPC
    N, B, M, nd = 10, 64, 2, 6
    dev = torch.device("cuda:0")
    bes = [HipBoundMPC(N, pool_slots=64) for _ in range(2)]
    batch = scenes.make_batch(nd * B, N, 31, bes[0].fk, randomize_sets=True)
    pinned = {k: torch.from_numpy(np.ascontiguousarray(batch[k])).pin_memory() for k in ("x0", "lbx", "ubx", "p")}
    hs = HostBatchStream(bes, pinned, B, M, dev)
    hs.run(5 * M)                       # calls 0, 2, 4 on handle 0; 1, 3 on handle 1
    ref = HipBoundMPC(N)
    for j, g in ((0, 4), (1, 3)):
        s = (g * M) % nd
        rows = slice(s * B, s * B + M * B)
        r = ref.solve_batch(batch["x0"][rows], batch["lbx"][rows], batch["ubx"][rows], batch["p"][rows])
        for k in ("x", "f", "iters", "status", "viol"):
            assert np.array_equal(hs.h_out[j][k][:M * B].numpy(), r[k]), (j, k)
    assert hs.fetched[0][0] == 4 and hs.fetched[1][1] == 3      # those calls' inputs did come from the prefetch
    del hs
    for b in bes + [ref]:
        b.close()
