"""How the HIP path is compared with the oracle, instance by instance -- TEST INFRASTRUCTURE.

Both sides run the same interior-point algorithm in FP64 and differ in summation order and libm only.  They follow the same
iterates until a branch of the algorithm (a line-search acceptance, an inertia decision, a barrier decrease, the termination
test) falls on different sides of that rounding noise; from there on they are two valid runs of the same algorithm from
neighbouring points.  What can be asserted PER INSTANCE, without bars read off observed maxima:

  (P1) solved tightly (tol = 1e-10, max_iter = 200: TIGHT) the two runs end at the same point: joint space <= 1e-5, task
       space <= 1e-7, objective <= 1e-8 relative (SURVEY 8(c) iii).  Instances outside these bars are OUTLIERS: listed
       (JSON), counted, and each must be either the same solution in a weakly determined direction (objective <= 1e-8
       relative, joint <= 1e-4, task <= 1e-5: the 7-DOF arm is redundant for the 6-D task and null-space motion is priced
       by weights 1e-3 / 1e-4, so two points whose KKT errors are both below 1e-10 can be that far apart) or another local
       solution of the non-convex NLP; in both cases a KKT point of the pinned NLP by the multipliers the HIP path returns.
       Measured on the MI355X (tests/diag/diag_parity_tol.py, 512 instances of configs[2]): at tol 1e-8 / 1e-9 / 1e-10 the
       bars hold for 87 / 89 / 97.3 % of the instances, the median joint-space distance is 5e-9 / 6e-10 / 3e-12, and the
       objectives agree to 3e-8 / 8e-9 / 7e-10 relative on every instance -- hence 1e-10 for this statement.
  (P2) at the reference's tol = 1e-5 an accepted iterate is a truncation of that sequence (and sits on the central path at
       mu = 1e-6 instead of 1e-11); its distance to the tight solution of the same side, e_side, is the solver's own
       truncation error (median 3e-3 rad in joint space, 5e-4 in task space on configs[2]).  By the triangle inequality the two loose
       solutions are at most e_hip + e_oracle + (P1 bar) apart -- asserted for every instance (it catches a wrong point
       returned under a right status) -- and the truncation errors of the two sides must have the same distribution
       (quantiles within a factor 2): the HIP path does not stop systematically further from the solution.
"""
import numpy as np

TIGHT = dict(tol=1e-10, max_iter=200)
BARS_TIGHT = {"joint": 1e-5, "task": 1e-7, "f_rel": 1e-8}


def blocks(N):
    return {"q": (0, 7 * N), "dq": (7 * N, 14 * N), "ddq": (14 * N, 21 * N), "u": (21 * N, 28 * N),
            "p": (28 * N, 34 * N), "v": (34 * N, 40 * N), "slacks": (40 * N, 44 * N + 6)}


def _dist(N, xa, xb):
    d = np.abs(xa - xb)
    return {"joint": d[:, :21 * N].max(axis=1), "u": d[:, 21 * N:28 * N].max(axis=1), "task": d[:, 28 * N:40 * N].max(axis=1),
            "slacks": d[:, 40 * N:].max(axis=1)}


def compare(N, hip5, or5, hip8, or8):
    """Results of the four solves (dicts with x, f, iters, status) -> report dict with boolean masks and distances."""
    rep = {"N": N, "B": int(hip5["x"].shape[0])}
    conv8 = (hip8["status"] == 0) & (or8["status"] == 0)
    d8 = _dist(N, hip8["x"], or8["x"])
    df8 = np.abs(hip8["f"] - or8["f"]) / np.maximum(1.0, np.abs(or8["f"]))
    tight = conv8 & (d8["joint"] <= BARS_TIGHT["joint"]) & (d8["task"] <= BARS_TIGHT["task"]) & (df8 <= BARS_TIGHT["f_rel"])
    rep.update(conv8=conv8, d8=d8, df8=df8, tight=tight, outliers=np.nonzero(conv8 & ~tight)[0])
    conv5 = (hip5["status"] == 0) & (or5["status"] == 0)
    rep.update(conv5=conv5, status_equal5=hip5["status"] == or5["status"], status_equal8=hip8["status"] == or8["status"],
               dit5=np.abs(hip5["iters"].astype(int) - or5["iters"].astype(int)))
    rep["d5"] = _dist(N, hip5["x"], or5["x"])
    rep["e_hip"] = _dist(N, hip5["x"], hip8["x"])
    rep["e_or"] = _dist(N, or5["x"], or8["x"])
    return rep


def assert_triangle(rep):
    """(P2) first half: every instance that converged four times and is tight at 1e-8."""
    m = rep["conv5"] & rep["tight"]
    p1 = {"joint": BARS_TIGHT["joint"], "task": BARS_TIGHT["task"]}
    for k in ("joint", "task"):
        bound = rep["e_hip"][k] + rep["e_or"][k] + p1[k]
        bad = m & (rep["d5"][k] > bound * (1 + 1e-9))
        assert not bad.any(), (k, np.nonzero(bad)[0][:5].tolist())


def truncation_quantiles(rep, qs=(0.5, 0.9, 0.99)):
    m = rep["conv5"] & rep["tight"]
    out = {}
    for k in ("joint", "u", "task"):
        out[k] = {"hip": [float(np.quantile(rep["e_hip"][k][m], q)) for q in qs],
                  "oracle": [float(np.quantile(rep["e_or"][k][m], q)) for q in qs]}
    return out


def assert_same_truncation(rep, factor=2.0, floor=1e-9):
    """(P2) second half; meaningful from a few hundred instances on."""
    for k, v in truncation_quantiles(rep).items():
        for a, b in zip(v["hip"], v["oracle"]):
            assert a <= factor * max(b, floor) and b <= factor * max(a, floor), (k, v)


def outlier_records(rep, hip5, or5, hip8, or8, kkt_res=None):
    rec = []
    for i in rep["outliers"]:
        r = {"i": int(i), "iters_tol8": [int(hip8["iters"][i]), int(or8["iters"][i])], "iters_tol5": [int(hip5["iters"][i]), int(or5["iters"][i])],
             "f_tol8": [float(hip8["f"][i]), float(or8["f"][i])], "df_rel": float(rep["df8"][i]),
             "d_tol8": {k: float(v[i]) for k, v in rep["d8"].items()}}
        weak = rep["df8"][i] <= 1e-8 and rep["d8"]["joint"][i] <= 1e-4 and rep["d8"]["task"][i] <= 1e-5
        r["kind"] = "same solution, weakly determined direction" if weak else "another local solution"
        if kkt_res is not None:
            r["hip_stationarity_residual"] = float(kkt_res[i])
        rec.append(r)
    return rec


def summary(rep):
    c5, c8 = rep["conv5"], rep["conv8"]
    s = {"instances": rep["B"], "tight": {"both_converged": int(c8.sum()), "status_equal": int(rep["status_equal8"].sum()),
                                            "within_tight_bars": int(rep["tight"].sum()), "outliers": int(len(rep["outliers"])),
                                            "max_over_tight": {k: float(v[rep["tight"]].max()) for k, v in rep["d8"].items()} if rep["tight"].any() else None},
         "tol1e-5": {"both_converged": int(c5.sum()), "status_equal": int(rep["status_equal5"].sum()),
                     "same_iterations": int((c5 & (rep["dit5"] == 0)).sum()), "within_1_iteration": int((c5 & (rep["dit5"] <= 1)).sum()),
                     "max_d_iters": int(rep["dit5"][c5].max()) if c5.any() else None,
                     "max_d": {k: float(v[c5].max()) for k, v in rep["d5"].items()} if c5.any() else None},
         "truncation_error_quantiles_50_90_99": truncation_quantiles(rep) if (c5 & rep["tight"]).sum() >= 8 else None}
    return s
