"""Iterate-for-iterate comparison of the HIP path with the oracle -- TEST INFRASTRUCTURE.

Both sides are run with max_iter = k for k = 1 .. K on the same instances (cold start): a solve that reaches the iteration limit
returns the iterate after k accepted steps together with the decisions of its last iteration (HIP: bmpc_debug_inst_state;
oracle: bmpc_oracle_solve_batch_info): barrier parameter, accepted step length and number of rejected trials, dual step length,
inertia correction delta_w, factorisation retries (Gauss-Newton fallback / delta_w escalation), exact-Hessian switch for the next
iteration, stall counter.  An instance is IN STEP up to k while every discrete decision of iterations 0 .. k-1 was the same on
both sides: termination, number of rejected line-search trials (alpha = alpha_ftb / 2^backtracks), a search that found no acceptable
step, factorisation retries, the exact-Hessian switch, the stall counter; the barrier parameter and delta_w, which move in steps of
a factor >= 3, to 1 %.  (The fraction-to-boundary lengths themselves are continuous functions of the iterate and are not decisions.)
For the instances in step the iterates are compared (relative to max(1, |x|) per block).  The first k at which an instance
leaves, and the decision that differed, are recorded.  `self_sensitivity` is the yardstick: the oracle against itself from a start
vector perturbed by 1e-14 relative -- how far rounding-level noise is carried by k iterations of this algorithm on these problems.
"""
import numpy as np

DISCRETE = ("status", "hess_next", "retries", "backtracks", "stall")
STEPPED = ("mu", "delta_w")          # compared to 1 %
FIELDS = ("iters", "status", "mu", "alpha", "alpha_dual", "alpha_ftb", "delta_w", "hess_next", "retries", "backtracks", "err_prev", "stall")


def blocks(N):
    return {"q": (0, 7 * N), "dq": (7 * N, 14 * N), "ddq": (14 * N, 21 * N), "u": (21 * N, 28 * N),
            "p": (28 * N, 34 * N), "v": (34 * N, 40 * N), "slacks": (40 * N, 44 * N + 6)}


def rel_diff(N, xa, xb):
    """per instance: max over the blocks of max|xa - xb| / max(1, max|xb|) of the block"""
    out = np.zeros(xa.shape[0])
    for a, b in blocks(N).values():
        d = np.abs(xa[:, a:b] - xb[:, a:b]).max(axis=1) / np.maximum(1.0, np.abs(xb[:, a:b]).max(axis=1))
        out = np.maximum(out, d)
    return out


def decisions_differ(ih, io):
    """per instance: name of the first decision field that differs between the two info rows, or '' """
    B = ih.shape[0]
    why = np.array([""] * B, dtype=object)
    for name in DISCRETE:
        j = FIELDS.index(name)
        m = (why == "") & (ih[:, j] != io[:, j])
        why[m] = name
    for name in STEPPED:
        j = FIELDS.index(name)
        a, b = ih[:, j], io[:, j]
        m = (why == "") & (np.abs(a - b) > 1e-2 * np.maximum(np.abs(b), 1e-300))
        why[m] = name
    ja = FIELDS.index("alpha")
    m = (why == "") & ((ih[:, ja] > 1e200) != (io[:, ja] > 1e200))      # alpha = 1e300: the search found no acceptable step
    why[m] = "no_acceptable_step"
    return why


def table(N, batch, hip_factory, oracle, ks, nthreads=0, **opts):
    """hip_factory(max_iter) -> HipBoundMPC; returns (rows, per-instance first departure k (0: never), reason)"""
    a = (batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    B = a[0].shape[0]
    in_step = np.ones(B, bool)           # same decisions so far, and nobody has finished early
    left_at = np.zeros(B, int)
    reason = np.array([""] * B, dtype=object)
    rows = []
    for k in ks:
        be = hip_factory(k)
        h = be.solve_batch(*a)
        ih = be.inst_state(B)
        o = oracle.solve_batch_info(N, *a, nthreads=nthreads, max_iter=k, **opts)
        io = o["info"]
        # an instance that converged before k on either side is no longer part of the comparison from there on (its later runs repeat
        # the same final point); it stays "in step" only up to its last common iteration
        running = (h["iters"] == k) & (o["iters"] == k)
        why = decisions_differ(ih, io)
        it_differs = in_step & (h["iters"] != o["iters"])
        newly = in_step & running & (why != "")
        for m, r in ((it_differs, "iterations"),):
            left_at[m & (left_at == 0)] = k; reason[m & (reason == "")] = r
        left_at[newly & (left_at == 0)] = k
        reason[newly & (reason == "")] = why[newly & (reason == "")]
        in_step &= ~(newly | it_differs)
        cmp_mask = in_step & running
        rd = rel_diff(N, h["x"], o["x"])
        rows.append({"k": int(k), "running_both": int(running.sum()), "in_step": int(cmp_mask.sum()),
                     "left_here": int((newly | it_differs).sum()),
                     "left_here_by": {r: int(((newly | it_differs) & (reason == r)).sum()) for r in sorted(set(reason[newly | it_differs]))},
                     "max_rel_dx_in_step": float(rd[cmp_mask].max()) if cmp_mask.any() else None,
                     "median_rel_dx_in_step": float(np.median(rd[cmp_mask])) if cmp_mask.any() else None,
                     "p90_rel_dx_in_step": float(np.quantile(rd[cmp_mask], 0.9)) if cmp_mask.any() else None,
                     "p99_rel_dx_in_step": float(np.quantile(rd[cmp_mask], 0.99)) if cmp_mask.any() else None,
                     "worst_in_step": [int(i) for i in np.argsort(-np.where(cmp_mask, rd, -1))[:3]],
                     "max_rel_dx_out_of_step": float(rd[~in_step & running].max()) if (~in_step & running).any() else None})
    return rows, left_at, reason


def quantiles(rd, m):
    return {"median": float(np.median(rd[m])), "p90": float(np.quantile(rd[m], 0.9)), "p99": float(np.quantile(rd[m], 0.99)), "max": float(rd[m].max())} if m.any() else None


def self_sensitivity(N, batch, oracle, ks, eps=1e-14, seed=1, nthreads=0, **opts):
    """The oracle against itself from a start vector perturbed by eps (relative, uniform): quantiles of the relative distance of iterate k
    over the instances that take the same decisions on both runs."""
    a = (batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    rng = np.random.default_rng(seed)
    x0p = a[0] * (1.0 + eps * rng.uniform(-1, 1, a[0].shape))
    x0p = np.where(a[1] == a[2], a[0], x0p)            # (pinned entries are data, not a start value)
    B = a[0].shape[0]
    in_step = np.ones(B, bool)
    rows = []
    for k in ks:
        r0 = oracle.solve_batch_info(N, *a, nthreads=nthreads, max_iter=k, **opts)
        r1 = oracle.solve_batch_info(N, x0p, *a[1:], nthreads=nthreads, max_iter=k, **opts)
        running = (r0["iters"] == k) & (r1["iters"] == k)
        in_step &= ~((decisions_differ(r0["info"], r1["info"]) != "") & running) & (r0["iters"] == r1["iters"])
        q = quantiles(rel_diff(N, r0["x"], r1["x"]), in_step & running)
        rows.append({"k": int(k), "in_step": int((in_step & running).sum()), **(q or {})})
    return rows
