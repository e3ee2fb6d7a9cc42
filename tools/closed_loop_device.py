#!/usr/bin/env python3
"""BASELINE.json configs[4] on the device-resident loop: R parallel rollouts x S MPC steps, horizon N, fixed sets,
warm start exactly as the reference warm-starts (previous solution unshifted, Q11).  Per step: prepare kernel ->
batched HIP solve -> finish kernel; the per-rollout state never leaves HBM, the host only polls the solver's
active count.  Rollouts that reached the path end keep stepping (idle solves count).

  python tools/closed_loop_device.py --rollouts 4096 --horizon 30 --steps 200
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# more than three rollout groups: their streams need hardware queues of their own (the runtime's default of 4 makes the fourth
# stream share one: measured 23 k instead of 37 k solves/s with four groups); read when HIP initialises, as in bench.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rollouts", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--chunk", type=int, default=25, help="MPC steps per bmpc_loop_run call (progress lines)")
    ap.add_argument("--seed", type=int, default=4096)
    ap.add_argument("--scene", choices=["free", "example"], default="free", help="free: configs[4] (no obstacles); example: the 12 "
                    "boxes of the reference's example scene, starts scattered around its start configuration, its goal pose")
    ap.add_argument("--async", dest="async_", action="store_true", help="rollouts do not wait for each other (bmpc_loop_run_async); one group")
    ap.add_argument("--dump-failing", default=None, help="one group only: .npz with the inputs (x0, lbx, ubx, p) of the last step's solves that did not converge, "
                    "and the per-rollout counts of failed solves")
    ap.add_argument("--diagnose", default=None, help="write a JSON with the classification of the rollouts that did not reach the path end")
    ap.add_argument("--groups", type=int, default=3, help="rollout groups stepped concurrently (own solver handle and stream "
                    "each): the straggler tail of one group's solve overlaps the bulk of another's")
    ap.add_argument("--opt", action="append", default=[], help="solver option key=value (bmpc_opts field), A/B runs")
    return ap


def run(args, progress=True):
    """The run described by `args` (parser()); returns the result record (bench.py calls this for its configs[4] leg)."""
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.params import Params, get_default_params, normalize_set_size
    from boundplanner_amd.solver import HipBoundMPC

    N, R = args.horizon, args.rollouts
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    G = 1 if args.async_ else max(1, args.groups)
    bounds = [R * g // G for g in range(G + 1)]
    kw = {}
    for kv in args.opt:
        k, v = kv.split("=")
        kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    bes = [HipBoundMPC(N, max_batch=bounds[g + 1] - bounds[g], **kw) for g in range(G)]
    be = bes[0]
    rng = np.random.default_rng(args.seed)
    t0 = time.perf_counter()
    obs = None
    if args.scene == "example":
        boxes, q0, goal_p, goal_r = scenes.example_scene()
        obs = scenes.boxes_to_sets(boxes)
        q_start = q0 + rng.uniform(-0.15, 0.15, size=(R, 7))
        fs = be.fk(q_start)
        fg = {"ee_pos": np.tile(goal_p, (R, 1)), "ee_rot": np.tile(goal_r, (R, 1, 1))}
    else:
        q_start, q_goal = scenes.sample_start_goal(rng, be.fk, R)
        fs, fg = be.fk(q_start), be.fk(q_goal)
    seed_objs = BatchMPCNode(be, q_start, params)          # host construction of the R BoundMPC objects (trivial start-up path)
    loops = [DeviceLoop(bes[g], bounds[g + 1] - bounds[g]) for g in range(G)]
    a_ee, b_ee = scenes._box_set([-1.0, -1.0, 0.0], [1.0, 1.0, 1.2])
    for g, loop in enumerate(loops):
        if obs is not None:
            loop.set_obstacles(*obs)
        for i, r in enumerate(range(bounds[g], bounds[g + 1])):
            loop.set_rollout(i, seed_objs.mpcs[r], seed_objs.q[r], seed_objs.dq[r], seed_objs.ddq[r], seed_objs.jerk[r],
                             seed_objs.qf[r], seed_objs.v[r], seed_objs.p_lie[r])
        loop.upload()
        loop.run(1, log=False)                             # start-up solve on the trivial path (example :28-29)
        V = loop.download()
        for i, r in enumerate(range(bounds[g], bounds[g + 1])):
            sets = normalize_set_size([[a_ee, b_ee]], 15)
            loop.replan(i, seed_objs.mpcs[r], [V["p_lie"][i][:3].copy(), fg["ee_pos"][r].copy()],
                        [fs["ee_rot"][r].copy(), fg["ee_rot"][r].copy()], [np.array([0.0, 0, 1])], [np.array([0.0, 0, 1])],
                        [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180], [sets[0][0]], [sets[0][1]])
        loop.upload()
    loop = loops[0]
    t_plan = time.perf_counter() - t0
    if progress:
        print(f"plan-time host setup of {R} rollouts: {t_plan:.1f} s", file=sys.stderr, flush=True)

    L = loop.LOG
    iters, fails, reached_at = [], [], np.full(R, -1)
    hist = []                  # per step: phi, q, error_count, iters, status (for --diagnose)
    ms_total = ms_solve = 0.0
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        n = min(args.chunk, args.steps - done)
        if args.async_:
            log = loop.run_async(n)
        elif G == 1:
            log = loop.run(n)
        else:       # one host thread per group; bmpc_loop_run releases the GIL for its whole duration
            import threading
            parts = [None] * G
            def work(g):
                parts[g] = loops[g].run(n)
            th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
            for t in th: t.start()
            for t in th: t.join()
            log = np.concatenate(parts, axis=1)
        ms_total += max(l.ms_total for l in loops); ms_solve += max(l.ms_solve for l in loops)
        iters.append(log[:, :, L["iters"]]); fails.append(log[:, :, L["error_count"]] > 0)
        if args.diagnose:
            hist.append(np.concatenate((log[:, :, [L["phi"], L["phi_max"], L["error_count"], L["iters"], L["status"], L["viol"], L["dead"]]],
                                        log[:, :, L["q"]]), axis=2))
        at_end = log[:, :, L["phi"]] >= log[:, :, L["phi_max"]] - 0.001
        for s in range(n):
            reached_at[(reached_at < 0) & at_end[s]] = done + s + 1
        done += n
        if progress:
            print(f"step {done}/{args.steps}: {1e3 * (time.perf_counter() - t0) / done:.1f} ms/step, mean iters {iters[-1].mean():.1f}, "
                  f"at path end {(reached_at > 0).mean():.3f}", file=sys.stderr, flush=True)
    wall = time.perf_counter() - t0
    it = np.concatenate(iters)
    dead = float(log[-1, :, L["dead"]].mean())
    out = {
        "scene": args.scene,
        "config": f"BASELINE configs[4]: closed loop, {R} rollouts x {args.steps} steps, N={N}, fixed sets, warm start (reference Q11), "
                  "device-resident loop (prepare kernel -> batched solve -> finish kernel)" + (f", {G} rollout groups in flight" if G > 1 else "")
                  + (", rollouts not in lock step (bmpc_loop_run_async)" if args.async_ else ""),
        "groups": G,
        "solves": int(R * args.steps), "wall_s": wall, "solves_per_s": R * args.steps / wall,
        "gpu_stream_ms_total": ms_total, "host_ms_inside_solves": ms_solve, "ms_per_step": 1e3 * wall / args.steps,
        "plan_time_host_setup_s": t_plan,
        "iters_mean": float(it.mean()), "iters_p50": float(np.median(it)), "iters_p99": float(np.percentile(it, 99)),
        "iters_first_step_mean": float(it[0].mean()), "fail_frac": float(np.concatenate(fails).mean()), "dead_frac": dead,
        "reached_end_frac": float((reached_at > 0).mean()),
        "steps_to_end_median": float(np.median(reached_at[reached_at > 0])) if (reached_at > 0).any() else None,
        # what bounds the schedules: lock step pays the slowest solve of every step, no lock step the slowest rollout's own chain
        "iters_sum_of_per_step_max": int(it.max(axis=1).sum()), "iters_per_rollout_total_max": int(it.sum(axis=0).max()),
        "iters_per_rollout_total_mean": float(it.sum(axis=0).mean()),
    }
    if args.async_:
        out["lanes"] = dict(bes[0].lane_stats(), fast_lane_max=int(os.environ.get("BMPC_FAST_LANE", "0")),
                            reserved_cus=int(os.environ.get("BMPC_FAST_CUS", "0")))
        tot = it.sum(axis=0)
        out["iters_per_rollout_total_quantiles"] = {f"p{q}": float(np.percentile(tot, q)) for q in (50, 75, 90, 95, 98, 99)}
    if args.dump_failing and G == 1:
        sol = loop.solution()
        x0_, lbx_, ubx_, p_ = loop.problem()
        bad = np.nonzero(sol["status"] != 0)[0]
        nfail = np.concatenate(fails).sum(axis=0)
        order = np.argsort(-nfail)[:16]
        np.savez(args.dump_failing, rows=bad, x0=x0_[bad], lbx=lbx_[bad], ubx=ubx_[bad], p=p_[bad], iters=sol["iters"][bad], status=sol["status"][bad],
                 fail_counts=nfail, worst=order, worst_counts=nfail[order], N=N)
        print(f"failed solves per rollout: worst {list(zip(order.tolist(), nfail[order].tolist()))}; last step not converged: {bad.tolist()}", file=sys.stderr)
    if args.diagnose:
        from boundplanner_amd.params import Q_LIM_LOWER, Q_LIM_UPPER
        H = np.concatenate(hist)                   # [steps][R][7 + 7]
        phi, phimax, err, its, stat, viol, deadv, q = H[:, :, 0], H[:, :, 1], H[:, :, 2], H[:, :, 3], H[:, :, 4], H[:, :, 5], H[:, :, 6], H[:, :, 7:]
        stuck = ~(reached_at > 0)
        last = min(50, H.shape[0] - 1)
        prog = phi[-1] - phi[-1 - last]                                        # progress over the last 50 steps
        at_limit = (np.minimum(q[-1] - Q_LIM_LOWER, Q_LIM_UPPER - q[-1]) < 2e-3).any(axis=1)
        failing = (err[-last:] > 0).mean(axis=0) > 0.2
        not_conv = (stat[-last:] != 0).mean(axis=0) > 0.2
        frac_left = 1.0 - phi[-1] / np.maximum(phimax[-1], 1e-9)
        cls = np.full(R, "reached", dtype=object)
        cls[stuck] = "stalled_other"
        cls[stuck & (prog > 2e-3)] = "still_moving"
        cls[stuck & (prog <= 2e-3) & at_limit] = "stalled_at_joint_limit"
        cls[stuck & (prog <= 2e-3) & ~at_limit & (failing | not_conv)] = "stalled_solver_failing"
        cls[deadv[-1] != 0] = "dead"
        diag = {"rollouts": R, "steps": int(H.shape[0]), "classes": {c: int((cls == c).sum()) for c in sorted(set(cls))},
                "stuck_frac_left_quantiles": [float(x) for x in np.quantile(frac_left[stuck], [0.1, 0.5, 0.9])] if stuck.any() else None,
                "stuck_mean_iters_last50": float(its[-last:, stuck].mean()) if stuck.any() else None,
                "reached_mean_iters_last50": float(its[-last:, ~stuck].mean()),
                "stuck_not_converged_frac_last50": float((stat[-last:, stuck] != 0).mean()) if stuck.any() else None,
                "phi_decrease_events": int((np.diff(phi, axis=0) < -1e-6).sum()), "phi_decrease_max": float(-np.diff(phi, axis=0).min()),
                "accepted_viol_max": float(viol[(err == 0) & (deadv == 0)].max()),
                "examples": {c: [int(i) for i in np.nonzero(cls == c)[0][:5]] for c in sorted(set(cls)) if c != "reached"}}
        # nearest joint limit of the stalled-at-limit rollouts: which joints
        if (cls == "stalled_at_joint_limit").any():
            sel = cls == "stalled_at_joint_limit"
            marg = np.minimum(q[-1][sel] - Q_LIM_LOWER, Q_LIM_UPPER - q[-1][sel])
            diag["limit_joint_hist"] = [int(x) for x in np.bincount(marg.argmin(axis=1), minlength=7)]
        # what holds the stalled rollouts back: the multipliers of their last solve (is it a KKT point? which rows are active?)
        if G == 1 and stuck.any():
            import torch
            n_w, n_g = be.n_w, be.n_g
            dev = torch.device("cuda", 0)
            lg = torch.empty((R, n_g), dtype=torch.float64, device=dev); lx = torch.empty((R, n_w), dtype=torch.float64, device=dev)
            be.multipliers_dev(R, lg.data_ptr(), lx.data_ptr())
            lg, lx = lg.cpu().numpy(), lx.cpu().numpy()
            sol = loop.solution()
            x0_, lbx_, ubx_, p_ = loop.problem()
            sel = np.nonzero((cls == "stalled_other") & (sol["status"] == 0))[0]
            g0 = 35 * (N - 1)
            rows = lg[:, g0:g0 + 112 * (N - 1)].reshape(R, N - 1, 112)
            groups = {"ee_set": rows[:, :, 0:15], "rot_bounds": rows[:, :, 15:21], "collision_sets": rows[:, :, 21:111], "phi_cap": rows[:, :, 111:112]}
            act = {k: np.abs(v).reshape(R, -1).max(axis=1) for k, v in groups.items()}
            act["terminal_rows"] = np.abs(lg[:, g0 + 112 * (N - 1):]).max(axis=1)
            for nm, blk in (("q_bounds", 0), ("dq_bounds", 1), ("ddq_bounds", 2), ("u_bounds", 3)):
                act[nm] = np.abs(lx[:, blk * 7 * N:(blk + 1) * 7 * N].reshape(R, 7, N)[:, :, 1:]).reshape(R, -1).max(axis=1)
            thr = 1e-2
            diag["stalled_other_active_rows_frac"] = {k: float((v[sel] > thr).mean()) for k, v in act.items()}
            diag["reached_active_rows_frac"] = {k: float((v[~stuck] > thr).mean()) for k, v in act.items()}
            diag["stalled_other_multiplier_median"] = {k: float(np.median(v[sel])) for k, v in act.items()}
            # stationarity of the pinned full-space NLP with these multipliers on a few stalled rollouts (oracle = checker)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            res = []
            for i in sel[:6]:
                _, _, gr, J = O.nlp_eval(N, sol["x"][i], p_[i])
                res.append(float(np.abs(gr + J.T @ lg[i] + lx[i]).max()))
            diag["stalled_other_kkt_stationarity_residual"] = res
            diag["stalled_other_final_speed_max"] = float(np.abs(sol["x"][sel][:, 7 * N + 1:14 * N:N]).max())
        json.dump(diag, open(args.diagnose, "w"), indent=1)
        print(json.dumps(diag), file=sys.stderr)
    for lp in loops:
        lp.close()
    return out


def main():
    print(json.dumps(run(parser().parse_args())))


if __name__ == "__main__":
    main()
