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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rollouts", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--chunk", type=int, default=25, help="MPC steps per bmpc_loop_run call (progress lines)")
    ap.add_argument("--seed", type=int, default=4096)
    ap.add_argument("--scene", choices=["free", "example"], default="free", help="free: configs[4] (no obstacles); example: the 12 "
                    "boxes of the reference's example scene, starts scattered around its start configuration, its goal pose")
    ap.add_argument("--groups", type=int, default=3, help="rollout groups stepped concurrently (own solver handle and stream "
                    "each): the straggler tail of one group's solve overlaps the bulk of another's")
    args = ap.parse_args()
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.params import Params, get_default_params, normalize_set_size
    from boundplanner_amd.solver import HipBoundMPC

    N, R = args.horizon, args.rollouts
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    G = max(1, args.groups)
    bounds = [R * g // G for g in range(G + 1)]
    bes = [HipBoundMPC(N, max_batch=bounds[g + 1] - bounds[g]) for g in range(G)]
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
    print(f"plan-time host setup of {R} rollouts: {t_plan:.1f} s", file=sys.stderr, flush=True)

    L = loop.LOG
    iters, fails, reached_at = [], [], np.full(R, -1)
    ms_total = ms_solve = 0.0
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        n = min(args.chunk, args.steps - done)
        if G == 1:
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
        at_end = log[:, :, L["phi"]] >= log[:, :, L["phi_max"]] - 0.001
        for s in range(n):
            reached_at[(reached_at < 0) & at_end[s]] = done + s + 1
        done += n
        print(f"step {done}/{args.steps}: {1e3 * (time.perf_counter() - t0) / done:.1f} ms/step, mean iters {iters[-1].mean():.1f}, "
              f"at path end {(reached_at > 0).mean():.3f}", file=sys.stderr, flush=True)
    wall = time.perf_counter() - t0
    it = np.concatenate(iters)
    dead = float(log[-1, :, L["dead"]].mean())
    out = {
        "scene": args.scene,
        "config": f"BASELINE configs[4]: closed loop, {R} rollouts x {args.steps} steps, N={N}, fixed sets, warm start (reference Q11), "
                  "device-resident loop (prepare kernel -> batched solve -> finish kernel)" + (f", {G} rollout groups in flight" if G > 1 else ""),
        "groups": G,
        "solves": int(R * args.steps), "wall_s": wall, "solves_per_s": R * args.steps / wall,
        "gpu_stream_ms_total": ms_total, "host_ms_inside_solves": ms_solve, "ms_per_step": 1e3 * wall / args.steps,
        "plan_time_host_setup_s": t_plan,
        "iters_mean": float(it.mean()), "iters_p50": float(np.median(it)), "iters_p99": float(np.percentile(it, 99)),
        "iters_first_step_mean": float(it[0].mean()), "fail_frac": float(np.concatenate(fails).mean()), "dead_frac": dead,
        "reached_end_frac": float((reached_at > 0).mean()),
        "steps_to_end_median": float(np.median(reached_at[reached_at > 0])) if (reached_at > 0).any() else None,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
