#!/usr/bin/env python3
"""Closed-loop receding-horizon run (BASELINE.json configs[4], reduced): R parallel rollouts advanced in
lock step with ONE batched GPU solve per MPC step, warm-started exactly as the reference warm-starts
(previous solution unshifted, Q11).  The host side of every rollout (BoundMPC.prepare / finish,
ReferencePath, integrate_joint) is the sequential per-instance logic of the reference; it is NOT timed
as part of the solver and is reported separately.

  python tools/closed_loop_run.py --rollouts 256 --horizon 30 --max-steps 200
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
    ap.add_argument("--rollouts", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=4096)
    args = ap.parse_args()
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.params import Params, get_default_params, normalize_set_size
    from boundplanner_amd.solver import HipBoundMPC

    N, Rn = args.horizon, args.rollouts
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    be = HipBoundMPC(N, max_batch=Rn)
    rng = np.random.default_rng(args.seed)
    q_start, q_goal = scenes.sample_start_goal(rng, be.fk, Rn)
    fs, fg = be.fk(q_start), be.fk(q_goal)
    node = BatchMPCNode(be, q_start, params)
    node.step()                                     # start-up solve on the trivial path (example :28-29)
    a_ee, b_ee = scenes._box_set([-1.0, -1.0, 0.0], [1.0, 1.0, 1.2])
    for b in range(Rn):
        sets = normalize_set_size([[a_ee, b_ee]], 15)
        node.update_reference(b, [node.p_lie[b][:3].copy(), fg["ee_pos"][b].copy()],
                              [fs["ee_rot"][b].copy(), fg["ee_rot"][b].copy()], [np.array([0.0, 0, 1])],
                              [np.array([0.0, 0, 1])], [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180],
                              [sets[0][0]], [sets[0][1]])
    node.iters.clear(); node.t_solve.clear(); node.t_host.clear(); node.fails.clear()
    t0 = time.perf_counter()
    steps = 0
    reached_at = np.full(Rn, -1)
    while steps < args.max_steps:
        node.step()
        steps += 1
        d = node.done()
        reached_at[(reached_at < 0) & d] = steps
        if d.all():
            break
    wall = time.perf_counter() - t0
    it = np.array(node.iters)
    out = {
        "config": f"closed loop: {Rn} rollouts x {steps} steps, N={N}, fixed sets, warm start (reference Q11)",
        "solves": int(Rn * steps), "wall_s": wall, "gpu_solve_s": float(np.sum(node.t_solve)), "host_s": float(np.sum(node.t_host)),
        "solves_per_s_gpu_calls": Rn * steps / float(np.sum(node.t_solve)), "solves_per_s_end_to_end": Rn * steps / wall,
        "ms_per_batched_solve": 1e3 * float(np.mean(node.t_solve)), "iters_mean": float(it.mean()), "iters_p99": float(np.percentile(it, 99)),
        "iters_first_step_mean": float(it[0].mean()), "fail_frac": float(np.mean(node.fails)),
        "reached_end_frac": float((reached_at > 0).mean()), "steps_to_end_median": float(np.median(reached_at[reached_at > 0])) if (reached_at > 0).any() else None,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
