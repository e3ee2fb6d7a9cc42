#!/usr/bin/env python3
"""The reference's example (/root/reference/boundplanner_with_mpc_example.py:19-166) as plan-then-track on the MI355X
solver: default iiwa14 start configuration, MPCNode start-up + warm-up solve (:21-29), BoundPlanner.plan_convex_set_path
on the 12-box scene (:102-115; boundplanner_amd.bound_planner, host), the plan handed to `update_reference` (:134) and the
hot loop `while phi < phi_max - 0.001: mpc_node.step()` (:140-157).
BASELINE.json configs[0] ("single instance plumbing"): one instance, N = 15 (the reference default).

    python examples/mpc_example.py            # needs an MI355X (no CPU fallback)
    python examples/mpc_example.py --device   # the same on the device-resident loop, with the example's 12 box
                                              # obstacles and per-step collision sets computed on the GPU
"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation as R

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from boundplanner_amd.mpc_node import MPCNode            # noqa: E402
from boundplanner_amd.robot_model import RobotModel       # noqa: E402
from boundplanner_amd.solver import HipBoundMPC, HipNlpSolver   # noqa: E402


def plan(p0, r0):
    """Example :102-115: sets, via points and orientation schedule from the start pose to the example's goal pose."""
    from boundplanner_amd import scenes
    from boundplanner_amd.bound_planner import BoundPlanner
    boxes, _, goal_p, goal_r = scenes.example_scene()
    planner = BoundPlanner(e_p_max=0.5, obstacles=boxes, workspace_max=[1.0, 0.38, 1.0], workspace_min=[-0.14, -1.0, 0.0], seed=0)
    t0 = time.perf_counter()
    p_via, r_via, bp1_list, sets_via = planner.plan_convex_set_path(p0, goal_p, r0, goal_r)
    print(f"Path planning took {time.perf_counter() - t0:.2f}s: {len(p_via)} via points through {len(sets_via)} of {planner.nr_sets} sets")
    n = len(bp1_list)
    erb = [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180 for _ in range(n)]
    return p_via, r_via, bp1_list, [np.array([0, 0, 1.0])] * n, erb, [s[0] for s in sets_via], [s[1] for s in sets_via], boxes


def main_device():
    """One rollout on the device-resident loop (boundplanner_amd.device_loop), example scene obstacles included."""
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.params import Params, get_default_params
    boxes, q0, goal_p, _ = scenes.example_scene()
    base = get_default_params()
    params = Params(n=15, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    be = HipBoundMPC(15)
    seed = BatchMPCNode(be, q0[None], params)                  # host construction of the start-up BoundMPC object
    loop = DeviceLoop(be, 1)
    loop.set_obstacles(*scenes.boxes_to_sets(boxes))
    loop.set_rollout(0, seed.mpcs[0], seed.q[0], seed.dq[0], seed.ddq[0], seed.jerk[0], seed.qf[0], seed.v[0], seed.p_lie[0])
    loop.upload()
    loop.run(1, log=False)                                     # warm-up solve, example :29
    V = loop.download()
    p0 = V["p_lie"][0].copy()
    p_via, r_via, bp1, br1, erb, a_sets, b_sets, _ = plan(p0[:3], R.from_rotvec(p0[3:]).as_matrix())
    loop.replan(0, seed.mpcs[0], p_via, r_via, bp1, br1, erb, a_sets, b_sets)
    loop.upload()
    L, t0, steps, iters = loop.LOG, time.perf_counter(), 0, []
    while steps < 300:
        log = loop.run(10)
        for row in log[:, 0]:
            steps += 1
            iters.append(row[L["iters"]])
            if row[L["phi"]] >= row[L["phi_max"]] - 0.001:
                break
        print(f"step {steps:3d}  phi {row[L['phi']]:.3f}/{row[L['phi_max']]:.3f}  sector {int(row[L['sector']])}  iters {int(row[L['iters']])}")
        if row[L["phi"]] >= row[L["phi_max"]] - 0.001:
            break
    print(f"reached phi_max in {steps} steps, {np.mean(iters):.1f} iterations/step, "
          f"{1e3 * (time.perf_counter() - t0) / max(steps, 1):.1f} ms/step wall (device loop, 12 obstacles), "
          f"final EE position {np.round(row[L['p_lie']][:3], 4)}")


def main():
    if "--device" in sys.argv:
        return main_device()
    q0 = np.array([0.0, 0.0, 0.0, -np.pi / 2, 0.0, np.pi / 2, 0.0])        # example :20
    be = HipBoundMPC(15)
    node = MPCNode(q0, RobotModel(be.fk), lambda n, dt: HipNlpSolver(n, dt, backend=be))
    node.step()                                                              # warm-up solve, example :29
    p0 = node.p_lie
    p_via, r_via, bp1, br1, erb, a_sets, b_sets, boxes = plan(p0[:3], R.from_rotvec(p0[3:]).as_matrix())
    from boundplanner_amd import scenes
    node.mpc.set_obstacle_sets(*scenes.boxes_to_sets(boxes))                 # per-step collision sets (BoundMPC.py:480-493)
    node.update_reference(p_via, r_via, bp1, br1, erb, a_sets, b_sets, [])
    t0, steps = time.perf_counter(), 0
    while node.mpc.phi_current[0] < node.mpc.phi_max[0] - 0.001 and steps < 300:
        node.step()
        steps += 1
        if steps % 10 == 0:
            print(f"step {steps:3d}  phi {node.mpc.phi_current[0]:.3f}/{node.mpc.phi_max[0]:.3f}  "
                  f"sector {node.mpc.ref_path.sector}/{node.mpc.ref_path.num_sectors}  iters {node.iters[-1]}  "
                  f"t_solve {1e3 * node.t_mpc:.1f} ms")
    print(f"reached phi_max in {steps} steps, {np.mean(node.iters):.1f} iterations/step, "
          f"{1e3 * (time.perf_counter() - t0) / max(steps, 1):.1f} ms/step wall, fails {int(np.sum(node.fails))}, "
          f"final EE position {np.round(node.p_lie[:3], 4)}")


if __name__ == "__main__":
    main()
