"""Synthetic problem-instance generators for the benchmark configurations.

Every instance is built exactly the way the reference builds one receding-horizon step after
a (re)plan: a 2-via-point ReferencePath from the start pose to the goal pose
(boundplanner_with_mpc_example.py:129-135 conventions: br1 seed (0,0,1), e_r_bound +-90 deg),
BoundMPC.update() state seeding (BoundMPC.py:271-336), then BoundMPC.step() preparation
(BoundMPC.py:388-589) -> (x0, lbx, ubx, p).

  config 2: fixed sets   -- EE set = workspace box [-1,-1,0]-[1,1,1.2] (BoundPlanner.py:32-33),
                            collision sets = 0.7 m boxes around each collision point
  config 3: as config 2 + K~U{3..9} random unit-normal halfspaces per set
"""
import numpy as np
from scipy.spatial.transform import Rotation as R

from .bound_mpc import BoundMPC
from .params import (COL_JOINT_SIZES, Params, Q_LIM_LOWER, Q_LIM_UPPER, get_default_params,
                     normalize_set_size)


def _box_set(lo, hi):
    a = np.vstack((np.eye(3), -np.eye(3)))
    b = np.concatenate((np.asarray(hi, float), -np.asarray(lo, float)))
    return a, b


def _rand_unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def sample_start_goal(rng, fk_batch, n, robot=None):
    """q_start, q_goal ~ U(0.5 q_lo, 0.5 q_hi); reject collision points below z=0.05 or
    start/goal closer than 0.1 m.  robot: table of .robots (None = iiwa14); unlimited joints are drawn from +-pi/2."""
    lo, hi = Q_LIM_LOWER, Q_LIM_UPPER
    if robot is not None:
        lo = np.maximum(np.asarray(robot["q_lower"], float), -np.pi); hi = np.minimum(np.asarray(robot["q_upper"], float), np.pi)
    qs, qg = [], []
    while len(qs) < n:
        m = max(64, 2 * (n - len(qs)))
        a = rng.uniform(0.5 * lo, 0.5 * hi, size=(m, 7))
        b = rng.uniform(0.5 * lo, 0.5 * hi, size=(m, 7))
        fa, fb = fk_batch(a), fk_batch(b)
        ok = (fa["col_pts"][:, :, 2].min(axis=1) >= 0.05) & (fb["col_pts"][:, :, 2].min(axis=1) >= 0.05)
        ok &= (fa["ee_pos"][:, 2] >= 0.05) & (fb["ee_pos"][:, 2] >= 0.05)
        ok &= np.linalg.norm(fa["ee_pos"] - fb["ee_pos"], axis=1) >= 0.1
        for i in np.nonzero(ok)[0]:
            if len(qs) < n:
                qs.append(a[i]); qg.append(b[i])
    return np.array(qs), np.array(qg)


_ERB = np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180
_WS_LO, _WS_HI = [-1.0, -1.0, 0.0], [1.0, 1.0, 1.2]


def _build_instance(args):
    """Deterministic part of one instance (no random draws, no kinematics calls): the reference's seeding sequence
    BoundMPC() -> update() -> step() preparation on a 2-via-point path.  Runs in a worker process when make_batch is
    given a pool.  Returns (x0, lbx, ubx, p, mpc)."""
    N, dt, q0, p0, p1, rot0, rot1, col0, a_ee, b_ee, aj_extra, bj_extra, keep_mpc, robot = args
    base = get_default_params()
    prm = Params(n=N, dt=dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    sets = normalize_set_size([[a_ee, b_ee]], 15)
    mpc = BoundMPC([p0[:3].copy(), p0[:3].copy()], [rot0.copy(), rot0.copy()],
                   [np.array([1.0, 0.0, 0.0])], [np.array([1.0, 0.0, 0.0])], [_ERB.copy()],
                   [np.zeros((15, 3))], [np.ones(15)], [], p0=p0, params=prm, robot=robot)
    mpc.update([p0[:3].copy(), p1.copy()], [rot0.copy(), rot1.copy()],
               [np.array([0.0, 0.0, 1.0])], [np.array([0.0, 0.0, 1.0])], [_ERB.copy()],
               [sets[0][0]], [sets[0][1]], [], np.zeros(6), p0=p0, params=prm)
    w0, lbx, ubx, p, _ = mpc.prepare(q0, np.zeros(7), np.zeros(7), p0, np.zeros(6),
                                     np.zeros(7), q0, col_pts0=col0, col_ptsf=col0)
    if aj_extra is not None:
        # extra random halfspaces on the 6 collision sets, appended after the 6 box rows
        aj = p[515:785].reshape(6, 3, 15).transpose(0, 2, 1).copy()   # [pt][row][c]
        bj = p[785:875].reshape(15, 6).T.copy()                       # [pt][row]
        for i in range(6):
            k = len(bj_extra[i])
            aj[i, 6:6 + k] = aj_extra[i]
            bj[i, 6:6 + k] = bj_extra[i]
        p[515:785] = aj.transpose(0, 2, 1).reshape(-1)
        p[785:875] = bj.T.reshape(-1)
    return w0, lbx, ubx, p, (mpc if keep_mpc else None)


def make_batch(B, N, seed, fk_batch, randomize_sets=False, dt=0.1, pool=None, robot=None):
    """Returns dict with x0, lbx, ubx, p ([B, n] float64, row-major per instance) + the state
    needed to drive closed loops.  All random draws happen here, in instance order, from one PCG64 stream; the
    per-instance host construction (1.3 ms of Python each) is deterministic given the draws and is mapped over
    `pool` (a multiprocessing pool of processes that never touch the GPU) when one is given -- same result, and
    `mpcs` is then not returned."""
    rng = np.random.default_rng(seed)
    sizes = COL_JOINT_SIZES if robot is None else robot["col_joint_sizes"]
    q_start, q_goal = sample_start_goal(rng, fk_batch, B, robot)
    fs, fg = fk_batch(q_start), fk_batch(q_goal)
    n_w = 44 * N + 6
    out = {k: np.zeros((B, n_w)) for k in ("x0", "lbx", "ubx")}
    out["p"] = np.zeros((B, 875))
    out["q_start"], out["q_goal"] = q_start, q_goal
    jobs = []
    for b in range(B):
        p0 = np.concatenate((fs["ee_pos"][b], R.from_matrix(fs["ee_rot"][b]).as_rotvec()))
        p1 = fg["ee_pos"][b]
        col0 = fs["col_pts"][b]
        a_ee, b_ee = _box_set(_WS_LO, _WS_HI)
        aj_extra = bj_extra = None
        if randomize_sets:
            k = rng.integers(3, 10)
            rows_a, rows_b = [], []
            while len(rows_a) < k:
                a = _rand_unit(rng)
                d = rng.uniform(0.05, 0.4)
                bb = a @ p0[:3] + d
                if a @ p1 <= bb:       # both path ends must satisfy the row
                    rows_a.append(a); rows_b.append(bb)
            a_ee = np.vstack((a_ee, rows_a)); b_ee = np.concatenate((b_ee, rows_b))
            aj_extra, bj_extra = [], []
            for i in range(6):
                k = rng.integers(3, 10)
                ra, rb = np.zeros((k, 3)), np.zeros(k)
                for r in range(k):
                    a = _rand_unit(rng)
                    ra[r] = a
                    rb[r] = a @ col0[i] + rng.uniform(0.05, 0.4) - sizes[i]
                aj_extra.append(ra); bj_extra.append(rb)
        jobs.append((N, dt, q_start[b], p0, p1, fs["ee_rot"][b], fg["ee_rot"][b], col0, a_ee, b_ee, aj_extra, bj_extra,
                     pool is None, robot))
    res = map(_build_instance, jobs) if pool is None else pool.imap(_build_instance, jobs, chunksize=64)
    mpcs = []
    for b, (w0, lbx, ubx, p, mpc) in enumerate(res):
        out["x0"][b], out["lbx"][b], out["ubx"][b], out["p"][b] = w0, lbx, ubx, p
        mpcs.append(mpc)
    if pool is None:
        out["mpcs"] = mpcs
    return out


def example_scene():
    """The 12 box obstacles of the reference's example scene as [xmin, ymin, zmin, xmax, ymax, zmax]
    (boundplanner_with_mpc_example.py:38-98: an open box of four 2 cm walls around (0.45, -0.48), the table, a shelf, two
    walls, two blocks and two 8 cm cubes), its start configuration (:19-25) and goal pose (:33-34)."""
    size, s_box, w, pb, h_box = 0.04, 0.12, 0.02, (0.45, -0.48, 0.05), 0.18
    top = pb[2] + h_box
    boxes = [
        [pb[0] + s_box - w, pb[1] - s_box, 0.0, pb[0] + s_box, pb[1] + s_box, top],
        [pb[0] - s_box, pb[1] - s_box, 0.0, pb[0] - s_box + w, pb[1] + s_box, top],
        [pb[0] - s_box, pb[1] - s_box - w, 0.0, pb[0] + s_box, pb[1] - s_box, top],
        [pb[0] - s_box, pb[1] + s_box, 0.0, pb[0] + s_box, pb[1] + s_box + w, top],
        [0.2, -1.0, -0.1, 1.0, 1.0, 0.0],
        [-0.3, -1.0, 0.53, 0.2, -0.35, 1.0],
        [-0.2, -1.0, 0.0, -0.14, 1.0, 1.0],
        [-1.0, 0.38, 0.0, 1.0, 0.5, 1.0],
        [0.4, -0.05, 0.0, 0.5, 0.05, 0.15],
        [0.1, -0.55, 0.0, 0.3, -0.35, 0.07],
        [0.5 - size, -0.2 - size, 0.03 - size, 0.5 + size, -0.2 + size, 0.03 + size],
        [0.4 - size, 0.3 - size, 0.03 - size, 0.4 + size, 0.3 + size, 0.03 + size],
    ]
    q0 = np.array([0.0, 0.0, 0.0, -np.pi / 2, 0.0, np.pi / 2, 0.0])
    goal_p = np.array([0.45, -0.5, 0.2])
    goal_r = R.from_euler("XYZ", [0, 90, 0], degrees=True).as_matrix()
    return np.array(boxes), q0, goal_p, goal_r


def boxes_to_sets(boxes):
    """Axis-aligned boxes -> ([A, b] polytopes, vertex arrays): the obstacle form the per-step collision-set finder
    takes (ConvexSetFinder.py:309-375; the reference builds the same from its box list with cdd)."""
    sets, pts = [], []
    for bx in np.asarray(boxes, float):
        lo, hi = bx[:3], bx[3:]
        sets.append([np.vstack((np.eye(3), -np.eye(3))), np.concatenate((hi, -lo))])
        pts.append(np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]))
    return sets, pts
