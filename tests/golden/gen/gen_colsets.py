"""Generate tests/golden/colsets.npz: per-step collision-avoidance sets produced by the REFERENCE's own finder.

Runs, in the build container only, the reference's unmodified ConvexSetFinder.find_set_collision_avoidance
(/root/reference/bound_planner/BoundPlanner/ConvexSetFinder.py:309-375 with init_halfspaces_point :400-421 and
compute_set_projs_line :491-510) on the 12 box obstacles of its example scene (boundplanner_with_mpc_example.py:38-98),
exactly as BoundMPC.step calls it (BoundMPC.py:480-493: the six collision points at q0 and qf, limit_space=True,
e_max=0.7, then b - joint size).  The one third-party slot on this path -- `self.projl_solver`, qpOASES through
ca.qpsol (ConvexSetFinder.py:87), absent from this image -- is filled by an exact solver of the same QP
(min |p0 + phi (p1 - p0) - x|^2  s.t.  A x <= b, 0 <= phi <= 1: enumeration of the active sets of its KKT system).
Pins rows a10 / f2 of SURVEY.md section 8.  The fixture is data only; no reference source is copied.

    python tests/golden/gen/gen_colsets.py
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.abspath(os.path.join(HERE, ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(1, REF)
sys.path.insert(2, os.path.join(ROOT, "tests"))
sys.path.insert(3, ROOT)
os.chdir(REF)

import oracle_lib as O  # noqa: E402
from boundplanner_amd import scenes  # noqa: E402  (the scene's box constants and start configuration)
from bound_planner.BoundPlanner.ConvexSetFinder import ConvexSetFinder  # noqa: E402
from bound_planner.utils import normalize_set_size  # noqa: E402

JOINT_SIZES = [0.09, 0.12, 0.09, 0.10, 0.07, 0.09, 0.075]      # RobotModel.py:37
MAXR = 24                                                       # 6 box rows + up to 12 obstacles + slack


def exact_segment_polytope_qp(A, b, p0, p1):
    """min_{x, phi} |p0 + phi d - x|^2  s.t.  A x <= b, 0 <= phi <= 1, by enumerating the active sets of the KKT system
    (4 unknowns, <= 8 non-trivial constraints).  Returns u = (x, phi)."""
    d = p1 - p0
    keep = np.abs(A).sum(axis=1) > 0
    A, b = A[keep], b[keep]
    m = A.shape[0]
    C = np.zeros((m + 2, 4)); e = np.zeros(m + 2)
    C[:m, :3] = A; e[:m] = b
    C[m, 3] = 1.0; e[m] = 1.0            # phi <= 1
    C[m + 1, 3] = -1.0; e[m + 1] = 0.0    # -phi <= 0
    H = 2 * np.block([[np.eye(3), -d[:, None]], [-d[None, :], np.array([[d @ d]])]])
    c = 2 * np.concatenate((-p0, [d @ p0]))
    best = None
    for k in range(0, 5):
        for act in itertools.combinations(range(m + 2), k):
            act = list(act)
            if m in act and m + 1 in act:
                continue
            Ca = C[act]
            K = np.block([[H, Ca.T], [Ca, np.zeros((k, k))]])
            rhs = np.concatenate((-c, e[act]))
            sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
            if np.abs(K @ sol - rhs).max() > 1e-9:
                continue
            u, lam = sol[:4], sol[4:]
            if (C @ u - e).max() > 1e-9 or (lam < -1e-9).any():
                continue
            f = 0.5 * u @ H @ u + c @ u
            if best is None or f < best[0] - 1e-12:
                best = (f, u)
        if best is not None and k >= 1:
            pass
    assert best is not None
    return best[1]


class ExactQP:
    """Stands in for the CasADi qpsol function object of ConvexSetFinder.py:87 (call convention of :498-505)."""

    def __call__(self, x0, lbx, ubx, lbg, ubg, p):
        p = np.asarray(p, float).ravel()
        A = p[:45].reshape(3, 15).T          # a_set.T.flatten() -> column-major 15 x 3
        b = p[45:60]
        return {"x": exact_segment_polytope_qp(A, b, p[60:63], p[63:66])}


def outside(pts, boxes, margin=0.002):
    """every point is outside every box by at least `margin` (the robot does not start inside an obstacle)"""
    lo, hi = boxes[:, None, :3] - margin, boxes[:, None, 3:] + margin
    inside = ((pts[None] >= lo) & (pts[None] <= hi)).all(axis=2)
    return not inside.any()


def shell_scene(center, radius=0.3, size=0.02):
    """14 small cubes on a sphere around `center` (face and corner directions of a cube): every one of them yields a
    halfspace and none hides another, so a set around `center` needs 6 + 14 = 20 rows > max_set_size."""
    dirs = [np.array(v, float) for v in itertools.product((-1, 0, 1), repeat=3) if sum(abs(c) for c in v) in (1, 3)]
    return np.array([np.concatenate((center + radius * d / np.linalg.norm(d) - size, center + radius * d / np.linalg.norm(d) + size))
                     for d in dirs])


def run_scene(boxes, c0, cf):
    sets, pts = scenes.boxes_to_sets(boxes)
    obs_sets = normalize_set_size([[a.copy(), b.copy()] for a, b in sets], 15)      # what the planner hands to the finder
    finder = ConvexSetFinder(obs_sets, pts, np.array([1.0, 0.38, 1.0]), np.array([-0.14, -1.0, 0.0]))
    finder.projl_solver = ExactQP()
    P = c0.shape[0]
    A = np.zeros((P, 6, MAXR, 3)); bb = np.zeros((P, 6, MAXR)); nrows = np.zeros((P, 6), int); coll = np.zeros((P, 6), bool)
    for i in range(P):
        for j in range(6):
            a_c, b_c, col = finder.find_set_collision_avoidance(c0[i, j], cf[i, j], limit_space=True, e_max=0.7)
            n = a_c.shape[0]
            A[i, j, :n] = a_c; bb[i, j, :n] = b_c - JOINT_SIZES[j]       # BoundMPC.py:490
            nrows[i, j] = n; coll[i, j] = col
    assert np.isfinite(A).all() and np.isfinite(bb).all()
    return A, bb, nrows, coll


def main():
    boxes, q_ex, _, _ = scenes.example_scene()
    rng = np.random.default_rng(2025)
    q0s, qfs = [], []
    while len(q0s) < 56:
        i = len(q0s)
        spread = [0.15, 0.4, 0.8, 1.2][i % 4]
        q0 = q_ex + rng.uniform(-spread, spread, 7)
        qf = q0 + rng.uniform(-0.3, 0.3, 7) * (i % 3)          # every third pair: q0 == qf (a stationary horizon end)
        k0, kf = O.fk_batch(q0[None])["col_pts"][0], O.fk_batch(qf[None])["col_pts"][0]
        if outside(k0, boxes) and outside(kf, boxes):
            q0s.append(q0); qfs.append(qf)
    q0s, qfs = np.array(q0s), np.array(qfs)
    c0, cf = O.fk_batch(q0s)["col_pts"], O.fk_batch(qfs)["col_pts"]
    A, bb, nrows, coll = run_scene(boxes, c0, cf)
    print("example scene: pairs", len(q0s), "rows per set: min", nrows.min(), "max", nrows.max(), "segment-through-obstacle flags:",
          int(coll.sum()), "sets that discarded obstacles:", int((nrows < 6 + len(boxes)).sum()), "of", nrows.size)
    # second scene: more halfspaces than max_set_size (the reference only prints an error there, util_functions.py:126-134)
    q2 = q_ex[None] + rng.uniform(-0.1, 0.1, (2, 7))
    c2 = O.fk_batch(q2)["col_pts"]
    boxes2 = shell_scene(c2[0, 4])                 # around collision point 4 (joint_7 origin) of the first configuration
    A2, b2, n2, coll2 = run_scene(boxes2, c2, c2)
    print("shell scene: rows per set", n2.tolist())
    assert n2.max() > 15
    np.savez_compressed(os.path.join(OUT, "colsets.npz"), boxes=boxes, q0=q0s, qf=qfs, p0=c0, p1=cf, A=A, b=bb, nrows=nrows,
                        collision=coll, joint_sizes=np.array(JOINT_SIZES[:6]),
                        shell_boxes=boxes2, shell_q=q2, shell_p=c2, shell_A=A2, shell_b=b2, shell_nrows=n2)


if __name__ == "__main__":
    main()
